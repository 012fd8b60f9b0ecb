#!/usr/bin/env python
"""Prompt wav + text in, WAVEX file out: IndexTTS on the MI355X engine's own API.

Three engine objects do what the reference's driver (IndexTTS/Inference_IndexTTS_ONNX.py) spreads over six ONNX Runtime
sessions: `IndexCond.run` (its graph A: prompt audio -> speaker conditioning for the vocoder + the GPT prompt's conditioning
rows), `IndexGPT.generate` (graphs B, C, D and the whole greedy loop over graph E — prompt pass, KV cache, repeat penalty,
stop test — as ONE call whose loop runs on the device), `BigVGANVocoder.run_latent` (graph F: the stacked last hidden
states -> int16 waveform).  `--device-type cuda` keeps the per-sentence tensors on the device between the three calls
(`generate_torch` / `run_latent_torch`); `cpu` passes numpy arrays.  Both forms write the same file.

With no checkpoint on disk the weights are the seeded synthetic ones (`--small`: reduced models for smoke runs): the output
is noise-like audio, but every shape, dtype and call is the real one.  For the reference driver's own call sequence through
`import mi355tts.ort_compat as onnxruntime`, see INTEGRATION.md section 4 and tests/test_gpu_compat.py.

    python examples/indextts_infer.py --prompt prompt.wav --text "..." --out generated.wav [--small] [--device-type cuda]
"""
import argparse
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))

from mi355tts import audio_io, weights                                       # noqa: E402
from mi355tts.bigvgan import BigVGANVocoder                                   # noqa: E402
from mi355tts.config import BigVGANConfig, IndexCondConfig, IndexGPTConfig    # noqa: E402
from mi355tts.indextts import IndexCond, IndexGPT                             # noqa: E402
from mi355tts.indextts_text import TextNormalizer, TextTokenizer              # noqa: E402


def build_tokenizer(model_file):
    import sentencepiece as spm
    sp = spm.SentencePieceProcessor(model_file=model_file)
    norm = TextNormalizer()
    try:
        norm.load()                            # WeTextProcessing / wetext when installed (number / date verbalisation is theirs alone)
    except ImportError:
        class Passthrough:
            def normalize(self, t):
                return t
        norm = TextNormalizer(zh=Passthrough(), en=Passthrough())
        print("WeTextProcessing is not installed: text goes to the tokenizer un-verbalised")
    return sp, TextTokenizer(sp, norm)


def build_engines(args, vocab):
    if args.small:
        gcfg = dataclasses.replace(IndexGPTConfig.small(), text_tokens=vocab + 2, max_text_pos=130, max_seq=512, max_mel_pos=300,
                                   max_generate_length=args.max_generate_length or 200)
        ccfg = IndexCondConfig.small()
        vcfg = BigVGANConfig(num_mels=gcfg.hidden, upsample_initial_channel=ccfg.voc_initial, upsample_rates=(4, 2),
                             upsample_kernel_sizes=(8, 4), use_bias_at_final=True, pre_layernorm=True, speaker_cond=True)
    else:
        gcfg, ccfg, vcfg = IndexGPTConfig(), IndexCondConfig(), BigVGANConfig.indextts()
        if args.max_generate_length:
            gcfg = dataclasses.replace(gcfg, max_generate_length=args.max_generate_length)
    fast = not args.small
    state = lambda spec: weights.synth_state(spec, args.seed, fast=fast)
    cond = IndexCond(ccfg, state(weights.cond_spec(ccfg)))
    gpt = IndexGPT(gcfg, state(weights.gpt_spec(gcfg)), dtype=args.dtype)
    voc = BigVGANVocoder(vcfg, state(weights.bigvgan_spec(vcfg)), dtype=args.dtype)
    return (gcfg, ccfg, vcfg), (cond, gpt, voc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", help="reference audio (RIFF/WAVE); default: a synthetic 3 s tone")
    ap.add_argument("--text", default="The quick brown fox jumps over the lazy dog. Pack my box with five dozen liquor jugs!")
    ap.add_argument("--out", default="generated.wav")
    ap.add_argument("--tokenizer", default=os.path.join(ROOT, "tests", "golden", "indextts_sp.model"),
                    help="sentencepiece model; default: the small fixture model of this repo")
    ap.add_argument("--dtype", default="f16", choices=["f32", "f16", "bf16"])
    ap.add_argument("--device-type", default="cpu", choices=["cpu", "cuda"], help="cuda: per-sentence tensors stay in HBM between the engine calls")
    ap.add_argument("--small", action="store_true", help="reduced synthetic models (smoke runs)")
    ap.add_argument("--max-generate-length", type=int, default=None)
    ap.add_argument("--ignore-stop", action="store_true", help="decode to the length limit (synthetic weights emit the stop code at random)")
    ap.add_argument("--seed", type=int, default=9527)
    args = ap.parse_args()

    sp, tokenizer = build_tokenizer(args.tokenizer)
    (gcfg, ccfg, vcfg), (cond, gpt, voc) = build_engines(args, sp.get_piece_size())
    rate = vcfg.sampling_rate
    if args.prompt:
        prompt_audio = np.asarray(audio_io.load_prompt(args.prompt, rate), dtype=np.int16).reshape(-1)
    else:
        prompt_audio = (0.1 * 32767 * np.sin(2 * np.pi * 220 * np.arange(3 * rate) / rate)).astype(np.int16)
    on_device = args.device_type == "cuda"
    if on_device:
        import torch
        dev = torch.device("cuda", 0)

    t_start = time.time()
    voc_cond_flat, conds_latent = cond.run(prompt_audio)                 # once per speaker
    stage_conds, embed_cond = cond.split_conds(voc_cond_flat)
    stops = [] if args.ignore_stop else [gcfg.stop_mel_token]
    gap = np.zeros((1, 1, int(rate * 0.2)), dtype=np.int16)              # 0.2 s of silence behind every sentence
    if on_device:
        voc_cond_dev = torch.from_numpy(np.concatenate([np.ravel(c) for c in stage_conds] + [np.ravel(embed_cond)]).astype(np.float32)).to(dev)
        penalty_dev = torch.ones(gcfg.mel_codes, dtype=torch.float32, device=dev)     # carried from sentence to sentence, like the host form
    pieces, n_codes = [], 0
    sentences = tokenizer.split_sentences(tokenizer.tokenize(args.text))
    for sentence in sentences:
        print("Generate the Voice for '" + "".join(sentence).replace("▁", " ") + "'")
        ids = np.asarray(tokenizer.convert_tokens_to_ids(sentence), dtype=np.int32)
        t_dec = time.time()
        if on_device:
            prompt_rows, prompt_len = gpt.concat(conds_latent[None], gpt.text_embed(ids), gpt.mel_embed(gcfg.start_mel_token, 0)[0])
            budget = gcfg.max_generate_length - int(prompt_len[0])
            codes = torch.zeros(max(budget, 1), dtype=torch.int32, device=dev)
            hidden = torch.zeros((max(budget, 1), gcfg.hidden), dtype=torch.float32, device=dev)
            n = gpt.generate_torch(torch.from_numpy(prompt_rows[0]).to(dev), budget, codes, hidden, stop_tokens=stops, repeat_penality=penalty_dev)
            hidden = hidden[:n].contiguous()
        else:
            _, hidden, _ = gpt.generate(conds_latent[None], ids, stop_tokens=stops)
            n = hidden.shape[0]
        print(f"Decode Speed: {n / max(time.time() - t_dec, 1e-9):.3f} tokens/s ({n} tokens)")
        n_codes += n
        if n >= 3:                                 # the vocoder needs three codes: (n - 2) * hop + 30 samples
            if on_device:
                wav = voc.run_latent_torch(hidden, voc_cond_dev).cpu().numpy()
            else:
                wav = voc.run_latent(hidden, list(stage_conds) + [embed_cond])
            pieces.append(np.concatenate([wav, gap], axis=-1))
    elapsed = time.time() - t_start
    out = np.concatenate(pieces, axis=-1) if pieces else gap
    audio_io.write_wavex(args.out, out.reshape(-1), rate)
    secs = out.size / rate
    print(f"{args.out}: {len(sentences)} sentence(s), {n_codes} mel codes, {secs:.2f} s of audio in {elapsed:.3f} s (RTF {elapsed / max(secs, 1e-9):.4f})")
    for e in (cond, gpt, voc):
        e.close()


if __name__ == "__main__":
    main()
