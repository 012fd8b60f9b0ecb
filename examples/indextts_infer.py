#!/usr/bin/env python
"""Prompt wav + text in, WAVEX file out: the IndexTTS driver flow on the MI355X engine.

What the reference's IndexTTS/Inference_IndexTTS_ONNX.py does with six ONNX Runtime sessions (:578-800), written against this
repo's onnxruntime-shaped module with the driver's own variable names and feed bookkeeping: session A (prompt audio -> vocoder
conditioning + conds_latent), per sentence sessions B / C / D (embeddings, concat), the greedy loop over session E (KV cache fed
back as OrtValues, repeat penalty on the host exactly as the driver keeps it) and C, session F (latents -> waveform), then the WAVEX
write.  `--device-type cuda` makes every `ortvalue_from_numpy(x, device_type, DEVICE_ID)` a device-resident value, as the driver
does under its CUDA provider.  With no checkpoint on disk the weights are the seeded synthetic ones (`--small`: reduced models for
smoke runs): the output is noise-like audio, but every shape, dtype and call is the real one.

    python examples/indextts_infer.py --prompt prompt.wav --text "..." --out generated.wav [--small] [--device-type cuda]

(The reference concatenates nothing across sentences — `generated_wav` is overwritten per sentence and only the last one reaches
`sf.write`, Inference_IndexTTS_ONNX.py:791,803; this script keeps every sentence, which is what the `save_generated_wav` list the
driver declares at :713 is evidently for.)
"""
import argparse
import dataclasses
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))

import mi355tts.ort_compat as onnxruntime                                    # noqa: E402   (the reference: `import onnxruntime`)
from mi355tts import audio_io, weights                                       # noqa: E402
from mi355tts.config import BigVGANConfig, IndexCondConfig, IndexGPTConfig    # noqa: E402
from mi355tts.indextts_text import TextNormalizer, TextTokenizer              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", help="reference audio (RIFF/WAVE); default: a synthetic 3 s tone")
    ap.add_argument("--text", default="The quick brown fox jumps over the lazy dog. Pack my box with five dozen liquor jugs!")
    ap.add_argument("--out", default="generated.wav")
    ap.add_argument("--tokenizer", default=os.path.join(ROOT, "tests", "golden", "indextts_sp.model"),
                    help="sentencepiece model (the driver's bpe.model); default: the small fixture model of this repo")
    ap.add_argument("--dtype", default="f16", choices=["f32", "f16", "bf16"])
    ap.add_argument("--device-type", default="cpu", choices=["cpu", "cuda"], help="where the driver's OrtValues live")
    ap.add_argument("--small", action="store_true", help="reduced synthetic models (smoke runs)")
    ap.add_argument("--max-generate-length", type=int, default=None)
    ap.add_argument("--ignore-stop", action="store_true", help="decode to the length limit (synthetic weights emit the stop code at random)")
    ap.add_argument("--seed", type=int, default=9527)
    a = ap.parse_args()

    import sentencepiece as spm
    sp = spm.SentencePieceProcessor(model_file=a.tokenizer)
    normalizer = TextNormalizer()
    try:
        normalizer.load()                      # WeTextProcessing / wetext, as the driver does (:578-579)
    except ImportError:                        # not installed: identity normalisers (number / date verbalisation is theirs alone)
        class _Identity:
            def normalize(self, t):
                return t
        normalizer = TextNormalizer(zh=_Identity(), en=_Identity())
        print("WeTextProcessing is not installed: text goes to the tokenizer un-verbalised")
    tokenizer = TextTokenizer(sp, normalizer)

    # ---- models: synthetic seeded weights -> manifests in a scratch directory --------------------------------------------------
    if a.small:
        gcfg = dataclasses.replace(IndexGPTConfig.small(), text_tokens=sp.get_piece_size() + 2, max_text_pos=130, max_seq=512,
                                   max_mel_pos=300, max_generate_length=a.max_generate_length or 200)
        ccfg = IndexCondConfig.small()
        vcfg = BigVGANConfig(num_mels=gcfg.hidden, upsample_initial_channel=ccfg.voc_initial, upsample_rates=(4, 2),
                             upsample_kernel_sizes=(8, 4), use_bias_at_final=True, pre_layernorm=True, speaker_cond=True)
    else:
        gcfg, ccfg, vcfg = IndexGPTConfig(), IndexCondConfig(), BigVGANConfig.indextts()
        if a.max_generate_length:
            gcfg = dataclasses.replace(gcfg, max_generate_length=a.max_generate_length)
    MAX_GENERATE_LENGTH, REPEAT_PENALITY, PENALITY_RANGE = gcfg.max_generate_length, gcfg.repeat_penalty, gcfg.penalty_range
    STOP_TOKEN = [] if a.ignore_stop else [gcfg.stop_mel_token]
    SAMPLE_RATE, DEVICE_ID, device_type = vcfg.sampling_rate, 0, a.device_type
    tmp = tempfile.mkdtemp(prefix="mi355tts_ix_")
    fast = not a.small
    files = {"gpt": (weights.pack_gpt, weights.gpt_spec, gcfg), "cond": (weights.pack_cond, weights.cond_spec, ccfg),
             "voc": (weights.pack_bigvgan, weights.bigvgan_spec, vcfg)}
    wf = {}
    for k, (pack, spec, cfg) in files.items():
        wf[k] = os.path.join(tmp, k + ".npy")
        np.save(wf[k], pack(cfg, weights.synth_state(spec(cfg), a.seed, fast=fast)))
    mk = lambda g, cfg, w, dt: onnxruntime.save_model(os.path.join(tmp, g + ".mi355.json"), g, cfg, wf[w], dt)
    onnx_model_A = mk("IndexTTS_A", ccfg, "cond", "f32")
    onnx_model_B, onnx_model_C, onnx_model_D, onnx_model_E = (mk("IndexTTS_" + g, gcfg, "gpt", a.dtype) for g in "BCDE")
    onnx_model_F = mk("IndexTTS_F", vcfg, "voc", a.dtype)

    session_opts = onnxruntime.SessionOptions()
    session_opts.add_session_config_entry("session.set_denormal_as_zero", "1")
    S = lambda p: onnxruntime.InferenceSession(p, sess_options=session_opts, providers=[], provider_options=None)
    ort_session_A, ort_session_B, ort_session_C = S(onnx_model_A), S(onnx_model_B), S(onnx_model_C)
    ort_session_D, ort_session_E, ort_session_F = S(onnx_model_D), S(onnx_model_E), S(onnx_model_F)
    in_name_A0 = ort_session_A.get_inputs()[0].name
    out_name_A = [o.name for o in ort_session_A.get_outputs()]
    last_output_indices_A = len(out_name_A) - 1
    in_name_B0, out_name_B0 = ort_session_B.get_inputs()[0].name, ort_session_B.get_outputs()[0].name
    in_name_C = [i.name for i in ort_session_C.get_inputs()]
    out_name_C = [o.name for o in ort_session_C.get_outputs()]
    in_name_D = [i.name for i in ort_session_D.get_inputs()]
    out_name_D = [o.name for o in ort_session_D.get_outputs()]
    print(f"Usable Providers: {ort_session_E.get_providers()[0]}")
    model_E_dtype = np.float16 if "float16" in ort_session_E._inputs_meta[0].type else np.float32
    in_names_E = [i.name for i in ort_session_E.get_inputs()]
    out_name_E = [o.name for o in ort_session_E.get_outputs()]
    amount_of_outputs_E = len(out_name_E)
    num_layers = (amount_of_outputs_E - 3) // 2
    num_layers_2 = num_layers + num_layers
    last_input_indices_E, last_output_indices_E = len(in_names_E) - 1, amount_of_outputs_E - 1
    second_last_output_indices_E = amount_of_outputs_E - 2
    in_name_F = [i.name for i in ort_session_F.get_inputs()]
    out_name_F0 = ort_session_F.get_outputs()[0].name

    # ---- Inference_IndexTTS_ONNX.py:668-698 -------------------------------------------------------------------------------------
    if a.prompt:
        audio = audio_io.load_prompt(a.prompt, SAMPLE_RATE)
    else:
        t = np.arange(3 * SAMPLE_RATE)
        audio = (0.1 * 32767 * np.sin(2 * np.pi * 220 * t / SAMPLE_RATE)).astype(np.int16)
    audio = np.asarray(audio, dtype=np.int16).reshape(1, 1, -1)
    OV = onnxruntime.OrtValue.ortvalue_from_numpy
    audio = OV(audio, device_type, DEVICE_ID)
    init_gpt_ids = OV(np.array([[gcfg.start_mel_token]], dtype=np.int32), device_type, DEVICE_ID)
    init_gen_len = OV(np.array([0], dtype=np.int64), device_type, DEVICE_ID)
    init_ids_len_1 = OV(np.array([1], dtype=np.int64), device_type, DEVICE_ID)
    init_history_len = OV(np.array([0], dtype=np.int64), device_type, DEVICE_ID)
    init_attention_mask_0 = OV(np.array([0], dtype=np.int8), device_type, DEVICE_ID)
    init_attention_mask_1 = OV(np.array([1], dtype=np.int8), device_type, DEVICE_ID)
    m = ort_session_E._inputs_meta
    init_past_keys_E = OV(np.zeros((m[0].shape[0], m[0].shape[1], 0), dtype=model_E_dtype), device_type, DEVICE_ID)
    init_past_values_E = OV(np.zeros((m[num_layers].shape[0], 0, m[num_layers].shape[2]), dtype=model_E_dtype), device_type, DEVICE_ID)
    repeat_penality = OV(np.ones((1, m[num_layers_2 + 1].shape[1]), dtype=model_E_dtype), device_type, DEVICE_ID)
    split_pad = np.zeros((1, 1, int(SAMPLE_RATE * 0.2)), dtype=np.int16)
    input_feed_F = {}
    input_feed_E = {in_names_E[last_input_indices_E]: init_attention_mask_1, in_names_E[num_layers_2]: init_history_len,
                    in_names_E[num_layers_2 + 1]: repeat_penality}
    for i in range(num_layers):
        input_feed_E[in_names_E[i]] = init_past_keys_E
    for i in range(num_layers, num_layers_2):
        input_feed_E[in_names_E[i]] = init_past_values_E

    # ---- :700-800 -----------------------------------------------------------------------------------------------------------------
    start_time = time.time()
    all_outputs_A = ort_session_A.run_with_ort_values(out_name_A, {in_name_A0: audio})
    for i in range(last_output_indices_A):
        input_feed_F[in_name_F[i]] = all_outputs_A[i]
    text_tokens_list = tokenizer.tokenize(a.text)
    sentences = tokenizer.split_sentences(text_tokens_list)
    save_generated_wav = []
    total_tokens = 0
    for sent in sentences:
        split_text = "".join(sent).replace("▁", " ")
        print(f"Generate the Voice for '{split_text}'")
        text_tokens = tokenizer.convert_tokens_to_ids(sent)
        text_ids = OV(np.array([text_tokens], dtype=np.int32), device_type, DEVICE_ID)
        text_hidden_state = ort_session_B.run_with_ort_values([out_name_B0], {in_name_B0: text_ids})[0]
        gpt_hidden_state, gen_len = ort_session_C.run_with_ort_values(out_name_C, {in_name_C[0]: init_gpt_ids, in_name_C[1]: init_gen_len})
        gpt_hidden_state, concat_len = ort_session_D.run_with_ort_values(
            out_name_D, {in_name_D[0]: all_outputs_A[last_output_indices_A], in_name_D[1]: text_hidden_state, in_name_D[2]: gpt_hidden_state})
        generate_limit = MAX_GENERATE_LENGTH - onnxruntime.OrtValue.numpy(concat_len)
        input_feed_E[in_names_E[num_layers_2 + 2]] = concat_len
        save_last_hidden_state, save_max_logits_ids = [], []
        reset_penality = num_decode = 0
        decode_time = time.time()
        while num_decode < generate_limit:
            input_feed_E[in_names_E[num_layers_2 + 3]] = gpt_hidden_state
            all_outputs_E = ort_session_E.run_with_ort_values(out_name_E, input_feed_E)
            max_logit_ids = onnxruntime.OrtValue.numpy(all_outputs_E[last_output_indices_E])
            save_max_logits_ids.append(max_logit_ids)
            save_last_hidden_state.append(all_outputs_E[second_last_output_indices_E])
            num_decode += 1
            if max_logit_ids in STOP_TOKEN:
                break
            if num_decode < 2:
                input_feed_E[in_names_E[last_input_indices_E]] = init_attention_mask_0
                input_feed_E[in_names_E[num_layers_2 + 2]] = init_ids_len_1
            for i in range(second_last_output_indices_E):
                input_feed_E[in_names_E[i]] = all_outputs_E[i]
            repeat_penality = onnxruntime.OrtValue.numpy(repeat_penality)
            repeat_penality[:, max_logit_ids] = REPEAT_PENALITY
            if (num_decode > PENALITY_RANGE) and (save_max_logits_ids[reset_penality] != max_logit_ids):
                repeat_penality[:, save_max_logits_ids[reset_penality]] = 1.0
                reset_penality += 1
            repeat_penality = OV(repeat_penality, device_type, DEVICE_ID)
            input_feed_E[in_names_E[num_layers_2 + 1]] = repeat_penality
            gpt_hidden_state, gen_len = ort_session_C.run_with_ort_values(
                out_name_C, {in_name_C[0]: all_outputs_E[last_output_indices_E], in_name_C[1]: gen_len})
        print(f"Decode Speed: {num_decode / (time.time() - decode_time):.3f} tokens/s ({num_decode} tokens)")
        total_tokens += num_decode
        for i in range(num_decode):
            save_last_hidden_state[i] = onnxruntime.OrtValue.numpy(save_last_hidden_state[i])
        if num_decode >= 3:                      # graph F needs T_codes >= 3 ((T_codes - 2) * hop + 30 samples)
            input_feed_F[in_name_F[last_output_indices_A]] = OV(np.concatenate(save_last_hidden_state, axis=0), device_type, DEVICE_ID)
            generated_wav = ort_session_F.run_with_ort_values([out_name_F0], input_feed_F)[0]
            save_generated_wav.append(np.concatenate([onnxruntime.OrtValue.numpy(generated_wav), split_pad], axis=-1))
        # Init (the next sentence starts from empty caches)
        input_feed_E[in_names_E[last_input_indices_E]] = init_attention_mask_1
        input_feed_E[in_names_E[num_layers_2]] = init_history_len
        for i in range(num_layers):
            input_feed_E[in_names_E[i]] = init_past_keys_E
        for i in range(num_layers, num_layers_2):
            input_feed_E[in_names_E[i]] = init_past_values_E
        # (like the driver, the repeat penalty vector is NOT reset between sentences: :793-800 re-arm the caches only)
    dt = time.time() - start_time
    wav = np.concatenate(save_generated_wav, axis=-1) if save_generated_wav else split_pad
    audio_io.write_wavex(a.out, wav.reshape(-1), SAMPLE_RATE)
    secs = wav.size / SAMPLE_RATE
    print(f"{a.out}: {len(sentences)} sentence(s), {total_tokens} mel codes, {secs:.2f} s of audio in {dt:.3f} s (RTF {dt / max(secs, 1e-9):.4f})")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
