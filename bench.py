#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native TTS hot path.

    python bench.py --gpus N --steps K --warmup W [--workload bigvgan|f5|indextts_f|indextts]

Workload (BASELINE.json configs[1]): BigVGAN-v2 24khz_100band_256x, fp16 HIP vocoder, mel (8,100,512)
per GPU, synthetic seeded weights and mel already resident in HBM.  One step = one vocoder pass over
the batch.  Metric = generated audio-seconds per wall second (whole job, all ranks); RTF = its
inverse is reported in `config`.

Multi-GPU: one process per GPU (torchrun env), utterance batches are independent => weak scaling,
no data-path collective; the packed weight blob is built on rank 0 and broadcast over RCCL.

Adds `roofline` (dominant kernel family = the implicit-GEMM conv stack, HIP events on the engine's
own stream, algorithmic bytes per SURVEY.md §8d) and `cpu_baseline` (numpy oracle on a bounded
sample, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TF = 2500.0        # dense bf16/f16
MFMA_F32_PEAK_TF = 157.3


def bigvgan_algorithmic_bytes(cfg, B: int, F: int, esz: int) -> float:
    """Layer-granular HBM bytes of one forward: per stage 85*E_i + E_{i-1} + E_i,
    conv_pre, post activation, conv_post, weights once."""
    total = 0.0
    T = F
    e_prev = B * cfg.upsample_initial_channel * T * esz
    total += B * cfg.num_mels * F * 4 + e_prev                      # conv_pre: read mel (fp32), write E_pre
    for i, u in enumerate(cfg.upsample_rates):
        T *= u
        e = B * cfg.stage_channels(i) * T * esz
        total += e_prev + e + 85.0 * e
        e_prev = e
    total += 2 * e_prev                                             # post AA activation
    total += e_prev + B * (T + 30) * 2                              # conv_post -> int16
    from mi355tts.weights import bigvgan_spec
    total += sum(int(np.prod(s)) for _, s, _ in bigvgan_spec(cfg)) * esz
    return total


def cpu_baseline_bigvgan(cfg, state, frames: int):
    """numpy oracle (kind 'port') on a bounded sample of the same workload."""
    from oracle import bigvgan_np as O
    from mi355tts.weights import synth_normal
    mel = synth_normal(11, "mel", (1, cfg.num_mels, frames), std=2.0, mean=-2.0).clip(-11.5, 2.5)
    t0 = time.perf_counter()
    w = O.bigvgan_int16(cfg, state, mel)
    dt = time.perf_counter() - t0
    secs = w.shape[-1] / cfg.sampling_rate
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": secs / dt, "unit": "audio_seconds_per_second", "cores": int(cores), "kind": "port",
            "sample": f"numpy oracle, BigVGAN-v2 fp32, mel (1,{cfg.num_mels},{frames}) = {secs:.2f} s audio in {dt:.1f} s"}


def f5_flops_per_eval(cfg, N: int) -> float:
    """Algorithmic FLOPs of one DiT CFG evaluation: tokens = 2N,
    MAC/token = depth*(4d^2 + 2*d*ff + 2*N*d) + (cat*d + 2*d*(d/g)*k + d*mel)."""
    d, ff = cfg.dim, cfg.ff_dim
    mac = cfg.depth * (4 * d * d + 2 * d * ff + 2 * N * d) + ((2 * cfg.mel_dim + cfg.text_dim) * d +
                                                               2 * d * (d // cfg.pos_conv_groups) * cfg.pos_conv_kernel + d * cfg.mel_dim)
    return 2.0 * (2 * N) * mac


def f5_synthetic_inputs(cfg, U: int, rank: int):
    from mi355tts import weights as W
    return W.f5_synthetic_inputs(cfg, U, rank)


def cpu_baseline_f5(cfg, raw_state, audio, ids, N, noise):
    """numpy oracle (kind 'port'): preprocess + ONE of the 31 DiT evaluations + decode, extrapolated to the
    full 31-evaluation utterance (every evaluation costs the same)."""
    from oracle import f5_np as O
    from mi355tts import weights as W
    st = W.fold_f5(cfg, raw_state)
    t0 = time.perf_counter()
    pre = O.preprocess(cfg, st, audio, ids, N, noise)
    tables = O.time_tables(cfg, st)
    t1 = time.perf_counter()
    x = O.transformer_step(cfg, st, tables, pre["noise"], pre, 0)
    t2 = time.perf_counter()
    w = O.decode(cfg, st, x, pre["ref_signal_len"])
    t3 = time.perf_counter()
    total = (t1 - t0) + (t2 - t1) * (cfg.nfe_step - 1) + (t3 - t2)
    secs = w.shape[-1] / cfg.sample_rate
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": secs / total, "unit": "audio_seconds_per_second", "cores": int(cores), "kind": "port",
            "sample": f"numpy oracle fp32: preprocess {t1 - t0:.1f} s + 1 of {cfg.nfe_step - 1} DiT evaluations "
                      f"{t2 - t1:.1f} s (x{cfg.nfe_step - 1} extrapolated) + decode {t3 - t2:.1f} s for one {secs:.2f} s utterance"}


def run_f5(args, world, rank, local, dev, dist, torch):
    from mi355tts.config import F5Config
    from mi355tts import weights as W
    from mi355tts import _lib
    from mi355tts.f5 import F5Engine
    cfg = F5Config()
    spec = W.f5_spec(cfg)
    raw = None
    nparam = sum(int(np.prod(sh)) for _, sh, _ in W.f5_packed_spec(cfg))
    if rank == 0:
        raw = W.synth_state(spec, 9527)
        blob_t = torch.from_numpy(W.pack_f5(cfg, raw)).to(dev)
    else:
        blob_t = torch.empty(nparam, dtype=torch.float32, device=dev)
    bcast_ms = 0.0
    if world > 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(blob_t, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    eng = F5Engine(cfg, blob=blob_t.cpu().numpy(), dtype=args.dtype, device=local)
    del blob_t
    U = args.batch
    audio, ids, N, noise = f5_synthetic_inputs(cfg, U, rank)
    R = audio.shape[1] // cfg.hop_length + 1
    t_audio, t_ids, t_noise = torch.from_numpy(audio).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(noise).to(dev)
    out = torch.empty((U, 1, (N - R - 1) * cfg.hop_length), dtype=torch.int16, device=dev)
    audio_s = U * out.shape[-1] / cfg.sample_rate
    # the engine runs a shape eagerly once, captures the 31-step loop into a hipGraph on its second use and replays
    # it afterwards: at least two untimed calls so the timed region is steady state
    for _ in range(max(args.warmup, 2)):
        eng.synthesize_torch(t_audio, t_ids, N, noise=t_noise, out=out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.synthesize_torch(t_audio, t_ids, N, noise=t_noise, out=out)
    barrier()
    dt = time.perf_counter() - t0
    # roofline leg: one more pass with HIP events around every GEMM / attention launch (the events force the eager,
    # un-graphed launch path, so this pass is timed separately and is NOT part of `value`)
    _lib.prof_reset()
    _lib.prof_enable(["conv_gemm", "attn"])
    eng.synthesize_torch(t_audio, t_ids, N, noise=t_noise, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(())
    pg, pa = _lib.prof_get("conv_gemm"), _lib.prof_get("attn")
    prof_steps = 1
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        eng.close()
        return
    peak = MFMA_F32_PEAK_TF if args.dtype == "f32" else MFMA_F16_PEAK_TF
    achieved = pg["flops"] / (pg["ms"] * 1e-3) / 1e12 if pg["ms"] > 0 else 0.0
    value = world * audio_s * args.steps / dt
    alg_flops = f5_flops_per_eval(cfg, N) * (cfg.nfe_step - 1) * U
    line = {
        "metric": "audio_seconds_per_second", "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"F5-TTS {args.dtype} NFE=32 (31 DiT evaluations, CFG batch 2) + Vocos/ISTFT end to end, "
                               f"{U} utterance(s) per GPU, 6 s ref audio, N={N} frames (BASELINE configs[2]/[3])",
                   "utterances_per_gpu": U, "frames": N, "audio_seconds_per_step_per_gpu": audio_s,
                   "rtf": dt / args.steps / audio_s, "weights": "synthetic seeded (337 M DiT + 13.5 M Vocos params)",
                   "weight_bcast_ms": bcast_ms, "end_to_end_TFLOPs_per_step": alg_flops / 1e12,
                   "end_to_end_TFLOP_per_s": alg_flops / (dt / args.steps) / 1e12},
        "roofline": {"bound": "mfma", "kernel": "conv_gemm_kernel (DiT linear layers, implicit-GEMM MFMA)", "achieved": achieved,
                     "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                     "launches_per_step": pg["launches"] / prof_steps, "avg_launch_ms": pg["ms"] / max(pg["launches"], 1),
                     "family_ms_per_step": pg["ms"] / prof_steps,
                     "attn_ms_per_step": pa["ms"] / prof_steps,
                     "note": "event-timed in a separate eager pass; the timed region replays a hipGraph",
                     "attn_tflops": pa["flops"] / (pa["ms"] * 1e-3) / 1e12 if pa["ms"] > 0 else 0.0},
    }
    if world == 1 and not args.no_cpu_baseline:
        if raw is None:
            raw = W.synth_state(spec, 9527)
        line["cpu_baseline"] = cpu_baseline_f5(cfg, raw, audio[0], ids[0], N, noise[0])
    print(json.dumps(line), flush=True)
    eng.close()


def run_indextts(args, world, rank, local, dev, dist, torch):
    """BASELINE configs[4] minus graph A: one sentence = GPT-2 prompt pass + greedy mel-code decode (graphs B/C/D/E and
    the loop, Inference_IndexTTS_ONNX.py:723-783) + the speaker-conditioned BigVGAN (graph F, :787).  conds_latent and
    the vocoder conditioning vectors (graph A's outputs) are synthetic."""
    from mi355tts.config import IndexGPTConfig, BigVGANConfig
    from mi355tts import weights as W
    from mi355tts import _lib
    from mi355tts.indextts import IndexGPT
    from mi355tts.bigvgan import BigVGANVocoder
    gcfg, vcfg = IndexGPTConfig(), BigVGANConfig.indextts()
    gspec, vspec = W.gpt_spec(gcfg), W.bigvgan_spec(vcfg)
    ng = sum(int(np.prod(sh)) for _, sh, _ in gspec)
    nv = sum(int(np.prod(sh)) for _, sh, _ in vspec)
    graw = None
    if rank == 0:
        graw = W.synth_state(gspec, 9527, fast=True)
        blob_t = torch.from_numpy(np.concatenate([W.pack_gpt(gcfg, graw),
                                                  W.pack_bigvgan(vcfg, W.synth_state(vspec, 9527, fast=True))])).to(dev)
    else:
        blob_t = torch.empty(ng + nv, dtype=torch.float32, device=dev)
    bcast_ms = 0.0
    if world > 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(blob_t, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    blob = blob_t.cpu().numpy()
    del blob_t
    NB = max(1, args.batch)
    gcfg.max_batch = NB
    gpt = IndexGPT(gcfg, blob=blob[:ng], dtype=args.dtype, device=local)
    voc = BigVGANVocoder(vcfg, blob=blob[ng:], dtype=args.dtype, device=local)
    del blob
    n_text, n_cond, n_tok = 30, 32, args.tokens
    text = (np.arange(n_text, dtype=np.int32) * 37 + 11 * rank) % (gcfg.text_tokens - 2) + 2
    conds = W.synth_normal_fast(100 + rank, "conds_latent", (1, n_cond, gcfg.hidden), std=0.5)
    mel_h, _ = gpt.mel_embed(gcfg.start_mel_token, 0)
    prompt_np, concat_len = gpt.concat(conds, gpt.text_embed(text), mel_h)
    P = int(concat_len[0])
    prompt = torch.from_numpy(prompt_np[0]).to(dev)
    toks = torch.zeros((n_tok,), dtype=torch.int32, device=dev)
    hid = torch.zeros((n_tok, gcfg.hidden), dtype=torch.float32, device=dev)
    ncond = vcfg.upsample_initial_channel + sum(vcfg.stage_channels(i) for i in range(vcfg.num_upsamples))
    vconds = torch.from_numpy(W.synth_normal_fast(100 + rank, "conds", (ncond,), std=0.2)).to(dev)
    wav = torch.empty((1, 1, (n_tok - 2) * vcfg.hop + 30), dtype=torch.int16, device=dev)
    audio_s = NB * wav.shape[-1] / vcfg.sampling_rate
    if NB > 1:      # NB sentences per step: different texts, one shared weight stream per decode step
        ps = []
        for b in range(NB):
            tb = (np.arange(n_text, dtype=np.int32) * 37 + 11 * rank + 101 * b) % (gcfg.text_tokens - 2) + 2
            ps.append(gpt.concat(conds, gpt.text_embed(tb), mel_h)[0][0])
        prompts_cat = torch.from_numpy(np.concatenate(ps, axis=0)).to(dev)
        toks_b = torch.zeros((NB, n_tok), dtype=torch.int32, device=dev)
        hid_b = torch.zeros((NB, n_tok, gcfg.hidden), dtype=torch.float32, device=dev)

    def gpt_leg():
        if NB == 1:
            n = gpt.generate_torch(prompt, n_tok, toks, hid, stop_tokens=[])
            assert n == n_tok
        else:
            n = gpt.generate_batch_torch(prompts_cat, [P] * NB, [n_tok] * NB, toks_b, hid_b, stop_tokens=[])
            assert (n == n_tok).all()

    def step():
        # stop_tokens=[]: a fixed amount of work per sentence (random weights never emit the stop code on cue)
        gpt_leg()
        for b in range(NB):
            voc.run_latent_torch(hid if NB == 1 else hid_b[b], vconds, wav)

    for _ in range(max(args.warmup, 2)):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # GPT leg alone (same state), then the roofline leg: one eager pass with HIP events around every GEMV / GEMM launch
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gpt_leg()
    torch.cuda.synchronize()
    gpt_s = time.perf_counter() - t1
    _lib.prof_reset()
    _lib.prof_enable(["conv_gemm", "attn"])
    gpt_leg()
    torch.cuda.synchronize()
    _lib.prof_enable(())
    pg, pa = _lib.prof_get("conv_gemm"), _lib.prof_get("attn")
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        gpt.close(); voc.close()
        return
    esz = 4 if args.dtype == "f32" else 2
    achieved = pg["bytes"] / (pg["ms"] * 1e-3) / 1e9 if pg["ms"] > 0 else 0.0
    wbytes = (gcfg.layers * 12 * gcfg.hidden * gcfg.hidden + gcfg.mel_codes * gcfg.hidden) * esz
    line = {
        "metric": "audio_seconds_per_second", "value": world * audio_s * args.steps / dt, "unit": "audio-s/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"IndexTTS-1.5 {args.dtype}: GPT-2 (24 x 1280, 20 heads) prompt pass of {P} rows + greedy decode of "
                               f"{n_tok} mel codes + BigVGAN graph F, {NB} sentence(s) per GPU per step (BASELINE configs[4] "
                               f"without graph A: conds_latent / speaker conditioning synthetic)",
                   "tokens": n_tok, "prompt_rows": P, "audio_seconds_per_step_per_gpu": audio_s,
                   "sentences_per_gpu": NB,
                   "rtf": dt / args.steps / audio_s, "gpt_leg_ms": gpt_s * 1e3, "decode_tokens_per_s": NB * n_tok / gpt_s,
                   "weight_bytes_streamed_per_token_GB": wbytes / 1e9,
                   "decode_weight_stream_GBps": wbytes * n_tok / gpt_s / 1e9,
                   "weights": "synthetic seeded (510 M GPT + vocoder)", "weight_bcast_ms": bcast_ms},
        "roofline": {"bound": "hbm", "kernel": "gemv_kernel (decode-step linear layers: weights streamed once per token)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "launches_per_step": pg["launches"], "avg_launch_ms": pg["ms"] / max(pg["launches"], 1),
                     "family_ms_per_step": pg["ms"], "attn_ms_per_step": pa["ms"],
                     "note": "event-timed in a separate eager pass (the timed region replays a hipGraph per token); the "
                             "family also holds the prompt pass's 4 x 24 MFMA GEMM launches"},
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle import gpt_np as O
        n_cpu = 6
        folds = [O.fold_layer(gcfg, graw, i) for i in range(gcfg.layers)]
        keys = [np.zeros((gcfg.heads, 64, 0), np.float32)] * gcfg.layers
        vals = [np.zeros((gcfg.heads, 0, 64), np.float32)] * gcfg.layers
        pen = np.ones((1, gcfg.mel_codes), np.float32)
        t2 = time.perf_counter()
        keys, vals, kvl, last, tok, _ = O.graph_e(gcfg, graw, keys, vals, 0, pen, P, prompt_np, 1, folds)
        gl = np.array([1])
        for _ in range(n_cpu - 1):
            hs, gl = O.graph_c(gcfg, graw, tok, gl)
            keys, vals, kvl, last, tok, _ = O.graph_e(gcfg, graw, keys, vals, int(kvl[0]), pen, 1, hs, 0, folds)
        cpu_s = time.perf_counter() - t2
        line["cpu_baseline"] = {"value": n_cpu * vcfg.hop / vcfg.sampling_rate / cpu_s, "unit": "audio-s/s",
                                "cores": os.cpu_count(), "kind": "port",
                                "sample": f"numpy oracle, fp32: prompt pass of {P} rows + {n_cpu - 1} decode steps of the "
                                          f"same GPT (graph E only, no vocoder leg), {cpu_s:.1f} s; audio = tokens x 1024 / 24 kHz"}
    print(json.dumps(line), flush=True)
    gpt.close(); voc.close()


def pmc_traffic_per_launch(args, ixf):
    """HBM-side bytes per launch of the conv family from the committed PMC passes (profiles/r1/, same command as this
    run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, FETCH doubled per MI355X_MICROARCH.md; tools/
    pmc_traffic.py).  Counters cannot be collected from inside the timed run, so the figure is only reported for the
    configuration those passes were taken on (the default one); otherwise null."""
    if ixf or args.dtype != "f16" or args.batch != 8 or args.frames != 512:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r1", "final_bigvgan_pmc_hbm_traffic.json")) as f:
            t = json.load(f)["_conv_family"]
        return t["traffic_GB_per_forward"] * 1e9 / t["launches_per_forward"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=256, help="indextts: mel codes decoded per sentence")
    ap.add_argument("--workload", default="bigvgan", choices=["bigvgan", "f5", "indextts_f", "indextts"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="mel batch (bigvgan, default 8) / utterances (f5, default 1) per GPU")
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--dtype", default=None, help="bigvgan: f16 (default) | f32 | bf16 ; f5: bf16 (default) | f32 | f16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=128)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mi355tts.config import BigVGANConfig
    from mi355tts import weights as W
    from mi355tts import _lib
    from mi355tts.bigvgan import BigVGANVocoder

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # MI355TTS_BENCH_BACKEND=gloo + MI355TTS_BENCH_ONE_GPU=1: exercise the multi-rank code path on a 1-GPU box
        dist.init_process_group(os.environ.get("MI355TTS_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if os.environ.get("MI355TTS_BENCH_ONE_GPU") == "1":
        local = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.steps is None:
        args.steps = 20 if args.workload == "bigvgan" else 3
    if args.batch is None:
        args.batch = 8 if args.workload == "bigvgan" else 1
    if args.dtype is None:
        args.dtype = "f16" if args.workload == "bigvgan" else "bf16"
    if args.workload == "indextts":
        if args.dtype == "bf16" and "--dtype" not in " ".join(sys.argv):
            args.dtype = "f16"
        run_indextts(args, world, rank, local, dev, dist, torch)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload == "f5":
        run_f5(args, world, rank, local, dev, dist, torch)
        if world > 1:
            dist.destroy_process_group()
        return

    ixf = args.workload == "indextts_f"        # BASELINE configs[4] vocoder leg: IndexTTS graph F, T_codes = 128
    cfg = BigVGANConfig.indextts() if ixf else BigVGANConfig()
    if ixf:
        args.batch, args.frames = 1, 126
    spec = W.bigvgan_spec(cfg)
    # weights: rank 0 packs, everybody else receives the blob over RCCL (xGMI)
    nparam = sum(int(np.prod(s)) for _, s, _ in spec)
    state = None
    if rank == 0:
        state = W.synth_state(spec, 9527)
        blob_t = torch.from_numpy(W.pack_bigvgan(cfg, state)).to(dev)
    else:
        blob_t = torch.empty(nparam, dtype=torch.float32, device=dev)
    bcast_ms = 0.0
    if world > 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(blob_t, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    voc = BigVGANVocoder(cfg, blob=blob_t.cpu().numpy(), dtype=args.dtype, device=local)
    del blob_t

    B, F = args.batch, args.frames
    out = torch.empty((B, 1, voc.out_len(F)), dtype=torch.int16, device=dev)
    audio_s = B * voc.out_len(F) / cfg.sampling_rate
    if ixf:
        latent = torch.from_numpy(W.synth_normal(100 + rank, "latent", (F + 2, cfg.num_mels), std=1.5, mean=0.3)).to(dev)
        ncond = cfg.upsample_initial_channel + sum(cfg.stage_channels(i) for i in range(cfg.num_upsamples))
        conds = torch.from_numpy(W.synth_normal(100 + rank, "conds", (ncond,), std=0.2)).to(dev)
        step = lambda: voc.run_latent_torch(latent, conds, out)
    else:
        mel = torch.from_numpy(W.bigvgan_synthetic_mel(cfg, B, F, rank)).to(dev)
        step = lambda: voc.run_torch(mel, out)

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    _lib.prof_reset()
    _lib.prof_enable(["conv_gemm"])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(())
    prof = _lib.prof_get("conv_gemm")
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        esz = 4 if args.dtype == "f32" else 2
        value = world * audio_s * args.steps / dt
        k_ms = prof["ms"] / max(prof["launches"], 1)
        achieved = prof["bytes"] / (prof["ms"] * 1e-3) / 1e9 if prof["ms"] > 0 else 0.0
        line = {
            "metric": "audio_seconds_per_second", "value": value, "unit": "audio-s/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": (f"IndexTTS graph F (speaker-conditioned BigVGAN, 1024x) {args.dtype}, T_codes = {F + 2} "
                                    f"(BASELINE configs[4] vocoder leg)") if ixf else
                                   (f"BigVGAN-v2 24khz_100band_256x {args.dtype} vocoder, mel ({B},100,{F}) per GPU "
                                    f"(BASELINE configs[1])"),
                       "batch_per_gpu": B, "frames": F, "audio_seconds_per_step_per_gpu": audio_s,
                       "rtf": dt / args.steps / audio_s, "weights": "synthetic seeded (112.4 M params)",
                       "weight_bcast_ms": bcast_ms,
                       "whole_forward_algorithmic_GB": bigvgan_algorithmic_bytes(cfg, B, F, esz) / 1e9,
                       "whole_forward_achieved_GBps": bigvgan_algorithmic_bytes(cfg, B, F, esz) / (dt / args.steps) / 1e9},
            "roofline": {"bound": "hbm", "kernel": "conv_gemm_kernel (implicit-GEMM Conv1d/ConvTranspose1d family)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic_per_launch(args, ixf), "launches_per_step": prof["launches"] / args.steps,
                         "algorithmic_bytes_per_launch": prof["bytes"] / max(prof["launches"], 1),
                         "avg_launch_ms": k_ms, "family_ms_per_step": prof["ms"] / args.steps,
                         "tflops": prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0,
                         "mfma_peak_tflops": MFMA_F32_PEAK_TF if args.dtype == "f32" else MFMA_F16_PEAK_TF,
                         "mfma_frac": (prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0) /
                                      (MFMA_F32_PEAK_TF if args.dtype == "f32" else MFMA_F16_PEAK_TF),
                         "note": "family = every implicit-GEMM launch of the forward (stages 0-2 are MFMA / LDS-fill bound, "
                                 "stages 3-5 + fused AA are HBM bound); bytes are the layer-granular algorithmic count"},
        }
        if world == 1 and not args.no_cpu_baseline and not ixf:
            if state is None:
                state = W.synth_state(spec, 9527)
            line["cpu_baseline"] = cpu_baseline_bigvgan(cfg, state, args.cpu_frames)
        print(json.dumps(line), flush=True)
    voc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
