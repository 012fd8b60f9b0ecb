#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native TTS hot path.

    python bench.py --gpus N --steps K --warmup W [--workload bigvgan|f5]

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=48)
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
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg = BigVGANConfig()
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
    mel = torch.from_numpy(W.synth_normal(100 + rank, "mel", (B, cfg.num_mels, F), std=2.0, mean=-2.0)
                           .clip(-11.5, 2.5)).to(dev)
    out = torch.empty((B, 1, voc.out_len(F)), dtype=torch.int16, device=dev)
    audio_s = B * voc.out_len(F) / cfg.sampling_rate

    for _ in range(args.warmup):
        voc.run_torch(mel, out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    _lib.prof_reset()
    _lib.prof_enable(["conv_gemm"])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        voc.run_torch(mel, out)
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
            "config": {"workload": f"BigVGAN-v2 24khz_100band_256x {args.dtype} vocoder, mel ({B},100,{F}) per GPU "
                                   f"(BASELINE configs[1])",
                       "batch_per_gpu": B, "frames": F, "audio_seconds_per_step_per_gpu": audio_s,
                       "rtf": dt / args.steps / audio_s, "weights": "synthetic seeded (112.4 M params)",
                       "weight_bcast_ms": bcast_ms,
                       "whole_forward_algorithmic_GB": bigvgan_algorithmic_bytes(cfg, B, F, esz) / 1e9,
                       "whole_forward_achieved_GBps": bigvgan_algorithmic_bytes(cfg, B, F, esz) / (dt / args.steps) / 1e9},
            "roofline": {"bound": "hbm", "kernel": "conv_gemm_kernel (implicit-GEMM Conv1d/ConvTranspose1d family)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "launches_per_step": prof["launches"] / args.steps,
                         "avg_launch_ms": k_ms, "family_ms_per_step": prof["ms"] / args.steps,
                         "tflops": prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            if state is None:
                state = W.synth_state(spec, 9527)
            line["cpu_baseline"] = cpu_baseline_bigvgan(cfg, state, args.cpu_frames)
        print(json.dumps(line), flush=True)
    voc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
