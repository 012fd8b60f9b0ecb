#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native TTS hot path.

    python bench.py --gpus N --steps K --warmup W [--workload bigvgan|f5|indextts_f|indextts]

Default workload = BASELINE.json's metric: F5-TTS NFE=32 end to end (preprocess -> 31 DiT evaluations with CFG ->
Vocos + ISTFT -> int16), synthetic seeded weights, 6 s reference audio + ~15-word texts (N = 1126 frames), inputs
resident in HBM.  One GPU: configs[2] (fp32, one utterance — the precision the 1e-3 RMS parity gate is stated at);
`--gpus N > 1`: the configs[3] shard (bf16, 8 utterances per GPU).  One step = one batch of utterances through the whole
path.  Metric = generated audio-seconds per wall second over all ranks; RTF (= its inverse per GPU) is in `config`.

OUTPUT CONTRACT: stdout carries exactly ONE line — a compact (< 4 KB) JSON object with metric / value / ms_per_step /
config / roofline / cpu_baseline / secondary_ms (one number per secondary workload).  Everything else (per-kernel tables,
instantiations, PMC detail, notes, the secondary blocks in full) is written to `bench_detail.json` (next to this file, and
under gpurun_out/ when that directory exists) and echoed on stderr.  This file holds the headline measurement only; the
secondary workloads, the PMC passes and the other `--workload`s live in bench_detail.py.

Multi-GPU: one process per GPU (torchrun env), utterances are independent => weak scaling, no data-path collective;
the packed weight blob is built on rank 0, broadcast over RCCL and consumed by the engine from device memory.

`roofline` = ONE kernel (the one with the largest event-timed total): its algorithmic flops (2*M*N*K) per launch / its
average launch duration (HIP events on the engine's own stream) / the matching gfx950 peak.
`cpu_baseline` = the numpy (OpenBLAS-threaded) oracle on a bounded sample of the same workload, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

from bench_common import (ROOT, USER_OPTIONS, MFMA_F16_PEAK_TF, MFMA_F32_PEAK_TF, f5_flops_per_eval, merge_instantiations,
                          dominant_kernel_roofline, bcast_device_blob, per_rank_times, max_over_ranks, f5_workload_name, emit, claim_stdout)


def cpu_baseline_f5_run(cfg, st, audio, ids, N, noise, threads: int, evals):
    """The numpy oracle (kind 'port') through the reference's bracket — preprocess, `evals` of the NFE - 1 DiT evaluations (None: ALL
    of them, the sampling loop as the reference runs it), decode — on `threads` threads (BLAS and the oracle's own row / head
    parallelism alike).  Returns measured seconds per stage; nothing is scaled here."""
    from oracle import f5_np as O
    O.set_threads(threads)
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(limits=threads, user_api="blas")
    except Exception:
        import contextlib
        lim = contextlib.nullcontext()
    nfe = cfg.nfe_step - 1
    k_run = nfe if evals is None else min(int(evals), nfe)
    with lim:
        t0 = time.perf_counter()
        pre = O.preprocess(cfg, st, audio, ids, N, noise)
        tables = O.time_tables(cfg, st)
        t1 = time.perf_counter()
        x = pre["noise"]
        per_eval = []
        for k in range(k_run):
            te = time.perf_counter()
            x = O.transformer_step(cfg, st, tables, x, pre, k)
            per_eval.append(time.perf_counter() - te)
        t2 = time.perf_counter()
        w = O.decode(cfg, st, x, pre["ref_signal_len"])
        t3 = time.perf_counter()
    O.set_threads(1)
    return {"threads": threads, "evaluations_run": k_run, "evaluations_of_the_workload": nfe, "preprocess_s": t1 - t0,
            "evaluations_s": per_eval, "decode_s": t3 - t2, "audio_s": w.shape[-1] / cfg.sample_rate}


def cpu_baseline_f5(cfg, raw_state, audio, ids, N, noise, full: bool = False, threads=None):
    """`cpu_baseline`: the numpy oracle (kind 'port', im2col + OpenBLAS sgemm) on the SAME utterance at the reference driver's own
    thread setting (MAX_THREADS = 8, F5-TTS-ONNX-Inference.py:36) — ONE thread setting (a 32-thread leg measured 2.3x slower on the
    256-thread GPU-box host: oversubscription, not a baseline; `--cpu-threads T` runs another setting on request).
    Default (round 6): preprocess + ALL 31 DiT evaluations + decode, every stage measured, nothing scaled
    (`all_evaluations_measured: true`; ~56 s on the GPU box's host, profiles/r4/cpu_baseline_full.json).  MI355TTS_CPU_EVALS=k
    bounds the leg to k evaluations (`value` then counts the ones not run at the mean measured evaluation and
    `all_evaluations_measured: false` says so); `--cpu-baseline-full` ignores that bound."""
    from mi355tts import weights as W
    st = W.fold_f5(cfg, raw_state)
    nproc = os.cpu_count() or 1
    T = int(threads or os.environ.get("MI355TTS_CPU_THREADS", min(8, nproc)))
    evals = None if (full or "MI355TTS_CPU_EVALS" not in os.environ) else int(os.environ["MI355TTS_CPU_EVALS"])
    r = cpu_baseline_f5_run(cfg, st, audio, ids, N, noise, T, evals)
    mean_eval = sum(r["evaluations_s"]) / len(r["evaluations_s"])
    nfe = r["evaluations_of_the_workload"]
    total = r["preprocess_s"] + sum(r["evaluations_s"]) + mean_eval * (nfe - r["evaluations_run"]) + r["decode_s"]
    return {"value": r["audio_s"] / total, "unit": "audio-s/s", "kind": "port", "cores": T, "host_nproc": nproc,
            "evaluations_run": r["evaluations_run"], "evaluations_of_the_workload": nfe,
            "all_evaluations_measured": r["evaluations_run"] == nfe, "utterance_s": total, "measured_cpu_s": r["preprocess_s"] + sum(r["evaluations_s"]) + r["decode_s"],
            "sample": (f"numpy oracle fp32, one {r['audio_s']:.2f} s utterance at {T} threads: preprocess + {r['evaluations_run']} of {nfe} DiT "
                       f"evaluations + decode, each measured" + ("" if r["evaluations_run"] == nfe else "; the rest counted at the mean measured evaluation")),
            "stages": r}


class F5Bench:
    """F5-TTS NFE=32 end to end (BASELINE configs[2] / [3]): the bracket of F5-TTS-ONNX-Inference.py:246-312 —
    preprocess (graph A) -> 31 DiT evaluations with CFG (graph B x 31) -> Vocos + ISTFT -> int16 (graph C) — with the
    audio / text ids / injected noise already resident in HBM and the int16 waveform left in HBM."""

    def __init__(self, torch, dist, world, rank, local, dev):
        from mi355tts.config import F5Config
        from mi355tts import weights as W
        self.torch, self.dist, self.world, self.rank, self.local, self.dev = torch, dist, world, rank, local, dev
        # MI355TTS_BENCH_SMALL=1: reduced model + 1 s of audio, for the 2-rank plumbing test on a one-GPU box (its line
        # says so in config.workload and is not a benchmark result)
        self.small = os.environ.get("MI355TTS_BENCH_SMALL") == "1"
        self.cfg = F5Config.small() if self.small else F5Config()
        self.cfg_over = {}            # F5Config fields set from the command line (--f32-arithmetic, --no-adaln-fold)
        self.L = 24000 if self.small else 144000
        self.W = W
        self.raw = None
        nparam = sum(int(np.prod(sh)) for _, sh, _ in W.f5_packed_spec(self.cfg))
        if rank == 0:
            self.raw = W.synth_state(W.f5_spec(self.cfg), 9527)
            self.blob_t = torch.from_numpy(W.pack_f5(self.cfg, self.raw)).to(dev)
        else:
            self.blob_t = torch.empty(nparam, dtype=torch.float32, device=dev)
        self.bcast_ms = 0.0
        self.dump_dir = None
        # one process per GPU: fail loudly if two ranks resolved to the same physical device (the plumbing test on a one-GPU box says so)
        from mi355tts.shard import assert_one_device_per_rank
        self.rank_devices = [f"{h}:{b}" for h, b in assert_one_device_per_rank(local, allow_shared=os.environ.get("MI355TTS_BENCH_ONE_GPU") == "1")]
        if world > 1 or dist.is_initialized():      # the one collective of the path: weights rank 0 -> all, RCCL over xGMI (one rank: --force-collective)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.blob_t = bcast_device_blob(torch, dist, self.blob_t, force=True)
            torch.cuda.synchronize()
            self.bcast_ms = (time.perf_counter() - t0) * 1e3

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def measure(self, dtype: str, U: int, steps: int, warmup: int, f32_arithmetic=None):
        import dataclasses
        from mi355tts import _lib
        from mi355tts.f5 import F5Engine
        torch, dev = self.torch, self.dev
        over = dict(self.cfg_over)
        if f32_arithmetic is not None:
            over["f32_arithmetic"] = f32_arithmetic
        cfg = dataclasses.replace(self.cfg, **over)
        eng = F5Engine(cfg, blob_device=self.blob_t, dtype=dtype, device=self.local)
        eng_info = eng.info()
        # the job's utterance list: world * U utterances (configs[3]: 64 on 8 GPUs, seeds 9527 ..), contiguous slices per rank
        from mi355tts.shard import shard_range
        lo, hi = shard_range(self.world * U, self.world, self.rank)
        assert hi - lo == U
        audio, ids, N, noise = self.W.f5_synthetic_inputs(cfg, U, self.rank, L=self.L, first=lo)
        R = cfg.ref_frames(audio.shape[1])
        t_audio, t_ids, t_noise = torch.from_numpy(audio).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(noise).to(dev)
        out = torch.empty((U, 1, (N - R - 1) * cfg.hop_length), dtype=torch.int16, device=dev)
        audio_s = U * out.shape[-1] / cfg.sample_rate
        # the engine runs a shape eagerly once, captures the 31-step loop into a hipGraph on its second use and replays
        # it afterwards: at least two untimed calls so the timed region is steady state
        for _ in range(max(warmup, 2)):
            eng.synthesize_torch(t_audio, t_ids, N, noise=t_noise, out=out)
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.synthesize_torch(t_audio, t_ids, N, noise=t_noise, out=out)
        self.barrier()
        dt = time.perf_counter() - t0
        # roofline leg: ONE more pass with HIP events (on the engine's own stream) around every launch of the GEMM and
        # attention families, attributed per kernel instantiation.  Events force the eager (un-graphed) launch path, so
        # this pass is separate from the timed region and not part of `value`.
        _lib.prof_reset()
        _lib.prof_enable(["conv_gemm", "attn", "norm", "other"])
        eng.synthesize_torch(t_audio, t_ids, N, noise=t_noise, out=out)
        torch.cuda.synchronize()
        _lib.prof_enable(())
        kernels = _lib.prof_kernels()
        rank_dt = per_rank_times(torch, self.dist, self.world, dt, dev)
        dt = max_over_ranks(torch, self.dist, self.world, dt, dev)
        # the reference's own bracket (F5-TTS-ONNX-Inference.py:246-312: host int16 audio + ids in, int16 waveform on the host):
        # the same steps through the host-pointer form of the C-ABI (H2D of audio / ids / noise, D2H of the waveform inside).
        # Reported next to `value`, never as `value` (bench contract: inputs resident in HBM).
        host_ms = None
        if self.world == 1:
            eng.synthesize(audio, ids, N, noise=noise)
            th = time.perf_counter()
            for _ in range(steps):
                eng.synthesize(audio, ids, N, noise=noise)
            host_ms = (time.perf_counter() - th) / steps * 1e3
        if self.dump_dir:
            np.save(os.path.join(self.dump_dir, f"f5_{dtype}_u{U}_rank{self.rank}.npy"), out.cpu().numpy())
        eng_info = eng.info()          # (after the runs: an fp16-pair engine that met its range limit has switched itself to bf16x3)
        eng.close()
        peak = MFMA_F32_PEAK_TF if dtype == "f32" else MFMA_F16_PEAK_TF
        gemm_like = [k for k in kernels if k["family"] in ("conv_gemm", "attn")]
        note = ("HIP events on the engine's stream around every launch, one separate eager pass after the timed region (the timed "
                "region replays a hipGraph; events cannot be recorded into it); flops = 2*M*N*K of the launch")
        gemm_like, merged = merge_instantiations(gemm_like)
        dom = gemm_like[0]["kernel"] if gemm_like else ""
        if "linear_x3p_kernel<float, true, 2" in dom or "linear_x3p_kernel<float, false, 2" in dom or "linear_x3d_kernel" in dom:
            # fp32 products as THREE fp16 x fp16 partial products (operands as {hi, lo * 2^11} fp16 pairs, two accumulator sets:
            # gemm_x3p.hip NP = 2): the ceiling is the dense fp16 MFMA peak / 3 in fp32-equivalent flops
            peak = MFMA_F16_PEAK_TF / 3.0
            note += ("; this kernel computes every fp32 product as 3 fp16 MFMA partial products (operands as fp16 {hi, lo} pairs, 22 "
                     "significant bits, fp32 accumulate): peak = 2500 / 3 TFLOP/s of fp32-equivalent work, achieved counts 2*M*N*K once")
        elif "linear_x3_kernel" in dom or "linear_x3p_kernel" in dom:
            # fp32 products formed as six exact bf16 x bf16 partial products (gemm_x3p.hip NP = 3 / gemm_x3.hip): the kernel runs on the bf16 pipes,
            # so its ceiling is the dense bf16 MFMA peak / 6 in fp32-equivalent flops, not the fp32 MFMA peak
            peak = MFMA_F16_PEAK_TF / 6.0
            note += ("; this kernel computes every fp32 product as 6 bf16 MFMA partial products (3-way exact operand split, fp32 "
                     "accumulate): peak = 2500 / 6 TFLOP/s of fp32-equivalent work, achieved counts 2*M*N*K once")
        if merged and "linear_x3d_kernel" in merged["kernel"]:
            note += ("; linear_x3d (gemm_x3d.hip) is the exact-fit data-parallel form of the fp16-pair GEMM, compiled per tile width and "
                     "epilogue (QKV 144 x 192 | FF1 144 x 128 | O, FF2 144 x 64): the row pools the instantiations (launch-weighted), "
                     "`instantiations` lists each, and the rocprofv3 summary carries them as linear_x3d_kernel<192 | 128 | 64, true, 1 | 2 | 3>")
        elif merged:
            note += ("; linear_x3p is compiled once per epilogue (QKV | FF1 | O / FF2: no register spills in the main loop) — the row "
                     "pools the three instantiations (launch-weighted), `instantiations` lists each, and the rocprofv3 summary "
                     "carries them as linear_x3p_kernel<float, true, 2, 0, true, 1 | 2 | 3>")
        roof = dominant_kernel_roofline(gemm_like, 1, peak, "mfma", note)
        if roof and merged and roof["kernel"] == merged["kernel"]:
            roof["instantiations"] = merged["instantiations"]
            roof["pmc_family"] = merged["pmc_family"]
        alg_flops = f5_flops_per_eval(cfg, N) * (cfg.nfe_step - 1) * U
        ev_ms = sum(k["ms"] for k in kernels)
        res = {"value": self.world * audio_s * steps / dt, "ms_per_step": dt / steps * 1e3, "dtype": dtype,
               "rtf": dt / steps / audio_s, "utterances_per_gpu": U, "frames": N, "audio_seconds_per_step_per_gpu": audio_s,
               "end_to_end_TFLOP_per_step": alg_flops / 1e12, "end_to_end_TFLOP_per_s": alg_flops / (dt / steps) / 1e12,
               "event_timed_kernel_ms_in_eager_pass": ev_ms, "roofline": roof,
               "host_io_ms_per_step": host_ms, "host_io_value": (audio_s / (host_ms * 1e-3)) if host_ms else None,
               "utterances_total": self.world * U, "utterance_seeds": [9527, 9527 + self.world * U - 1],
               "per_rank_ms": [t / steps * 1e3 for t in rank_dt],
               "arithmetic_kind": eng_info["f32_arithmetic"] if dtype == "f32" else f"{dtype} operands, fp32 accumulate",
               "adaln_fold": eng_info["adaln_fold"], "saturation_events": eng_info["saturation_events"]}
        return res, (audio, ids, N, noise)



def run_f5(args, world, rank, local, dev, dist, torch):
    fb = F5Bench(torch, dist, world, rank, local, dev)
    fb.dump_dir = args.dump_dir
    if args.f32_arithmetic:
        fb.cfg_over["f32_arithmetic"] = args.f32_arithmetic
    if args.no_adaln_fold:
        fb.cfg_over["adaln_fold"] = False
    if args.adaln_fold:
        fb.cfg_over["adaln_fold"] = True
    res, (audio, ids, N, noise) = fb.measure(args.dtype, args.batch, args.steps, args.warmup)
    if fb.small:
        args.no_secondary = args.no_cpu_baseline = True
    secondary = {}
    if world == 1 and not args.no_secondary:
        import bench_detail
        secondary = bench_detail.f5_secondaries(torch, dist, fb, args, N, local, dev)
    del fb.blob_t
    if rank != 0:
        return
    native = secondary.get("f5_f32_native_mfma")
    line = {
        "metric": "audio_seconds_per_second", "value": res["value"], "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f5_workload_name(args.dtype, args.batch, N, fb.small),
                   "utterances_per_gpu": args.batch, "utterances_total": res["utterances_total"], "utterance_seeds": res["utterance_seeds"],
                   "per_rank_ms": res["per_rank_ms"], "rank_devices": fb.rank_devices, "frames": N,
                   "audio_seconds_per_step_per_gpu": res["audio_seconds_per_step_per_gpu"], "rtf": res["rtf"],
                   "weights": "synthetic seeded (337 M DiT + 13.5 M Vocos)", "weight_bcast_ms": fb.bcast_ms,
                   "collective_backend": dist.get_backend() if dist.is_initialized() else None,
                   "arithmetic_kind": res["arithmetic_kind"], "adaln_fold": res["adaln_fold"], "saturation_events": res["saturation_events"],
                   "native_fp32_ms_per_step": native["ms_per_step"] if native else None,
                   "arithmetic": ("fp32 values, fp32 accumulation; the DiT linear layers form each fp32 product as three fp16 x fp16 partial products "
                                  "(operands as fp16 {hi, lo * 2^11} pairs = 22 significant bits, gemm_x3p.hip: measured error against float64 "
                                  "BELOW the native fp32 MFMA's), both products of attention and the grouped position convolution the same way "
                                  "(attention with the low parts unscaled: its operands are of order one) — same fp32 parity gates as the "
                                  "native fp32 MFMA path, which is timed in secondary.f5_f32_native_mfma") if args.dtype == "f32" else "16-bit operands, fp32 accumulation, fp32 residual stream",
                   "end_to_end_TFLOP_per_step": res["end_to_end_TFLOP_per_step"],
                   "end_to_end_TFLOP_per_s": res["end_to_end_TFLOP_per_s"],
                   "inputs": "resident in HBM; int16 waveform left in HBM",
                   "host_io_ms_per_step": res["host_io_ms_per_step"], "host_io_audio_s_per_s": res["host_io_value"],
                   "host_io_note": "the reference's bracket (host int16 audio + ids in, int16 waveform back on the host, H2D / D2H inside the call) timed over the same steps; reported beside value, not as value",
                   "reference_published": "README.md:29-30: 180 s (i7-1165G7, ORT CPU) / 62 s (MX150) per utterance"},
        "roofline": res["roofline"],
    }
    if secondary:
        line["secondary"] = secondary
    if world == 1 and not args.no_pmc and not fb.small and line["roofline"]:
        import bench_detail
        tb, detail = bench_detail.pmc_traffic(line["roofline"]["kernel"], args.dtype, args.batch, family=line["roofline"].get("pmc_family"))
        line["roofline"]["traffic"] = tb
        line["roofline"]["traffic_detail"] = detail
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_f5(fb.cfg, fb.raw, audio[0], ids[0], N, noise[0], full=args.cpu_baseline_full, threads=args.cpu_threads)
    emit(line)


def spawn_ranks(n: int) -> int:
    """Re-run this command line under torch.distributed.run with n ranks on this node; returns its exit code.  stdout
    (rank 0's one JSON line) and stderr pass straight through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=256, help="indextts: mel codes decoded per sentence")
    ap.add_argument("--workload", default="f5", choices=["f5", "bigvgan", "indextts_f", "indextts"],
                    help="f5 (default) = BASELINE.json's metric: F5-TTS NFE=32 end to end")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None,
                    help="f5: utterances per GPU (default 1 on one GPU = configs[2]; 8 with --gpus > 1 = configs[3] shard) / "
                         "bigvgan: mel batch (default 8)")
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--dtype", default=None, help="f5: f32 on one GPU (configs[2]), bf16 with --gpus > 1 (configs[3]) | f16 ; "
                                                  "bigvgan: f16 (default) | f32 | bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="f5: the CPU baseline runs ALL 31 evaluations at both thread settings (minutes), nothing extrapolated")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="f5: print only the cpu_baseline object (no GPU needed)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="mi_set_option(KEY, VALUE) before anything runs (A/B measurements); repeatable")
    ap.add_argument("--f32-arithmetic", default=None, choices=["fp16x2-pairs", "bf16x3", "native-fp32-mfma"],
                    help="f5 fp32: F5Config.f32_arithmetic of the engine (default: the library default, fp16x2-pairs)")
    ap.add_argument("--adaln-fold", action="store_true", help="f5: F5Config.adaln_fold = True (16-bit engines: the fold is opt-in)")
    ap.add_argument("--no-adaln-fold", action="store_true", help="f5: F5Config.adaln_fold = False (row-norm launches; A/B of the fold)")
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1: create a one-rank process group (backend nccl = RCCL) and send the weight blob through its broadcast, "
                         "as every rank of an N-GPU run does; the line then carries collective_backend and weight_bcast_ms")
    ap.add_argument("--no-pmc", action="store_true", help="f5 on one GPU: skip the two rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--no-secondary", action="store_true", help="f5 on one GPU: skip the configs[1] / configs[3]-shard blocks")
    ap.add_argument("--cpu-frames", type=int, default=128)
    ap.add_argument("--cpu-threads", type=int, default=None, help="f5: threads of the cpu_baseline leg (default 8 = the reference driver's MAX_THREADS)")
    ap.add_argument("--dump-dir", default=None, help="f5: every rank saves its int16 waveforms there (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
        # 127.0.0.1) and hand over; under `python -m torch.distributed.run ... bench.py --gpus N` the env is already there
        raise SystemExit(spawn_ranks(args.gpus))

    if args.cpu_baseline_only:
        from mi355tts.config import F5Config
        from mi355tts import weights as W
        cfg = F5Config()
        audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 1, 0)
        print(json.dumps({"cpu_baseline": cpu_baseline_f5(cfg, W.synth_state(W.f5_spec(cfg), 9527), audio[0], ids[0], N, noise[0],
                                                         full=args.cpu_baseline_full, threads=args.cpu_threads)}), flush=True)
        return

    claim_stdout()          # (after the self-launch above: the parent passes its children's stdout through)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs an MI355X (no CPU fallback) [rank {rank} of {world}]")
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                       # --force-collective: a one-rank process group so that the weight broadcast runs through RCCL
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
        # MI355TTS_BENCH_BACKEND=gloo + MI355TTS_BENCH_ONE_GPU=1: exercise the multi-rank code path on a 1-GPU box
        dist.init_process_group(os.environ.get("MI355TTS_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if os.environ.get("MI355TTS_BENCH_ONE_GPU") == "1":
        local = 0
    if os.environ.get("MI355TTS_BENCH_ONE_GPU") != "1" and torch.cuda.device_count() < world and world > 1:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) visible "
                         "(MI355TTS_BENCH_ONE_GPU=1 MI355TTS_BENCH_BACKEND=gloo maps every rank to device 0 for plumbing tests)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.option:
        from mi355tts import _lib
        for kv in args.option:
            k, v = kv.split("=", 1)
            _lib.set_option(k, int(v))
            USER_OPTIONS[k] = int(v)

    if args.workload == "f5":
        # one GPU: configs[2] (fp32, one utterance) — the config parity is gated on; N > 1: the configs[3] shard
        if args.dtype is None:
            args.dtype = "f32" if world == 1 else "bf16"
        if args.batch is None:
            args.batch = 1 if world == 1 else 8
        if args.steps is None:
            args.steps = 3
        if args.warmup is None:
            args.warmup = 2
        run_f5(args, world, rank, local, dev, dist, torch)
    else:
        import bench_detail
        bench_detail.run_other(args, world, rank, local, dev, dist, torch)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

