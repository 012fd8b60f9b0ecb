#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native TTS hot path.

    python bench.py --gpus N --steps K --warmup W [--workload bigvgan|f5|indextts_f|indextts]

Default workload = BASELINE.json's metric: F5-TTS NFE=32 end to end (preprocess -> 31 DiT evaluations with CFG ->
Vocos + ISTFT -> int16), synthetic seeded weights, 6 s reference audio + ~15-word texts (N = 1126 frames), inputs
resident in HBM.  One GPU: configs[2] (fp32, one utterance — the precision the 1e-3 RMS parity gate is stated at);
`--gpus N > 1`: the configs[3] shard (bf16, 8 utterances per GPU).  One step = one batch of utterances through the whole
path.  Metric = generated audio-seconds per wall second over all ranks; RTF (= its inverse per GPU) is in `config`.
On one GPU the line also carries `secondary`: configs[1] (BigVGAN-v2 fp16, mel (8,100,512)) and the configs[3] shard.

Multi-GPU: one process per GPU (torchrun env), utterances are independent => weak scaling, no data-path collective;
the packed weight blob is built on rank 0, broadcast over RCCL and consumed by the engine from device memory.

`roofline` = ONE kernel instantiation (the one with the largest event-timed total): its algorithmic flops (2*M*N*K) or
bytes per launch / its average launch duration (HIP events on the engine's own stream) / the matching gfx950 peak.
`cpu_baseline` = the numpy (OpenBLAS-threaded) oracle on a bounded sample of the same workload, rank 0, N=1 only.
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

USER_OPTIONS = {}                # mi_set_option keys given with --option (restored after passes that flip them)
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
    return {"value": secs / dt, "unit": "audio-s/s", "cores": int(cores), "kind": "port",
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


def cpu_baseline_f5(cfg, raw_state, audio, ids, N, noise, full: bool = False):
    """`cpu_baseline`: the numpy oracle (kind 'port', im2col + OpenBLAS sgemm) on the SAME utterance, at the reference driver's own
    thread setting (MAX_THREADS = 8, F5-TTS-ONNX-Inference.py:36) and at min(nproc, 32) threads.
    Default (the bench contract's bounded sample, 10-30 s of CPU work per thread setting): preprocess + MI355TTS_CPU_EVALS (2) of
    the 31 DiT evaluations + decode, every stage measured; `value` is the utterance rate with the mean measured evaluation time
    standing for the ones not run, and `sample` says so.  `--cpu-baseline-full`: ALL 31 evaluations run — no scaling anywhere
    (minutes per thread setting; profiles/r4/cpu_baseline_full.json is such a run on the GPU box's host)."""
    from mi355tts import weights as W
    st = W.fold_f5(cfg, raw_state)
    nproc = os.cpu_count() or 1
    tN = int(os.environ.get("MI355TTS_CPU_THREADS", min(nproc, 32)))
    evals = None if full else int(os.environ.get("MI355TTS_CPU_EVALS", "2"))
    out = {"unit": "audio-s/s", "kind": "port", "host_nproc": nproc}
    for key, T in (("tN", tN), ("t8", min(8, nproc))):
        if key == "t8" and T == tN:          # an 8-core host: one run serves both
            out[key] = dict(out["tN"])
            continue
        r = cpu_baseline_f5_run(cfg, st, audio, ids, N, noise, T, evals)
        mean_eval = sum(r["evaluations_s"]) / len(r["evaluations_s"])
        total = r["preprocess_s"] + sum(r["evaluations_s"]) + mean_eval * (r["evaluations_of_the_workload"] - r["evaluations_run"]) + r["decode_s"]
        r["utterance_s"] = total
        r["value"] = r["audio_s"] / total
        r["all_evaluations_measured"] = r["evaluations_run"] == r["evaluations_of_the_workload"]
        out[key] = r
    best = "t8" if out["t8"]["value"] >= out["tN"]["value"] else "tN"     # (on the 256-thread GPU-box host 8 threads beat 32: 1.8 s vs 4.1 s per evaluation)
    out["value"] = out[best]["value"]
    out["cores"] = out[best]["threads"]
    tN = out[best]["threads"]
    ran = out["tN"]["evaluations_run"]
    out["sample"] = (f"numpy oracle fp32, one {out['tN']['audio_s']:.2f} s utterance: preprocess + {ran} of {cfg.nfe_step - 1} DiT evaluations + decode, each "
                     f"measured, at {out['tN']['threads']} threads (tN) and at {out['t8']['threads']} threads (t8: the reference driver's MAX_THREADS); value = the faster of the two ({tN} threads); "
                     + ("every evaluation of the sampling loop ran: nothing is extrapolated" if full else
                        "the evaluations not run are counted at the mean measured evaluation time (bounded sample; the full run with "
                        "nothing extrapolated: bench.py --cpu-baseline-full, record in profiles/r4/cpu_baseline_full.json)"))
    return out


def _cores() -> int:
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] + [1]))
    except Exception:
        return os.cpu_count() or 1


X3P_ROLES = ("QKV", "FF1", "O / FF2")


def merge_instantiations(kernels):
    """linear_x3p NP = 2 is one kernel compiled per epilogue (template parameter EPK, gemm_x3p.hip): the roofline row is
    quoted on the kernel, so the per-epilogue instantiations are pooled (flops, bytes, time, launches summed) and kept
    beside the pooled row.  Returns (kernels with the pooled row in place, the pooled row or None)."""
    for tag, fam in (("AdaLN fold", r"linear_x3p_kernel<float, true, 2, 0, true, [123]>"), ("", r"linear_x3p_kernel<float, true, 2, 0, false, [123]>")):
        pre = "linear_x3p_kernel<float, true, 2, " + (tag + ", " if tag else "")
        names = [pre + r + ">" for r in X3P_ROLES] if tag else ["linear_x3p_kernel<float, true, 2, QKV>", "linear_x3p_kernel<float, true, 2, planes out>", "linear_x3p_kernel<float, true, 2>"]
        inst = [k for k in kernels if k["kernel"] in names and k["launches"] > 0]
        if len(inst) < 2:
            continue
        m = dict(inst[0])
        m["kernel"] = "linear_x3p_kernel<float, true, 2" + (", " + tag if tag else "") + ">"
        for f in ("ms", "launches", "flops", "bytes"):
            m[f] = sum(k[f] for k in inst)
        m["instantiations"] = [{"kernel": k["kernel"], "launches": k["launches"], "avg_launch_us": k["ms"] / k["launches"] * 1e3,
                                "tflops": k["flops"] / (k["ms"] * 1e-3) / 1e12} for k in inst]
        m["pmc_family"] = fam
        rest = [k for k in kernels if k not in inst]
        out = sorted(rest + [m], key=lambda k: -k["ms"])
        return out, m
    return kernels, None


def dominant_kernel_roofline(kernels, steps: int, peak: float, bound: str, note: str):
    """`roofline` of ONE kernel instantiation: the one with the largest event-timed total among `kernels`
    (_lib.prof_kernels()).  achieved = its algorithmic flops (or bytes) per launch / its average launch duration."""
    ks = [k for k in kernels if k["launches"] > 0]
    if not ks:
        return None
    k = ks[0]
    avg_ms = k["ms"] / k["launches"]
    if bound == "mfma":
        per_launch, unit = k["flops"] / k["launches"], "TFLOP/s"
        achieved = per_launch / (avg_ms * 1e-3) / 1e12
    else:
        per_launch, unit = k["bytes"] / k["launches"], "GB/s"
        achieved = per_launch / (avg_ms * 1e-3) / 1e9
    total = sum(x["ms"] for x in ks)
    return {"bound": bound, "kernel": k["kernel"], "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
            "traffic": None, "launches_per_step": k["launches"] / steps, "avg_launch_ms": avg_ms,
            ("algorithmic_flops_per_launch" if bound == "mfma" else "algorithmic_bytes_per_launch"): per_launch,
            "kernel_ms_per_step": k["ms"] / steps, "share_of_event_timed_ms": k["ms"] / total if total > 0 else 0.0,
            "note": note,
            "kernels": [{"kernel": x["kernel"], "launches_per_step": x["launches"] / steps, "ms_per_step": x["ms"] / steps,
                         "avg_launch_us": x["ms"] / x["launches"] * 1e3,
                         "tflops": x["flops"] / (x["ms"] * 1e-3) / 1e12 if x["ms"] > 0 else 0.0,
                         "alg_GBps": x["bytes"] / (x["ms"] * 1e-3) / 1e9 if x["ms"] > 0 else 0.0} for x in ks[:8]]}


def pmc_traffic(kernel_label: str, dtype: str, U: int, child=None, family=None):
    """`roofline.traffic` of the dominant kernel: fabric-side bytes per launch from rocprofv3 PMC counters, collected as
    MI355X_MICROARCH.md (HBM section) prescribes — FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one
    pass), FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B), WRITE_SIZE as reported (uncalibrated), both in KB.
    The passes run a SHORT child command (tools/pmc_f5_eval.py: one DiT evaluation of the same utterance shape on the same
    engine, ~260 dispatches — a PMC pass costs ~40 ms per dispatch) and the counters of the launches of that kernel are
    averaged.  `family`: a regular expression over the demangled kernel names — every matching instantiation is pooled
    (launch-weighted, as the event timing of a merged roofline row is) and listed on its own in the detail.
    Returns (bytes_per_launch, detail) or (None, reason)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "bench.py is itself running under a profiler: nested PMC passes skipped"
    base = kernel_label.split("<")[0].strip()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    sums, counts, names, totals, insts = {}, {}, {}, {}, {}
    with tempfile.TemporaryDirectory(prefix="mi355tts_pmc_", dir="/tmp") as td:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(td, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable,
                   *(child or [os.path.join(ROOT, "tools", "pmc_f5_eval.py"), dtype, str(U), "1"])]
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {ctr} timed out"
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} failed: {r.stderr[-300:]}"
            per = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != ctr:
                        continue
                    totals[ctr] = totals.get(ctr, 0.0) + float(row["Counter_Value"])
                    nm = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", row["Kernel_Name"])).replace("mi::", "")
                    if (not re.search(family, nm)) if family else (base not in nm):   # (f16 instantiations stay mangled in the CSV — the demangler does not know _Float16 — but carry the name)
                        continue
                    e = per.setdefault(nm, [0.0, 0])
                    e[0] += float(row["Counter_Value"]); e[1] += 1
            if not per:
                return None, f"no {base} dispatch in the {ctr} pass"
            if family:
                sums[ctr], counts[ctr], names[ctr] = sum(v[0] for v in per.values()), sum(v[1] for v in per.values()), family
                insts[ctr] = {k: v[0] * 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0) / v[1] for k, v in per.items()}
                continue
            nm = max(per, key=lambda k: per[k][1])              # the instantiation with the most launches
            sums[ctr], counts[ctr], names[ctr] = per[nm][0], per[nm][1], nm
    fetch = 2.0 * sums["FETCH_SIZE"] * 1024.0 / counts["FETCH_SIZE"]
    write = sums["WRITE_SIZE"] * 1024.0 / counts["WRITE_SIZE"]
    return fetch + write, {"kernel": names["FETCH_SIZE"], "launches_sampled": counts["FETCH_SIZE"],
                           "fetch_bytes_per_launch_x2_corrected": fetch, "write_bytes_per_launch": write,
                           "whole_command_bytes": 2.0 * totals.get("FETCH_SIZE", 0.0) * 1024.0 + totals.get("WRITE_SIZE", 0.0) * 1024.0,
                           **({"per_instantiation_bytes_per_launch": {k: insts["FETCH_SIZE"][k] + insts["WRITE_SIZE"].get(k, 0.0)
                                                                       for k in insts["FETCH_SIZE"]}} if family else {}),
                           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over " + (os.path.relpath(child[0], ROOT) + " (one forward, same shapes)" if child else "tools/pmc_f5_eval.py (one DiT evaluation, same shapes)") +
                                     "; FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md; "
                                     "fabric-side bytes (Infinity-Cache hits are counted)"}


def bcast_device_blob(torch, dist, blob_t):
    """rank 0 -> all: the library helper (mi355tts/shard.py broadcast_blob_device — RCCL over xGMI on the device buffer itself;
    gloo in the one-GPU plumbing test is staged through host memory)."""
    from mi355tts.shard import broadcast_blob_device
    return broadcast_blob_device(blob_t, src=0)


def per_rank_times(torch, dist, world, dt, dev):
    """[seconds of the timed region on rank 0, 1, ...] gathered to every rank (the line reports them next to the maximum)."""
    if world <= 1:
        return [dt]
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    out = [torch.zeros_like(tt) for _ in range(world)]
    dist.all_gather(out, tt)
    return [float(x.item()) for x in out]


def max_over_ranks(torch, dist, world, dt, dev):
    if world <= 1:
        return dt
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


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
        if world > 1:                       # the one collective of the path: weights rank 0 -> all, RCCL over xGMI
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.blob_t = bcast_device_blob(torch, dist, self.blob_t)
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
        dom = gemm_like[0]["kernel"] if gemm_like else ""
        if "linear_x3p_kernel<float, true, 2" in dom or "linear_x3p_kernel<float, false, 2" in dom:
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
        gemm_like, merged = merge_instantiations(gemm_like)
        if merged:
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


def measure_f5_plus_bigvgan(torch, fb, f5_dtype: str, voc_dtype: str, U: int, steps: int, warmup: int):
    """The pipeline BASELINE.json's metric names — F5-TTS NFE=32 + BigVGAN-v2 24 kHz: preprocess -> 31 DiT evaluations with
    CFG -> the generated mel frames handed to the BigVGAN engine (mi_f5_synthesize_mel -> mi_bigvgan_forward) -> int16, all
    on the device.  (The reference's exported F5 graphs decode with Vocos — that is the headline line; this block is the
    same sampler with the BigVGAN vocoder of configs[0]/[1] behind it.)"""
    from mi355tts.config import BigVGANConfig
    from mi355tts.f5 import F5Engine
    from mi355tts.bigvgan import BigVGANVocoder
    import dataclasses
    dev, W = fb.dev, fb.W
    # the prompt features of the F5 *_bigvgan checkpoints: the bigvgan-type mel front end (modeling_modified/F5/modules.py:30-72)
    cfg = dataclasses.replace(fb.cfg, mel_spec_type="bigvgan", **fb.cfg_over)
    vcfg = BigVGANConfig()
    eng = F5Engine(cfg, blob_device=fb.blob_t, dtype=f5_dtype, device=fb.local)
    voc = BigVGANVocoder(vcfg, blob=W.pack_bigvgan(vcfg, W.synth_state(W.bigvgan_spec(vcfg), 9527)), dtype=voc_dtype, device=fb.local)
    audio, ids, N, noise = W.f5_synthetic_inputs(cfg, U, fb.rank, L=fb.L)
    R = cfg.ref_frames(audio.shape[1])
    F = N - R
    t_audio, t_ids, t_noise = torch.from_numpy(audio).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(noise).to(dev)
    mel = torch.empty((U, cfg.mel_dim, F), dtype=torch.float32, device=dev)
    out = torch.empty((U, 1, voc.out_len(F)), dtype=torch.int16, device=dev)

    def step():
        eng.synthesize_mel_torch(t_audio, t_ids, N, noise=t_noise, out=mel)
        voc.run_torch(mel, out)

    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tv = time.perf_counter()
    for _ in range(steps):
        voc.run_torch(mel, out)
    torch.cuda.synchronize()
    voc_ms = (time.perf_counter() - tv) / steps * 1e3
    audio_s = U * out.shape[-1] / vcfg.sampling_rate
    eng.close(); voc.close()
    return {"value": audio_s / dt, "unit": "audio-s/s", "ms_per_step": dt * 1e3, "rtf": dt / audio_s, "dtype": f"{f5_dtype} DiT + {voc_dtype} vocoder",
            "vocoder_ms_per_step": voc_ms, "mel_frames": F, "utterances_per_gpu": U,
            "mel_spec_type": cfg.mel_spec_type,
            "workload": f"F5-TTS {f5_dtype} NFE=32 (N={N}, bigvgan-type prompt mel: slaney basis, center=False, {R} prompt frames) -> generated mel "
                        f"({U},100,{F}) -> BigVGAN-v2 24khz_100band_256x {voc_dtype} -> int16, one device-resident pipeline "
                        f"(mi_f5_synthesize_mel + mi_bigvgan_forward)"}


def measure_f5_two_requests(torch, fb, dtype: str, steps: int, warmup: int):
    """Two single-utterance requests served CONCURRENTLY: two engine handles (two HIP streams, each replaying its own hipGraph) driven
    from two host threads — the serving form of configs[2].  The launch tails and gaps of one persistent-kernel chain are filled by
    the other (LOG.md round 4).  Not the headline (that is one utterance at a time): a secondary block."""
    import dataclasses
    import threading
    from mi355tts.f5 import F5Engine
    cfg = dataclasses.replace(fb.cfg, **fb.cfg_over)
    dev, W = fb.dev, fb.W
    engs = [F5Engine(cfg, blob_device=fb.blob_t, dtype=dtype, device=fb.local) for _ in range(2)]
    ins, outs = [], []
    for i in range(2):
        audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 1, fb.rank, L=fb.L, first=i)
        R = cfg.ref_frames(audio.shape[1])
        ins.append((torch.from_numpy(audio).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(noise).to(dev)))
        outs.append(torch.empty((1, 1, (N - R - 1) * cfg.hop_length), dtype=torch.int16, device=dev))

    def run(i, n):
        for _ in range(n):
            engs[i].synthesize_torch(ins[i][0], ins[i][1], N, noise=ins[i][2], out=outs[i])

    for i in range(2):
        run(i, max(warmup, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, steps)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    audio_s = 2 * outs[0].shape[-1] / cfg.sample_rate
    for e in engs:
        e.close()
    return {"value": audio_s / dt, "unit": "audio-s/s", "ms_per_round_of_two": dt * 1e3, "ms_per_utterance": dt * 5e2, "rtf": dt / audio_s, "dtype": dtype,
            "workload": f"two concurrent F5-TTS {dtype} NFE=32 requests (one utterance each, N={N}) on two engine handles / HIP streams of one GPU"}


def f5_workload_name(dtype, U, N, small=False):
    if small:
        return f"PLUMBING TEST ONLY (MI355TTS_BENCH_SMALL=1): reduced F5 model, {dtype}, {U} utterance(s) per GPU, N={N}"
    which = "configs[2]" if (dtype == "f32" and U == 1) else "configs[3] shard" if (dtype == "bf16" and U == 8) else "configs[2]/[3] variant"
    return (f"F5-TTS {dtype} NFE=32 (32-point grid = 31 DiT evaluations, CFG batch 2) + Vocos/ISTFT end to end, "
            f"{U} utterance(s) per GPU, 6 s ref audio + ~15-word texts, N={N} frames (BASELINE {which})")


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
    secondary = {}
    if fb.small:
        args.no_secondary = args.no_cpu_baseline = True
    if world == 1 and not args.no_secondary:
        if not (args.dtype == "bf16" and args.batch == 8):
            r2, _ = fb.measure("bf16", 8, 10, 2)
            r2["workload"] = f5_workload_name("bf16", 8, N)
            secondary["f5_bf16_u8"] = r2
        if args.dtype == "f32":
            # the same fp32 workload with the linear layers on the native fp32 MFMA (v_mfma_f32_32x32x2_f32) instead of the
            # exact bf16x3 products: both pass the same fp32 parity gates; reported so that either can be taken as the fp32 number
            r3, _ = fb.measure("f32", args.batch, 5, 2, f32_arithmetic="native-fp32-mfma")
            r3["workload"] = f5_workload_name("f32", args.batch, N) + " — linear layers, attention and position convolution on the native fp32 MFMA (F5Config.f32_arithmetic = native-fp32-mfma)"
            secondary["f5_f32_native_mfma"] = r3
            if args.batch == 1:
                # the same fp32 arithmetic with four utterances per step (8 CFG rows): what one GPU serves when requests
                # can be batched — the fixed per-launch cost of the DiT linear layers is shared by four times the rows
                r4, _ = fb.measure("f32", 4, 3, 1)
                r4["workload"] = f5_workload_name("f32", 4, N)
                secondary["f5_f32_u4"] = r4
    if world == 1 and not args.no_secondary and not fb.small:
        secondary["f5_plus_bigvgan"] = measure_f5_plus_bigvgan(torch, fb, args.dtype, "f16", args.batch, 3, 2)
        if args.dtype == "f32" and args.batch == 1:
            secondary["f5_f32_two_requests"] = measure_f5_two_requests(torch, fb, "f32", 4, 2)
    del fb.blob_t
    if rank != 0:
        return
    if world == 1 and not args.no_secondary:
        secondary["bigvgan_f16_b8"] = measure_bigvgan(torch, dist, 1, 0, local, dev, "f16", 8, 512, 10, 3, False)[0]
        if not args.no_pmc:
            bigvgan_pmc(secondary["bigvgan_f16_b8"], "f16", 8)
    line = {
        "metric": "audio_seconds_per_second", "value": res["value"], "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f5_workload_name(args.dtype, args.batch, N, fb.small),
                   "utterances_per_gpu": args.batch, "utterances_total": res["utterances_total"], "utterance_seeds": res["utterance_seeds"],
                   "per_rank_ms": res["per_rank_ms"], "rank_devices": fb.rank_devices, "frames": N,
                   "audio_seconds_per_step_per_gpu": res["audio_seconds_per_step_per_gpu"], "rtf": res["rtf"],
                   "weights": "synthetic seeded (337 M DiT + 13.5 M Vocos params)", "weight_bcast_ms": fb.bcast_ms,
                   "collective_backend": dist.get_backend() if world > 1 else None,
                   "arithmetic_kind": res["arithmetic_kind"], "adaln_fold": res["adaln_fold"], "saturation_events": res["saturation_events"],
                   "arithmetic": ("fp32 values, fp32 accumulation; the DiT linear layers form each fp32 product as three fp16 x fp16 partial products "
                                  "(operands as fp16 {hi, lo * 2^11} pairs = 22 significant bits, gemm_x3p.hip: measured error against float64 "
                                  "BELOW the native fp32 MFMA's), both products of attention and the grouped position convolution the same way "
                                  "(attention with the low parts unscaled: its operands are of order one) — same fp32 parity gates as the "
                                  "native fp32 MFMA path, which is timed in secondary.f5_f32_native_mfma") if args.dtype == "f32" else "16-bit operands, fp32 accumulation, fp32 residual stream",
                   "end_to_end_TFLOP_per_step": res["end_to_end_TFLOP_per_step"],
                   "end_to_end_TFLOP_per_s": res["end_to_end_TFLOP_per_s"],
                   "inputs": "audio / text ids / injected noise resident in HBM, int16 waveform left in HBM",
                   "host_io_ms_per_step": res["host_io_ms_per_step"], "host_io_audio_s_per_s": res["host_io_value"],
                   "host_io_note": "the reference's bracket (host int16 audio + ids in, int16 waveform back on the host, H2D / D2H inside the call) timed over the same steps; reported beside value, not as value",
                   "reference_published": "README.md:29-30: 180 s (i7-1165G7, ORT CPU) / 62 s (MX150) per utterance"},
        "roofline": res["roofline"],
    }
    if secondary:
        line["secondary"] = secondary
    if world == 1 and not args.no_pmc and not fb.small and line["roofline"]:
        tb, detail = pmc_traffic(line["roofline"]["kernel"], args.dtype, args.batch, family=line["roofline"].get("pmc_family"))
        line["roofline"]["traffic"] = tb
        line["roofline"]["traffic_detail"] = detail
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_f5(fb.cfg, fb.raw, audio[0], ids[0], N, noise[0], full=args.cpu_baseline_full)
    print(json.dumps(line), flush=True)


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
        blob_t = bcast_device_blob(torch, dist, blob_t)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    NB = max(1, args.batch)
    gcfg.max_batch = NB
    gpt = IndexGPT(gcfg, blob_device=blob_t[:ng].contiguous(), dtype=args.dtype, device=local)
    voc = BigVGANVocoder(vcfg, blob_device=blob_t[ng:].contiguous(), dtype=args.dtype, device=local)
    del blob_t
    n_text, n_tok = 30, args.tokens
    text = (np.arange(n_text, dtype=np.int32) * 37 + 11 * rank) % (gcfg.text_tokens - 2) + 2
    # graph A (Inference_IndexTTS_ONNX.py:700-707): 6 s of int16 prompt audio -> conds_latent (the GPT prompt's first rows) and
    # the vocoder conditioning vectors; synthetic seeded weights like the other engines
    from mi355tts.config import IndexCondConfig
    from mi355tts.indextts import IndexCond
    ccfg = IndexCondConfig()
    cond_eng = IndexCond(ccfg, W.synth_state(W.cond_spec(ccfg), 9527, fast=True), device=local)
    tt = np.arange(144000) / 24000.0
    prompt_audio = np.clip(0.1 * 32767 * np.sin(2 * np.pi * 220.0 * tt) + W.synth_normal_fast(7 + rank, "prompt_audio", (144000,), std=500.0),
                           -32768, 32767).astype(np.int16)
    text_h = gpt.text_embed(text)
    mel_h, _ = gpt.mel_embed(gcfg.start_mel_token, 0)
    n_cond = ccfg.latents

    def graph_a():
        vc, lat = cond_eng.run(prompt_audio)
        pr, cl = gpt.concat(lat[None], text_h, mel_h)
        return torch.from_numpy(vc).to(dev), torch.from_numpy(pr[0]).to(dev), int(cl[0]), pr

    vconds, prompt, P, prompt_np = graph_a()
    toks = torch.zeros((n_tok,), dtype=torch.int32, device=dev)
    hid = torch.zeros((n_tok, gcfg.hidden), dtype=torch.float32, device=dev)
    ncond = vcfg.upsample_initial_channel + sum(vcfg.stage_channels(i) for i in range(vcfg.num_upsamples))
    assert ncond == cond_eng.ncond and gcfg.hidden == ccfg.model_dim
    wav = torch.empty((1, 1, (n_tok - 2) * vcfg.hop + 30), dtype=torch.int16, device=dev)
    audio_s = NB * wav.shape[-1] / vcfg.sampling_rate
    if NB > 1:      # NB sentences per step: different texts, one shared weight stream per decode step
        ps = []
        for b in range(NB):
            tb = (np.arange(n_text, dtype=np.int32) * 37 + 11 * rank + 101 * b) % (gcfg.text_tokens - 2) + 2
            ps.append(gpt.concat(prompt_np[:, :n_cond], gpt.text_embed(tb), mel_h)[0][0])
        prompts_cat = torch.from_numpy(np.concatenate(ps, axis=0)).to(dev)
        toks_b = torch.zeros((NB, n_tok), dtype=torch.int32, device=dev)
        hid_b = torch.zeros((NB, n_tok, gcfg.hidden), dtype=torch.float32, device=dev)

    state = {"vconds": vconds, "prompt": prompt}

    def gpt_leg():
        if NB == 1:
            n = gpt.generate_torch(state["prompt"], n_tok, toks, hid, stop_tokens=[])
            assert n == n_tok
        else:
            n = gpt.generate_batch_torch(prompts_cat, [P] * NB, [n_tok] * NB, toks_b, hid_b, stop_tokens=[])
            assert (n == n_tok).all()

    def step():
        # prompt audio -> graph A -> prompt rows + vocoder conditioning (once per utterance, as the driver does), then
        # stop_tokens=[]: a fixed amount of work per sentence (random weights never emit the stop code on cue)
        state["vconds"], state["prompt"], _, _ = graph_a()
        gpt_leg()
        for b in range(NB):
            voc.run_latent_torch(hid if NB == 1 else hid_b[b], state["vconds"], wav)

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
    dt = max_over_ranks(torch, dist, world, dt, dev)
    if rank != 0:
        gpt.close(); voc.close(); cond_eng.close()
        return
    esz = 4 if args.dtype == "f32" else 2
    achieved = pg["bytes"] / (pg["ms"] * 1e-3) / 1e9 if pg["ms"] > 0 else 0.0
    wbytes = (gcfg.layers * 12 * gcfg.hidden * gcfg.hidden + gcfg.mel_codes * gcfg.hidden) * esz
    line = {
        "metric": "audio_seconds_per_second", "value": world * audio_s * args.steps / dt, "unit": "audio-s/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"IndexTTS-1.5 {args.dtype}: GPT-2 (24 x 1280, 20 heads) prompt pass of {P} rows + greedy decode of "
                               f"{n_tok} mel codes + BigVGAN graph F, {NB} sentence(s) per GPU per step, 6 s of int16 prompt audio through graph A "
                               f"(Conformer / Perceiver / ECAPA) every step (BASELINE configs[4], all six graphs)",
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
    gpt.close(); voc.close(); cond_eng.close()


def measure_bigvgan(torch, dist, world, rank, local, dev, dtype, B, F, steps, warmup, ixf):
    """BigVGAN-v2 (BASELINE configs[0]/[1]) or IndexTTS graph F (`ixf`): one step = one vocoder pass over the batch, mel
    resident in HBM.  Per-kernel HIP events are taken in a separate pass after the timed region."""
    from mi355tts.config import BigVGANConfig
    from mi355tts import weights as W
    from mi355tts import _lib
    from mi355tts.bigvgan import BigVGANVocoder
    cfg = BigVGANConfig.indextts() if ixf else BigVGANConfig()
    spec = W.bigvgan_spec(cfg)
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
        blob_t = bcast_device_blob(torch, dist, blob_t)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    voc = BigVGANVocoder(cfg, blob_device=blob_t, dtype=dtype, device=local)
    del blob_t
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
    for _ in range(warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(torch, dist, world, dt, dev)
    # per-kernel durations: a separate pass after the timed region, with every launch on the engine's one stream — in the
    # timed region the AMP blocks of a stage run on side streams (bigvgan_streams), where a launch's event-to-event time
    # includes whatever ran beside it
    _lib.set_option("bigvgan_streams", 1)
    step()
    _lib.prof_reset()
    _lib.prof_enable(["conv_gemm", "aa_act", "conv_post"])
    psteps = min(steps, 5)
    for _ in range(psteps):
        step()
    torch.cuda.synchronize()
    _lib.prof_enable(())
    kernels = _lib.prof_kernels()
    _lib.set_option("bigvgan_streams", USER_OPTIONS.get("bigvgan_streams", 3))       # (what the caller asked for with --option, else the default)
    voc.close()
    esz = 4 if dtype == "f32" else 2
    alg = bigvgan_algorithmic_bytes(cfg, B, F, esz)
    # dominant kernel: the MFMA-bound implicit-GEMM of stages 0-2 when it leads, else the HBM-bound fused AA conv
    roof = None
    if kernels:
        lead = kernels[0]
        mfma_bound = lead["kernel"].startswith("conv_gemm")
        roof = dominant_kernel_roofline(
            kernels, psteps, (MFMA_F32_PEAK_TF if dtype == "f32" else MFMA_F16_PEAK_TF) if mfma_bound else HBM_PEAK_GBS,
            "mfma" if mfma_bound else "hbm",
            "HIP events on the engine's stream around every launch, one separate one-stream pass after the timed region (the "
            "timed region runs the AMP blocks of a stage on side streams); per-launch work = 2*M*N*K flops "
            "(implicit GEMM) / layer-granular algorithmic bytes (x + w + out [+ res])")
    res = {"value": world * audio_s * steps / dt, "ms_per_step": dt / steps * 1e3, "dtype": dtype,
           "rtf": dt / steps / audio_s, "batch_per_gpu": B, "frames": F, "audio_seconds_per_step_per_gpu": audio_s,
           "workload": (f"IndexTTS graph F (speaker-conditioned BigVGAN, 1024x) {dtype}, T_codes = {F + 2} (BASELINE configs[4] "
                        f"vocoder leg)") if ixf else
                       (f"BigVGAN-v2 24khz_100band_256x {dtype} vocoder, mel ({B},100,{F}) per GPU (BASELINE configs[1])"),
           "weight_bcast_ms": bcast_ms, "whole_forward_algorithmic_GB": alg / 1e9,
           "whole_forward_algorithmic_GBps": alg / (dt / steps) / 1e9,
           "whole_forward_frac_of_hbm_peak": alg / (dt / steps) / 1e9 / HBM_PEAK_GBS, "roofline": roof}
    return res, (cfg, state)


def bigvgan_pmc(res, dtype: str, B: int):
    """roofline.traffic of the vocoder block: fabric-side bytes per launch of its dominant kernel AND of the whole forward (one
    forward of the same mel shape under two separate --pmc passes, tools/pmc_bigvgan.py), next to the layer-granular algorithmic
    bytes — traffic above the algorithmic figure is re-reads, below it is what the fusion saved."""
    if not res.get("roofline"):
        return
    tb, detail = pmc_traffic(res["roofline"]["kernel"], dtype, B, child=[os.path.join(ROOT, "tools", "pmc_bigvgan.py"), dtype, str(B), "1"])
    res["roofline"]["traffic"] = tb
    res["roofline"]["traffic_detail"] = detail
    if isinstance(detail, dict) and detail.get("whole_command_bytes"):
        res["whole_forward_fabric_GB"] = detail["whole_command_bytes"] / 1e9
        res["whole_forward_fabric_over_algorithmic"] = detail["whole_command_bytes"] / 1e9 / res["whole_forward_algorithmic_GB"]


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
    ap.add_argument("--no-pmc", action="store_true", help="f5 on one GPU: skip the two rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--no-secondary", action="store_true", help="f5 on one GPU: skip the configs[1] / configs[3]-shard blocks")
    ap.add_argument("--cpu-frames", type=int, default=128)
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
                                                         full=args.cpu_baseline_full)}), flush=True)
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs an MI355X (no CPU fallback) [rank {rank} of {world}]")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    elif args.workload == "indextts":
        args.steps = 3 if args.steps is None else args.steps
        args.warmup = 3 if args.warmup is None else args.warmup
        args.batch = 1 if args.batch is None else args.batch
        args.dtype = "f16" if args.dtype is None else args.dtype
        run_indextts(args, world, rank, local, dev, dist, torch)
    else:
        ixf = args.workload == "indextts_f"        # BASELINE configs[4] vocoder leg: IndexTTS graph F, T_codes = 128
        args.steps = 20 if args.steps is None else args.steps
        args.warmup = 3 if args.warmup is None else args.warmup
        args.dtype = "f16" if args.dtype is None else args.dtype
        B, F = (1, 126) if ixf else (8 if args.batch is None else args.batch, args.frames)
        res, (cfg, state) = measure_bigvgan(torch, dist, world, rank, local, dev, args.dtype, B, F, args.steps, args.warmup, ixf)
        if rank == 0:
            line = {"metric": "audio_seconds_per_second", "value": res["value"], "unit": "audio-s/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                    "config": {k: v for k, v in res.items() if k not in ("value", "ms_per_step", "dtype", "roofline")},
                    "roofline": res["roofline"]}
            line["config"]["weights"] = "synthetic seeded (112.4 M params)"
            if world == 1 and not args.no_pmc and not ixf:
                bigvgan_pmc(res, args.dtype, B)
                line["roofline"] = res["roofline"]
                for k in ("whole_forward_fabric_GB", "whole_forward_fabric_over_algorithmic"):
                    if k in res:
                        line["config"][k] = res[k]
            if world == 1 and not args.no_cpu_baseline and not ixf:
                line["cpu_baseline"] = cpu_baseline_bigvgan(cfg, state, args.cpu_frames)
            print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
