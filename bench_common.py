"""bench_common.py — constants and helpers shared by bench.py (the headline line) and bench_detail.py (secondary
workloads, PMC passes): gfx950 peaks, the per-kernel roofline row from the library's HIP-event table, rank reductions."""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
sys.path.insert(0, ROOT)

USER_OPTIONS = {}                # mi_set_option keys given with --option (restored after passes that flip them)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TF = 2500.0        # dense bf16/f16
MFMA_F32_PEAK_TF = 157.3


def f5_flops_per_eval(cfg, N: int) -> float:
    """Algorithmic FLOPs of one DiT CFG evaluation: tokens = 2N,
    MAC/token = depth*(4d^2 + 2*d*ff + 2*N*d) + (cat*d + 2*d*(d/g)*k + d*mel)."""
    d, ff = cfg.dim, cfg.ff_dim
    mac = cfg.depth * (4 * d * d + 2 * d * ff + 2 * N * d) + ((2 * cfg.mel_dim + cfg.text_dim) * d +
                                                               2 * d * (d // cfg.pos_conv_groups) * cfg.pos_conv_kernel + d * cfg.mel_dim)
    return 2.0 * (2 * N) * mac



X3P_ROLES = ("QKV", "FF1", "O / FF2")


def merge_instantiations(kernels):
    """linear_x3p NP = 2 is one kernel compiled per epilogue (template parameter EPK, gemm_x3p.hip): the roofline row is
    quoted on the kernel, so the per-epilogue instantiations are pooled (flops, bytes, time, launches summed) and kept
    beside the pooled row.  Returns (kernels with the pooled row in place, the pooled row or None)."""
    # round 6: the exact-fit data-parallel form of the same GEMM (gemm_x3d.hip), compiled per tile width x epilogue
    for tag, fam in (("AdaLN fold", r"linear_x3d_kernel<(64|128|192), true, [123]>"), ("", r"linear_x3d_kernel<(64|128|192), false, [123]>")):
        roles = ("AdaLN fold, QKV", "AdaLN fold, FF1", "AdaLN fold, O / FF2") if tag else ("QKV", "planes out", "rows out")
        names = [f"linear_x3d_kernel<{w}, {r}>" for w in (64, 128, 192) for r in roles]
        inst = [k for k in kernels if k["kernel"] in names and k["launches"] > 0]
        if len(inst) < 2:
            continue
        m = dict(inst[0])
        m["kernel"] = "linear_x3d_kernel<144 x {64,128,192} tiles, fp16 pairs" + (", " + tag if tag else "") + ">"
        for f in ("ms", "launches", "flops", "bytes"):
            m[f] = sum(k[f] for k in inst)
        m["instantiations"] = [{"kernel": k["kernel"], "launches": k["launches"], "avg_launch_us": k["ms"] / k["launches"] * 1e3,
                                "tflops": k["flops"] / (k["ms"] * 1e-3) / 1e12} for k in inst]
        m["pmc_family"] = fam
        rest = [k for k in kernels if k not in inst]
        out = sorted(rest + [m], key=lambda k: -k["ms"])
        return out, m
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


def bcast_device_blob(torch, dist, blob_t, force=False):
    """rank 0 -> all: the library helper (mi355tts/shard.py broadcast_blob_device — RCCL over xGMI on the device buffer itself;
    gloo in the one-GPU plumbing test is staged through host memory)."""
    from mi355tts.shard import broadcast_blob_device
    return broadcast_blob_device(blob_t, src=0, force=force)


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



def f5_workload_name(dtype, U, N, small=False):
    if small:
        return f"PLUMBING TEST ONLY (MI355TTS_BENCH_SMALL=1): reduced F5 model, {dtype}, {U} utterance(s) per GPU, N={N}"
    which = "configs[2]" if (dtype == "f32" and U == 1) else "configs[3] shard" if (dtype == "bf16" and U == 8) else "configs[2]/[3] variant"
    return (f"F5-TTS {dtype} NFE=32 (32-point grid = 31 DiT evaluations, CFG batch 2) + Vocos/ISTFT end to end, "
            f"{U} utterance(s) per GPU, 6 s ref audio + ~15-word texts, N={N} frames (BASELINE {which})")



# ---- the output contract: ONE compact JSON line on stdout, the full record beside it -------------------------------------------

COMPACT_LIMIT = 4096
_TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_ROOF = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch",
         "avg_launch_ms", "launches_per_step", "kernel_ms_per_step")
_CPU = ("value", "unit", "kind", "cores", "host_nproc", "evaluations_run", "evaluations_of_the_workload", "all_evaluations_measured", "sample")


def _short(v, limit):
    """scalars, short strings and short flat lists pass; anything else is detail."""
    if v is None or isinstance(v, (bool, int)):
        return True
    if isinstance(v, float):
        return v == v and abs(v) != float("inf")
    if isinstance(v, str):
        return len(v) <= limit
    if isinstance(v, (list, tuple)):
        return len(v) <= 8 and all(_short(x, 48) and not isinstance(x, (list, tuple)) for x in v)
    return False


def _num(v):
    if isinstance(v, float):
        return float(f"{v:.10g}") if (v == v and abs(v) != float("inf")) else None
    return v


def compact_line(line: dict) -> dict:
    """The driver-readable form of a full bench record: required keys, flat `config`, a flat `roofline`, a flat `cpu_baseline`,
    one number per secondary workload (`secondary_ms`) and the few secondary roofline fractions the review asks for."""
    out = {k: _num(line[k]) for k in _TOP if k in line}
    cfg = {}
    for k, v in (line.get("config") or {}).items():
        if _short(v, 260 if k == "workload" else 64):
            cfg[k] = [_num(x) for x in v] if isinstance(v, (list, tuple)) else _num(v)
    out["config"] = cfg
    roof = line.get("roofline")
    out["roofline"] = {k: (roof[k][:96] if k == "kernel" else _num(roof[k])) for k in _ROOF if k in roof} if roof else None
    cpu = line.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {k: (cpu[k][:200] if k == "sample" else _num(cpu[k])) for k in _CPU if k in cpu}
    sec = line.get("secondary") or {}
    if sec:
        out["secondary_ms"] = {k: _num(v.get("ms_per_utterance", v.get("ms_per_step"))) for k, v in sec.items()}
        for k, v in sec.items():
            if "ms_per_step_host_numpy_form" in v:                # the façade loop in its host-array form, next to the io-binding form
                out["secondary_ms"][k + "_host_numpy"] = _num(v["ms_per_step_host_numpy_form"])
        sr = {}
        for k, v in sec.items():
            r = v.get("roofline") if isinstance(v, dict) else None
            if r:
                sr[k] = {"kernel": r["kernel"][:64], "frac": _num(r["frac"]), "traffic": _num(r.get("traffic")),
                         "alg_per_launch": _num(r.get("algorithmic_flops_per_launch", r.get("algorithmic_bytes_per_launch")))}
                if "whole_forward_frac_of_hbm_peak" in v:
                    sr[k]["whole_forward_frac_of_hbm_peak"] = _num(v["whole_forward_frac_of_hbm_peak"])
        if sr:
            out["secondary_roofline"] = sr
    out["detail_file"] = "bench_detail.json"
    return out


_LINE_FD = None


def claim_stdout() -> None:
    """stdout belongs to the ONE result line: from here on everything else this process (and the native libraries it loads: gloo
    announces its ranks on stdout, RCCL can be told to) writes to fd 1 lands on stderr; emit() writes the line to the real stdout."""
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line: dict) -> str:
    """Write the full record to bench_detail.json (+ gpurun_out/ if present, + stderr) and print the compact line as the ONE stdout line."""
    import json
    full = json.dumps(line)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
    sys.stderr.write("BENCH_DETAIL " + full + "\n")
    sys.stderr.flush()
    c = compact_line(line)
    s = json.dumps(c, allow_nan=False)
    while len(s) >= COMPACT_LIMIT and c.get("secondary_roofline"):       # (never expected; keeps the contract if a kernel name grows)
        c["secondary_roofline"].popitem()
        s = json.dumps(c, allow_nan=False)
    if len(s) >= COMPACT_LIMIT:
        c["config"] = {k: v for k, v in c["config"].items() if not isinstance(v, (str, list)) or k == "workload"}
        s = json.dumps(c, allow_nan=False)
    assert len(s) < COMPACT_LIMIT, len(s)
    if _LINE_FD is not None:
        sys.stdout.flush()
        os.write(_LINE_FD, (s + "\n").encode())
    else:
        print(s, flush=True)
    return s
