"""ctypes binding of libmi355tts.so (C-ABI: include/mi355tts.h).

The product path has NO CPU fallback: if the shared library is missing or no MI355X is visible,
calls raise loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MI_F32, MI_F16, MI_BF16 = 0, 1, 2
MI_HOST, MI_DEVICE = 0, 1
DTYPES = {"f32": MI_F32, "fp32": MI_F32, "float32": MI_F32, "f16": MI_F16, "fp16": MI_F16, "float16": MI_F16,
          "bf16": MI_BF16, "bfloat16": MI_BF16}

# MI355TTS_LIB: another build of the same library (A/B measurements of kernel variants); the default is the in-tree build
_LIB_PATH = os.environ.get("MI355TTS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmi355tts.so")
_lib = None


class MiError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """dlopen the engine (no GPU needed for this step) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise MiError(f"{_LIB_PATH} is missing: build it with `python text-to-speech-tts-onnx_amd/build.py` "
                      f"(or __graft_entry__.build()); there is no CPU fallback")
    # torch-ROCm ships its own copy of the HIP runtime.  If this library's copy initialises first, a later
    # `torch.cuda` initialisation in the same process reports "No HIP GPUs are available" (seen with the device-resident
    # OrtValues of ort_compat in a script that opened its sessions before touching torch); the other order works.  So when
    # torch is installed it goes first (MI355TTS_NO_TORCH=1 skips this for pure-numpy users who never hand over torch tensors).
    # A torch install that cannot even be imported (missing, or broken: OSError / RuntimeError from a mismatched build) must not
    # make THIS library unloadable: numpy-only callers never need it.
    if os.environ.get("MI355TTS_NO_TORCH") != "1":
        try:
            import torch
            torch.cuda.is_available()
        except Exception:
            pass
    L = C.CDLL(_LIB_PATH)
    i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    vp = C.c_void_p
    L.mi_init.argtypes = [C.c_int]; L.mi_init.restype = C.c_int
    L.mi_device_count.argtypes = []; L.mi_device_count.restype = C.c_int
    L.mi_last_error.argtypes = []; L.mi_last_error.restype = C.c_char_p
    L.mi_version.argtypes = []; L.mi_version.restype = C.c_char_p
    L.mi_bigvgan_param_count.argtypes = [i32p, C.c_int]; L.mi_bigvgan_param_count.restype = C.c_int64
    L.mi_bigvgan_create.argtypes = [i32p, C.c_int, f32p, C.c_int64, C.c_int, C.c_int]
    L.mi_bigvgan_create.restype = vp
    L.mi_bigvgan_create_mem.argtypes = [i32p, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.mi_bigvgan_create_mem.restype = vp
    L.mi_bigvgan_destroy.argtypes = [vp]; L.mi_bigvgan_destroy.restype = None
    L.mi_bigvgan_out_len.argtypes = [vp, C.c_int]; L.mi_bigvgan_out_len.restype = C.c_int64
    L.mi_bigvgan_forward.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int]; L.mi_bigvgan_forward.restype = C.c_int
    L.mi_bigvgan_forward_f32.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int]
    L.mi_bigvgan_forward_f32.restype = C.c_int
    L.mi_bigvgan_forward_latent.argtypes = [vp, vp, C.c_int, vp, C.c_int64, vp, vp, C.c_int]
    L.mi_bigvgan_forward_latent.restype = C.c_int
    L.mi_aa_activation1d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, f32p]
    L.mi_aa_activation1d.restype = C.c_int
    L.mi_aa_conv1d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.c_int, f32p, C.c_int, f32p]
    L.mi_aa_conv1d.restype = C.c_int
    L.mi_conv1d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_int, f32p]
    L.mi_conv1d.restype = C.c_int
    L.mi_conv_transpose1d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, f32p]
    L.mi_conv_transpose1d.restype = C.c_int
    i16p, i64p = C.POINTER(C.c_int16), C.POINTER(C.c_int64)
    L.mi_f5_param_count.argtypes = [i32p, C.c_int, f32p, C.c_int]; L.mi_f5_param_count.restype = C.c_int64
    L.mi_f5_create.argtypes = [i32p, C.c_int, f32p, C.c_int, f32p, C.c_int64, C.c_int, C.c_int]
    L.mi_f5_create.restype = vp
    L.mi_f5_create_mem.argtypes = [i32p, C.c_int, f32p, C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.mi_f5_create_mem.restype = vp
    L.mi_f5_destroy.argtypes = [vp]; L.mi_f5_destroy.restype = None
    L.mi_f5_tables.argtypes = [vp, f32p, f32p]; L.mi_f5_tables.restype = C.c_int
    L.mi_f5_info.argtypes = [vp, C.c_char_p]; L.mi_f5_info.restype = C.c_int64
    L.mi_f5_preprocess.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_int64, vp, C.c_uint64, vp, vp, vp, vp, vp,
                                   i64p, C.c_int]
    L.mi_f5_preprocess.restype = C.c_int
    L.mi_f5_stft.argtypes = [vp, vp, C.c_int64, vp, C.c_int]; L.mi_f5_stft.restype = C.c_int
    L.mi_f5_transformer_step.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, i32p, C.c_int, C.c_int]
    L.mi_f5_transformer_step.restype = C.c_int
    L.mi_f5_sample.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.mi_f5_sample.restype = C.c_int
    L.mi_f5_dit_eval.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int64, C.c_int, vp, C.c_int]
    L.mi_f5_dit_eval.restype = C.c_int
    L.mi_f5_decode.argtypes = [vp, vp, C.c_int, C.c_int64, C.c_int64, vp, vp, i64p, C.c_int]
    L.mi_f5_decode.restype = C.c_int
    L.mi_f5_synthesize.argtypes = [vp, C.c_int, vp, C.c_int64, vp, C.c_int64, C.c_int64, vp, C.c_uint64, vp, i64p,
                                   C.c_int]
    L.mi_f5_synthesize.restype = C.c_int
    L.mi_f5_synthesize_mel.argtypes = [vp, C.c_int, vp, C.c_int64, vp, C.c_int64, C.c_int64, vp, C.c_uint64, vp, i64p,
                                       C.c_int]
    L.mi_f5_synthesize_mel.restype = C.c_int
    L.mi_indextts_cond_param_count.argtypes = [C.POINTER(C.c_int32), C.c_int]; L.mi_indextts_cond_param_count.restype = C.c_int64
    L.mi_indextts_cond_create.argtypes = [C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float), C.c_int64, C.c_int]
    L.mi_indextts_cond_create.restype = vp
    L.mi_indextts_cond_destroy.argtypes = [vp]; L.mi_indextts_cond_destroy.restype = None
    L.mi_indextts_cond_run.argtypes = [vp, vp, C.c_int64, vp, vp, vp, C.c_int]; L.mi_indextts_cond_run.restype = C.c_int
    L.mi_gpt_param_count.argtypes = [C.POINTER(C.c_int32), C.c_int]; L.mi_gpt_param_count.restype = C.c_int64
    L.mi_gpt_create.argtypes = [C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float), C.c_int64, C.c_int, C.c_int]
    L.mi_gpt_create.restype = C.c_void_p
    L.mi_gpt_create_mem.argtypes = [C.POINTER(C.c_int32), C.c_int, vp, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.mi_gpt_create_mem.restype = C.c_void_p
    L.mi_gpt_destroy.argtypes = [vp]; L.mi_gpt_destroy.restype = None
    L.mi_gpt_text_embed.argtypes = [vp, vp, C.c_int, vp, C.c_int]; L.mi_gpt_text_embed.restype = C.c_int
    L.mi_gpt_mel_embed.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_int]; L.mi_gpt_mel_embed.restype = C.c_int
    L.mi_gpt_reset.argtypes = [vp]; L.mi_gpt_reset.restype = C.c_int
    L.mi_gpt_history_len.argtypes = [vp]; L.mi_gpt_history_len.restype = C.c_int64
    L.mi_gpt_step.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int]; L.mi_gpt_step.restype = C.c_int
    L.mi_gpt_kv_read.argtypes = [vp, C.c_int, vp, vp, C.c_int]; L.mi_gpt_kv_read.restype = C.c_int
    L.mi_gpt_kv_write.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int]; L.mi_gpt_kv_write.restype = C.c_int
    L.mi_gpt_generate.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp,
                                  C.POINTER(C.c_int32), C.c_int]
    L.mi_gpt_generate.restype = C.c_int
    L.mi_gpt_generate_batch.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp, C.c_int,
                                        C.c_float, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_int32), C.c_int]
    L.mi_gpt_generate_batch.restype = C.c_int
    L.mi_bench_conv_gemm.argtypes = [C.c_int] * 9 + [C.POINTER(C.c_double)]
    L.mi_bench_conv_gemm.restype = C.c_int
    L.mi_set_option.argtypes = [C.c_char_p, C.c_int64]; L.mi_set_option.restype = C.c_int
    L.mi_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]; L.mi_device_pci_bus_id.restype = C.c_int
    L.mi_prof_enable.argtypes = [C.c_int]; L.mi_prof_enable.restype = C.c_int
    L.mi_prof_reset.argtypes = []; L.mi_prof_reset.restype = C.c_int
    L.mi_prof_get.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                              C.POINTER(C.c_double)]
    L.mi_prof_get.restype = C.c_int
    L.mi_prof_kernel_count.argtypes = []; L.mi_prof_kernel_count.restype = C.c_int
    L.mi_prof_kernel_get.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mi_prof_kernel_get.restype = C.c_int
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mi_last_error().decode("utf-8", "replace")
        raise MiError(f"{what}: {msg} (code {rc})")


def f32p(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def i32p(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def init(device: int = 0) -> None:
    check(load().mi_init(device), "mi_init")


PROF_FAMILIES = ("conv_gemm", "aa_act", "conv_post", "attn", "norm", "other")


def prof_enable(families=()) -> None:
    """families: iterable of family names, or True for all, or False/() for off."""
    if families is True:
        mask = -1
    elif not families:
        mask = 0
    else:
        mask = 0
        for f in families:
            mask |= 1 << PROF_FAMILIES.index(f)
    load().mi_prof_enable(mask)


def prof_reset() -> None:
    check(load().mi_prof_reset(), "mi_prof_reset")


def prof_get(family: str) -> dict:
    ms, n, by, fl = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    check(load().mi_prof_get(family.encode(), C.byref(ms), C.byref(n), C.byref(by), C.byref(fl)), "mi_prof_get")
    return {"ms": ms.value, "launches": n.value, "bytes": by.value, "flops": fl.value}


def prof_kernels() -> list:
    """Per-kernel-instantiation accumulators since the last prof_reset, sorted by total time (descending)."""
    L = load()
    out = []
    for i in range(L.mi_prof_kernel_count()):
        name, fam = C.create_string_buffer(256), C.create_string_buffer(32)
        ms, by, fl, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
        check(L.mi_prof_kernel_get(i, name, 256, fam, 32, C.byref(ms), C.byref(n), C.byref(by), C.byref(fl)), "mi_prof_kernel_get")
        out.append({"kernel": name.value.decode(), "family": fam.value.decode(), "ms": ms.value, "launches": int(n.value),
                    "bytes": by.value, "flops": fl.value})
    return sorted(out, key=lambda k: -k["ms"])


def bench_conv_gemm(dtype: str, B: int, T: int, Cin: int, N: int, taps: int = 1, dil: int = 1, with_res: bool = False,
                    iters: int = 20) -> float:
    """Average ms per launch of the implicit-GEMM kernel on random device data (tuning hook)."""
    ms = C.c_double()
    check(load().mi_bench_conv_gemm(DTYPES[dtype], B, T, Cin, N, taps, dil, int(with_res), iters, C.byref(ms)),
          "mi_bench_conv_gemm")
    return ms.value


def device_pci_bus_id(device: int) -> str:
    buf = C.create_string_buffer(64)
    check(load().mi_device_pci_bus_id(int(device), buf, 64), "mi_device_pci_bus_id")
    return buf.value.decode()


def set_option(key: str, value: int) -> None:
    check(load().mi_set_option(key.encode(), int(value)), "mi_set_option")
