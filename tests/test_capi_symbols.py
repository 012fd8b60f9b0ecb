"""CPU: the C-ABI shared library loads and exports every symbol include/mi355tts.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from mi355tts import _lib
from mi355tts.config import BigVGANConfig
from mi355tts import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mi355tts.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    L = _lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mi355tts.h but not exported"


def test_param_count_matches_packer():
    L = _lib.load()
    for cfg in (BigVGANConfig(), BigVGANConfig.small()):
        ci = np.asarray(cfg.to_int_array(), dtype=np.int32)
        n = L.mi_bigvgan_param_count(_lib.i32p(ci), len(ci))
        assert n == sum(int(np.prod(s)) for _, s, _ in W.bigvgan_spec(cfg))
    assert L.mi_bigvgan_param_count(_lib.i32p(np.zeros(3, np.int32)), 3) < 0      # malformed cfg -> error code
    assert b"cfg" in L.mi_last_error()


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.MiError):
        _lib.init(0)


def test_host_paths_under_the_sanitizer_build():
    """the library's host code (loader, config parsing, blob validation, error paths) under
    AddressSanitizer + UBSan.  `build.py --sanitize` makes libmi355tts_asan.so; skipped when it has not been built.
    (GPU tests cannot run under it: the ROCm ASan runtime intercepts hsa_amd_memory_pool_allocate and needs
    xnack device builds, profiles/r4/asan_gpu_attempt.log.)"""
    import subprocess
    import sys
    lib = os.path.join(ROOT, "text-to-speech-tts-onnx_amd", "mi355tts", "libmi355tts_asan.so")
    if not os.path.exists(lib) or os.environ.get("MI355TTS_LIB"):
        pytest.skip("sanitizer build absent (python text-to-speech-tts-onnx_amd/build.py --sanitize)")
    r = subprocess.run([os.path.join(ROOT, "tools", "sanitize.sh"), sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_capi_symbols.py"), os.path.join(ROOT, "tests", "test_checkpoint.py"),
                        "-k", "not sanitizer"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert r.returncode == 0, out[-3000:]
