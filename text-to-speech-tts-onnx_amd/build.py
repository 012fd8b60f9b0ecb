#!/usr/bin/env python
"""Build libmi355tts.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python text-to-speech-tts-onnx_amd/build.py [--force] [-j N] [--variant NAME] [--tuning] [--sanitize]

--sanitize: a second library, libmi355tts_asan.so, whose HOST code is built with AddressSanitizer + UBSan (the kernels are
the product ones); run anything against it with tools/sanitize.sh (LD_PRELOAD of the clang ASan runtime + MI355TTS_LIB).
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "mi355tts", "libmi355tts.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-fno-gpu-rdc", "-DNDEBUG"] + os.environ.get("MI355TTS_EXTRA_FLAGS", "").split()


LINK_FLAGS: list = []
SAN_HOST = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "mi355tts.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str, force: bool, hdig: str) -> str:
    obj = os.path.join(OBJ, src + ".o")
    stamp = obj + ".stamp"
    key = hashlib.sha256(open(os.path.join(CSRC, src), "rb").read() + hdig.encode()).hexdigest()
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj
    cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    open(stamp, "w").write(key)
    return obj


def build(force: bool = False, jobs: int = 4, variant: str = "") -> str:
    """variant: a second build beside the product one (objects in build_<variant>/, library libmi355tts_<variant>.so —
    load it with MI355TTS_LIB=...): A/B measurements of kernel variants without touching the in-tree product library."""
    global OBJ, LIB
    if variant:
        OBJ = os.path.join(HERE, "build_" + variant)
        LIB = os.path.join(HERE, "mi355tts", f"libmi355tts_{variant}.so")
    os.makedirs(OBJ, exist_ok=True)
    hdig = _headers_digest()
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdig), srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *LINK_FLAGS, *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    j = 4
    if "-j" in sys.argv:
        j = int(sys.argv[sys.argv.index("-j") + 1])
    v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    if "--tuning" in sys.argv:          # the ablation instantiations of the kernels (MI355TTS_GEMM_DBG): tools/r3/*.sh use a `tune` variant
        FLAGS.append("-DMI355TTS_TUNING")
    if "--sanitize" in sys.argv:        # host side only: -Xarch_host keeps the device code as shipped
        v = "asan"
        for f in SAN_HOST:
            FLAGS.extend(["-Xarch_host", f])
        LINK_FLAGS.extend(["-fsanitize=address,undefined", "-shared-libsan"])
    print(build(force="--force" in sys.argv, jobs=j, variant=v))
