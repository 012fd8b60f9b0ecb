"""GPU: the exact-fit data-parallel fp32 linear kernel (csrc/gemm_x3d.hip, round 6) through the C-ABI's k = 1 convolution
(mi_conv1d: bias, plain row epilogue) against float64 numpy, and against the stream-K kernel it replaces where the tiling fits.
The fused epilogues (QKV + RoPE + V^T, AdaLN fold consumer / producer) are covered at full size by tests/test_gpu_f5.py
(test_full_size_fp32_against_reference_fixture runs N = 1126 = the 16 x 16-tile shape on this kernel)."""
import numpy as np
import pytest

from mi355tts import _lib, bigvgan

pytestmark = pytest.mark.gpu


def _case(T, Cin, Cout, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((1, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = np.einsum("oc,ct->ot", w[:, :, 0].astype(np.float64), x[0].astype(np.float64)) + b[:, None].astype(np.float64)
    return x, w, b, ref


# rows, K, N: one round of 256 tiles per tile width (192 | 128 | 64), a last row group cut by M, whole row groups, four rounds,
# K = 2048 (64 chunks), and a shape the planner must refuse (half a round: 49 % useful area) so that the launch stays on stream-K
@pytest.mark.parametrize("T,Cin,Cout", [(2252, 1024, 3072), (2252, 1024, 2048), (2252, 1024, 1024), (2252, 2048, 1024),
                                        (2100, 1024, 1024), (2304, 1024, 3072), (9008, 1024, 1024), (1126, 1024, 3072)])
def test_exact_fit_linear_against_float64_and_stream_k(T, Cin, Cout):
    x, w, b, ref = _case(T, Cin, Cout, 7)
    scale = np.abs(ref).max()
    outs = {}
    try:
        for on in (1, 0):
            _lib.set_option("gemm_x3d", on)
            y = bigvgan.conv1d(x, w, b, dtype="f32")[0].astype(np.float64)
            assert np.abs(y - ref).max() / scale < 1e-6, (on, np.abs(y - ref).max() / scale)      # fp16-pair products: 22-bit operands
            assert np.array_equal(y, bigvgan.conv1d(x, w, b, dtype="f32")[0].astype(np.float64))    # run to run
            outs[on] = y
    finally:
        _lib.set_option("gemm_x3d", 1)
    assert np.abs(outs[1] - outs[0]).max() / scale < 1e-6


def test_exact_fit_threshold_option():
    """gemm_x3d_min_eff: the planner takes the kernel from that share of useful tile area; at 100 nothing qualifies (every shape
    pays the 144-row rounding) and the results are the stream-K kernel's, bit for bit."""
    x, w, b, ref = _case(2252, 1024, 1024, 11)
    try:
        _lib.set_option("gemm_x3d", 0)
        sk = bigvgan.conv1d(x, w, b, dtype="f32")
        _lib.set_option("gemm_x3d", 1)
        _lib.set_option("gemm_x3d_min_eff", 100)
        assert np.array_equal(bigvgan.conv1d(x, w, b, dtype="f32"), sk)
        _lib.set_option("gemm_x3d_min_eff", 90)
        fit = bigvgan.conv1d(x, w, b, dtype="f32")
        assert not np.array_equal(fit, sk) and np.abs(fit[0].astype(np.float64) - ref).max() / np.abs(ref).max() < 1e-6
    finally:
        _lib.set_option("gemm_x3d", 1)
        _lib.set_option("gemm_x3d_min_eff", 90)
