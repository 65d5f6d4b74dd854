"""The fused RPN head of the fp16 path (conv_f16_epilogue_head in csrc/conv_mfma_f16.hip: 3x3 conv + BN + ReLU with the cls + bbox 1x1 on the tile while it
is in LDS, configs[4]) against the two launches it replaces: BIT-identical with the 3x3 on the same row-strip tile (t is rounded to fp16 where the two-launch path stores it; the 1x1 walks the same
16 k-steps).  Levels too small for the 192 x 256 row-strip tile must report fused = 0 and launch nothing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng, N, H, W, Cin=256, cout2=15):
    x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)
    w = (rng.standard_normal((256, 3, 3, Cin)) * (2.0 / (9 * Cin)) ** 0.5).astype(np.float16).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 256).astype(np.float32); sh = (rng.standard_normal(256) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((cout2, 1, 1, 256)) * 0.05).astype(np.float16).astype(np.float32)
    sc2 = np.ones(cout2, np.float32); sh2 = (rng.standard_normal(cout2) * 0.1).astype(np.float32)
    return x, w, sc, sh, w2, sc2, sh2


@pytest.mark.parametrize("shape", [(2, 100, 168), (1, 200, 336), (2, 131, 197), (8, 50, 84), (1, 211, 333), (3, 67, 91)])
@pytest.mark.parametrize("cout2", [15, 3, 32])
def test_fused_head_equals_two_launches(ffi, shape, cout2):
    N, H, W = shape
    rng = np.random.default_rng(H * 131 + W + cout2)
    x, w, sc, sh, w2, sc2, sh2 = _case(rng, N, H, W, 256, cout2)
    got, fused = ffi.conv3x3_head_f16(x, w, sc, sh, w2, sc2, sh2)
    assert fused == (N * H * W > 127 * 192), "fused from half a round of 192-row tiles on"
    if not fused:
        return
    t = ffi.conv2d_f16(x, w, 1, 1, sc, sh, None, 1, 30)   # the row-strip tile the fusion runs on (K walked (r, cin, s); other tiles: another fp32 association)
    ref = ffi.conv2d_f16(t, w2, 1, 0, sc2, sh2, None, 0, 0, out_f32=True)
    assert got.shape == ref.shape == (N, H, W, cout2) and got.dtype == np.float32
    assert np.abs(ref).max() > 0.1
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "fused head != two launches: %d of %d differ, max |d| %g" % (
        int((got != ref).sum()), got.size, float(np.abs(got - ref).max()))


def test_small_levels_are_not_fused(ffi):
    rng = np.random.default_rng(5)
    x, w, sc, sh, w2, sc2, sh2 = _case(rng, 1, 13, 21)
    got, fused = ffi.conv3x3_head_f16(x, w, sc, sh, w2, sc2, sh2)
    assert not fused and got is None


@pytest.mark.parametrize("shape", [(2, 100, 168), (1, 131, 197)])
def test_fused_head_close_to_oracle(ffi, shape):
    """The fused RPN head against the ORACLE: t = fp16(relu(ora.conv2d 3x3 on the fp16 operands)), logits / deltas = ora.conv2d 1x1 on t in fp32.
    TOLERANCE (stated): t may differ from the ordered chain by one fp16 ulp per element (the f16 MFMA's own association of 16 products:
    tests/test_conv_f16_gpu.py); the 1x1 sums 256 products t * w2: an ulp flip of t[k] moves an output by 2^-10 |t[k] w2[k]|, so
    |got - ref| <= 2^-10 * sum_k |t[k] w2[k]| (every t flipped the same way: the worst case) + 2e-5 * max(1, |ref|max) for the fp32
    association of the 1x1 itself."""
    from oracle import ora
    N, H, W = shape
    rng = np.random.default_rng(H * 31 + W)
    x, w, sc, sh, w2, sc2, sh2 = _case(rng, N, H, W, 256, 15)
    got, fused = ffi.conv3x3_head_f16(x, w, sc, sh, w2, sc2, sh2)
    assert fused
    t = ora.conv2d(x.astype(np.float32), w, 1, 1, sc, sh, None, 1).astype(np.float16).astype(np.float32)
    ref = ora.conv2d(t, w2, 1, 0, sc2, sh2, None, 0)
    bound = ora.conv2d(t, np.abs(w2), 1, 0, None, None, None, 0) * 2.0 ** -10 + 2e-5 * max(1.0, float(np.abs(ref).max()))
    d = np.abs(got - ref)
    assert got.shape == ref.shape and np.all(d <= bound), float((d - bound).max())
    assert np.mean(d <= 2e-5 * max(1.0, float(np.abs(ref).max()))) >= 0.9   # and most outputs see no flipped t at all
