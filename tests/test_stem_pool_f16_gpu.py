"""The fused fp16 stem (csrc/stem_pool_f16.hip: conv 7x7/2 + BN + ReLU + max-pool 3x3/2 in one launch, configs[4]) against the two launches it
replaces: the fp16 stem conv (isegmi_op_conv2d_f16 on the haloed image) followed by a 3x3/2/1 max-pool -- BIT-identical (the pool of fp16 values
is exact, so the reference pool is taken in numpy over the conv launch's output).  Shapes: sizes that cut strips (30 pooled columns) and row
segments at every residue, odd sizes, images smaller than one strip, several images; flags 1 = an 8-block grid (blocks walk many units),
2 = the shortest units (a seam every three pooled rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pool(x):  # 3x3 / 2 / pad 1 over fp16 NHWC, -inf padding
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xp = np.full((N, 2 * Ho + 1, 2 * Wo + 1, C), -np.inf, np.float16)
    xp[:, 1:H + 1, 1:W + 1] = x
    out = np.full((N, Ho, Wo, C), -np.inf, np.float16)
    for dr in range(3):
        for dc in range(3):
            out = np.maximum(out, xp[:, dr:dr + 2 * Ho:2, dc:dc + 2 * Wo:2])
    return out


def _case(rng, N, H, W):
    x = rng.uniform(-120.0, 130.0, (N, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((64, 7, 7, 4)) * (2.0 / 147.0) ** 0.5).astype(np.float16).astype(np.float32)
    w[..., 3] = 0.0
    sc = (rng.uniform(0.5, 1.5, 64) / 60.0).astype(np.float32); sh = (rng.standard_normal(64) * 0.3).astype(np.float32)
    return x, w, sc, sh


SHAPES = [(1, 32, 32), (2, 50, 70), (1, 37, 45), (1, 64, 33), (1, 123, 251), (3, 96, 128), (1, 200, 488), (2, 17, 9), (1, 5, 5)]


@pytest.mark.parametrize("flags", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", SHAPES)
def test_fused_stem_equals_conv_then_pool(ffi, shape, flags):
    N, H, W = shape
    rng = np.random.default_rng(H * 1000 + W * 7 + N)
    x, w, sc, sh = _case(rng, N, H, W)
    conv, _ = ffi.stem_f16(x, w, sc, sh, 0)
    ref = _pool(conv)
    got = ffi.stem_pool_f16(x, w, sc, sh, flags)
    assert got.shape == ref.shape and got.dtype == np.float16
    assert np.isfinite(ref.astype(np.float32)).all() and (ref > 0).mean() > 0.3   # the case exercises the ReLU both ways
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), "fused stem != conv + pool: %d of %d differ, max |d| %g" % (
        int((got != ref).sum()), got.size, float(np.abs(got.astype(np.float32) - ref.astype(np.float32)).max()))


def test_fused_stem_at_the_bench_size(ffi):
    """configs[4]'s own canvas (800 x 1344; two of the eight images of a rank's batch): 12 strips x row segments chosen by the launcher's own cost model."""
    rng = np.random.default_rng(4)
    x, w, sc, sh = _case(rng, 2, 800, 1344)
    conv, _ = ffi.stem_f16(x, w, sc, sh, 0)
    got = ffi.stem_pool_f16(x, w, sc, sh, 0)
    assert got.shape == (2, 200, 336, 64)
    assert np.array_equal(got.view(np.uint16), _pool(conv).view(np.uint16))


@pytest.mark.parametrize("shape", [(2, 50, 70), (1, 123, 251), (1, 64, 33)])
def test_fused_stem_close_to_oracle(ffi, shape):
    """The fused launch against the ORACLE (not another HIP launch): ora.conv2d on the fp16-rounded image (fourth channel zero) with fp32
    accumulation, rounded to fp16 where the engine stores fp16, then ora.maxpool 3x3/2/1 (exact on fp16 values).  TOLERANCE (stated, as
    tests/test_conv_f16_gpu.py): the f16 MFMA sums 16 products per instruction in its own order, so a conv output may differ from the ordered
    chain by one fp16 ulp (2^-10 relative) + 1e-3 absolute; a max over nine such values inherits the bound; >= 99 % exactly equal."""
    from oracle import ora
    N, H, W = shape
    rng = np.random.default_rng(H * 77 + W)
    x, w, sc, sh = _case(rng, N, H, W)
    got = ffi.stem_pool_f16(x, w, sc, sh, 0)
    x4 = np.concatenate([x.astype(np.float16).astype(np.float32), np.zeros((N, H, W, 1), np.float32)], -1)
    conv16 = ora.conv2d(x4, w, 2, 3, sc, sh, None, 1).astype(np.float16)
    ref = ora.maxpool(conv16.astype(np.float32), 3, 2, 1).astype(np.float16)
    assert got.shape == ref.shape
    d = np.abs(got.astype(np.float32) - ref.astype(np.float32))
    assert np.all(d <= np.abs(ref.astype(np.float32)) * 2.0 ** -10 + 1e-3), float(d.max())
    assert np.mean(got == ref) >= 0.99
