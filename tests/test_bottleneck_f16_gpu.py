"""Fused identity bottleneck (csrc/bottleneck_f16.hip, configs[4]) against (a) the three-launch fp16 path it replaces -- BIT-identical: the
fused kernel rounds t1 / t2 to fp16 where the three launches round them when they store them, walks K in the same (r, s, cin) order and
feeds the same 16-deep MFMA steps -- and (b) the oracle on fp16-rounded operands under the fp16 conv tolerance of test_conv_f16_gpu.py
applied three times over.  Shapes cover: tiles cut by the right / bottom image border, images smaller than one tile, several images,
and (flags = 1: an 8-block grid) blocks that walk many tiles, i.e. the loader stream crossing tile boundaries."""
import numpy as np
import pytest

from oracle import ora

pytestmark = pytest.mark.gpu


def _weights(rng, Cin, Cmid):
    def bn(c):
        return rng.uniform(0.5, 1.5, c).astype(np.float32), (rng.standard_normal(c) * 0.1).astype(np.float32)
    w1 = (rng.standard_normal((Cmid, 1, 1, Cin)) * (2.0 / Cin) ** 0.5).astype(np.float16).astype(np.float32)
    w2 = (rng.standard_normal((Cmid, 3, 3, Cmid)) * (2.0 / (9 * Cmid)) ** 0.5).astype(np.float16).astype(np.float32)
    w3 = (rng.standard_normal((Cin, 1, 1, Cmid)) * (2.0 / Cmid) ** 0.5).astype(np.float16).astype(np.float32)
    return w1, bn(Cmid), w2, bn(Cmid), w3, bn(Cin)


def _three_launches(ffi, x, w1, sb1, w2, sb2, w3, sb3, tile):
    t1 = ffi.conv2d_f16(x, w1, 1, 0, sb1[0], sb1[1], None, 1, tile)
    t2 = ffi.conv2d_f16(t1, w2, 1, 1, sb2[0], sb2[1], None, 1, tile)
    return ffi.conv2d_f16(t2, w3, 1, 0, sb3[0], sb3[1], x, 1, tile)


SHAPES = [(1, 8, 16), (1, 5, 9), (2, 19, 37), (1, 24, 48), (3, 33, 30), (1, 50, 84), (2, 9, 61)]


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("ch", [(256, 64), (512, 128)])
def test_fused_bottleneck_equals_three_launches(ffi, ch, shape, flags):
    Cin, Cmid = ch
    N, H, W = shape
    rng = np.random.default_rng(Cin * 7919 + H * 131 + W * 7 + N)
    x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)  # a block input is a ReLU output
    ws = _weights(rng, Cin, Cmid)
    got = ffi.bottleneck_f16(x, *ws, flags=flags)
    ref = _three_launches(ffi, x, *ws, tile=4)  # 64x64 generic tile: K walked (r, s, cin), one MFMA step per 16 k
    assert got.shape == ref.shape and got.dtype == np.float16
    assert np.array_equal(got, ref), "fused != three launches: %d of %d differ, max |d| %g" % (
        int((got != ref).sum()), got.size, float(np.abs(got.astype(np.float32) - ref.astype(np.float32)).max()))


@pytest.mark.parametrize("ch", [(256, 64), (512, 128)])
def test_fused_bottleneck_close_to_oracle(ffi, ch):
    Cin, Cmid = ch
    N, H, W = 2, 21, 35
    rng = np.random.default_rng(Cin + 5)
    x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)
    w1, sb1, w2, sb2, w3, sb3 = _weights(rng, Cin, Cmid)
    got = ffi.bottleneck_f16(x, w1, sb1, w2, sb2, w3, sb3).astype(np.float32)
    xf = x.astype(np.float32)
    t1 = ora.conv2d(xf, w1, 1, 0, sb1[0], sb1[1], None, 1).astype(np.float16).astype(np.float32)
    t2 = ora.conv2d(t1, w2, 1, 1, sb2[0], sb2[1], None, 1).astype(np.float16).astype(np.float32)
    ref = ora.conv2d(t2, w3, 1, 0, sb3[0], sb3[1], xf, 1).astype(np.float16).astype(np.float32)
    # TOLERANCE (stated): each of the three stages may flip a result by one fp16 ulp against the ordered-chain oracle (the f16 MFMA sums 16
    # products per instruction in its own order); a flipped t1 / t2 element moves the outputs it feeds by ~ulp * |w|: 4 ulp + 4e-3 absolute
    d = np.abs(got - ref)
    assert np.all(d <= np.abs(ref) * 2.0 ** -8 + 4e-3), float(d.max())
    assert np.mean(got == ref) >= 0.97


@pytest.mark.parametrize("flags", [0, 1])
@pytest.mark.parametrize("shape", SHAPES)
def test_fused_projection_bottleneck_equals_four_launches(ffi, shape, flags):
    """res2's FIRST block (64 -> 64 -> 64 -> 256 with the 1x1 projection shortcut 64 -> 256) in one launch against conv1, conv2, the projection
    and conv3 (+ the fp16 shortcut tensor as residual) as four launches: bit-identical."""
    N, H, W = shape
    rng = np.random.default_rng(H * 131 + W * 7 + N + 5)
    x = np.maximum(rng.standard_normal((N, H, W, 64)), 0).astype(np.float16)  # the max-pooled stem output
    def bn(c):
        return rng.uniform(0.5, 1.5, c).astype(np.float32), (rng.standard_normal(c) * 0.1).astype(np.float32)
    w1 = (rng.standard_normal((64, 1, 1, 64)) * (2.0 / 64) ** 0.5).astype(np.float16).astype(np.float32)
    w2 = (rng.standard_normal((64, 3, 3, 64)) * (2.0 / 576) ** 0.5).astype(np.float16).astype(np.float32)
    w3 = (rng.standard_normal((256, 1, 1, 64)) * (2.0 / 64) ** 0.5).astype(np.float16).astype(np.float32)
    wd = (rng.standard_normal((256, 1, 1, 64)) * (2.0 / 64) ** 0.5).astype(np.float16).astype(np.float32)
    sb1, sb2, sb3, sbd = bn(64), bn(64), bn(256), bn(256)
    got = ffi.bottleneck_ds_f16(x, w1, sb1, w2, sb2, w3, sb3, wd, sbd, flags=flags)
    t1 = ffi.conv2d_f16(x, w1, 1, 0, sb1[0], sb1[1], None, 1, 4)
    t2 = ffi.conv2d_f16(t1, w2, 1, 1, sb2[0], sb2[1], None, 1, 4)
    sc = ffi.conv2d_f16(x, wd, 1, 0, sbd[0], sbd[1], None, 0, 4)
    ref = ffi.conv2d_f16(t2, w3, 1, 0, sb3[0], sb3[1], sc, 1, 4)
    assert got.shape == ref.shape == (N, H, W, 256) and got.dtype == np.float16
    assert np.array_equal(got, ref), "fused != four launches: %d of %d differ, max |d| %g" % (
        int((got != ref).sum()), got.size, float(np.abs(got.astype(np.float32) - ref.astype(np.float32)).max()))
