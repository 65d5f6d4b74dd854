"""fp16 MFMA conv (configs[4]) vs the oracle run on fp16-rounded operands with fp32 accumulation.
TOLERANCE (stated): the f16 MFMA sums 16 exact products per instruction in an unspecified order, so fp32
accumulators can differ from the ordered chain in the last bits; after rounding the result to fp16 we require
|got - ref| <= 1 fp16 ulp of the reference (2^-10 relative) + 1e-3 absolute, and >= 99 % exactly equal."""
import numpy as np
import pytest

from oracle import ora

pytestmark = pytest.mark.gpu

CASES = [(2, 19, 23, 64, 48, 3, 1, 1), (3, 14, 14, 128, 96, 3, 1, 1), (1, 30, 300, 64, 256, 3, 1, 1), (40, 9, 9, 64, 256, 3, 1, 1),
         (64, 7, 8, 64, 256, 3, 1, 1), (1, 35, 35, 64, 64, 1, 1, 0), (2, 35, 33, 128, 128, 3, 2, 1), (1, 18, 18, 256, 405, 1, 1, 0),
         (1, 7, 7, 256, 1024, 7, 1, 0), (1, 40, 56, 256, 256, 3, 1, 1)]


# 32-40: the persistent loader-wave kernels; + 2048 = their test hook, an 8-block grid, so that these small shapes make a block walk
# several tiles (the stream of the next tile entering the ring during the epilogue of this one)
PERSIST = [32, 34, 37, 39]
# round 5: the row-strip tile (40 = 30) and the persistent tiles (44 / 47 / 49 = 34 / 37 / 39) on v_mfma_f32_16x16x32_f16 (csrc/conv_mfma_f16_m16.hip):
# against the oracle here, and bit-identical to their 32 x 32 x 16 twins below
M16 = [40, 41, 44, 46, 47, 49, 2048 + 44, 2048 + 46, 2048 + 47, 2048 + 49]   # (41 / 46: the 144-row forms, 48-row wave tiles)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 20, 26, 27, 28, 29, 30, 31] + PERSIST + [2048 + t for t in PERSIST] + M16)
def test_conv_f16_close_to_oracle(ffi, case, tile):
    N, H, W, Cin, Cout, R, stride, pad = case
    strip = 26 <= tile <= 31 or tile in (40, 41)
    if strip and not (R == 3 and stride == 1 and pad == 1):
        pytest.skip("row-strip tiles are 3x3 / stride 1 / pad 1 only")
    if strip and W < 9:
        pytest.skip("row-strip tiles need <= 32 image-row segments per tile (the auto rule falls back to the generic kernel)")
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 32))
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, R, R, Cin)) * (2.0 / (R * R * Cin)) ** 0.5).astype(np.float16)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32); sh = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    Ho = (H + 2 * pad - R) // stride + 1; Wo = (W + 2 * pad - R) // stride + 1
    res = rng.standard_normal((N, Ho, Wo, Cout)).astype(np.float16)
    for act, use_res, f32 in ((1, True, False), (0, False, False), (0, False, True)):
        ref = ora.conv2d(x.astype(np.float32), w.astype(np.float32), stride, pad, sc, sh, res.astype(np.float32) if use_res else None, act)
        got = ffi.conv2d_f16(x, w.astype(np.float32), stride, pad, sc, sh, res if use_res else None, act, tile, out_f32=f32)
        if f32:
            assert got.dtype == np.float32 and np.max(np.abs(got - ref)) <= 2e-5 * max(1.0, np.abs(ref).max())
        else:
            ref16 = ref.astype(np.float16)
            d = np.abs(got.astype(np.float32) - ref16.astype(np.float32))
            assert np.all(d <= np.abs(ref16.astype(np.float32)) * 2.0 ** -10 + 1e-3)
            assert np.mean(got == ref16) >= 0.99


@pytest.mark.parametrize("tile", [0, 8])
@pytest.mark.parametrize("shape", [(2, 50, 70), (1, 37, 45), (1, 64, 33)])
def test_stem_f16_close_to_oracle(ffi, shape, tile):
    """fp16 stem: haloed fp16 image (exact) + 7x7/2 conv on the 8x8x4 zero-extended filter; same tolerance as above."""
    N, H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    x = rng.uniform(-120.0, 130.0, (N, H, W, 3)).astype(np.float32)
    w = (rng.standard_normal((64, 7, 7, 4)) * (2.0 / 147.0) ** 0.5).astype(np.float16).astype(np.float32)
    w[..., 3] = 0.0
    sc = rng.uniform(0.5, 1.5, 64).astype(np.float32); sh = (rng.standard_normal(64) * 0.1).astype(np.float32)
    got, halo = ffi.stem_f16(x, w, sc, sh, tile)
    x16 = x.astype(np.float16)
    assert halo.shape == (N, H + 6, (W + 7) & ~1, 4)
    assert np.array_equal(halo[:, 3:3 + H, 3:3 + W, :3], x16)
    mask = np.ones(halo.shape, bool); mask[:, 3:3 + H, 3:3 + W, :3] = False
    assert not halo[mask].any()
    x4 = np.concatenate([x16.astype(np.float32), np.zeros((N, H, W, 1), np.float32)], -1)
    ref16 = ora.conv2d(x4, w, 2, 3, sc, sh, None, 1).astype(np.float16)
    assert got.shape == ref16.shape
    d = np.abs(got.astype(np.float32) - ref16.astype(np.float32))
    assert np.all(d <= np.abs(ref16.astype(np.float32)) * 2.0 ** -10 + 1e-3)
    assert np.mean(got == ref16) >= 0.99


@pytest.mark.parametrize("case", [(2, 100, 168, 256, 256, 3, 1, 1), (1, 50, 84, 1024, 256, 1, 1, 0), (2, 50, 84, 256, 1024, 1, 1, 0), (1, 100, 168, 128, 512, 1, 1, 0),
                                  (1, 40, 56, 256, 256, 3, 1, 1), (8, 50, 84, 256, 256, 3, 1, 1), (8, 50, 84, 1024, 256, 1, 1, 0)])
def test_mfma_shape_does_not_change_results(ffi, case):
    """One v_mfma_f32_16x16x32_f16 sums its 32 products bit for bit as two chained v_mfma_f32_32x32x16_f16 do (tools/microbench/mfma_shape.hip), and the
    16 x 16 x 32 tiles walk K in the order of their 32 x 32 x 16 twins: every tile pair, and tile 0 under every isegmi_set_f16_mfma_shape setting, is
    BIT-identical -- with residual + ReLU, without, and with fp32 output"""
    N, H, W, Cin, Cout, R, stride, pad = case
    rng = np.random.default_rng(Cin + Cout + R)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, R, R, Cin)) * (2.0 / (R * R * Cin)) ** 0.5).astype(np.float16).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32); sh = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float16)
    assert ffi.get_f16_mfma_shape() == 3, "the default: row strips on 16 x 16 x 32 + the 144-row forms"
    for act, r, f32 in ((1, res, False), (0, None, False), (0, None, True)):
        run = lambda t: ffi.conv2d_f16(x, w, stride, pad, sc, sh, r, act, t, out_f32=f32)
        for a, b in (((30, 40), (40, 41)) if R == 3 else ((34, 44), (37, 47), (39, 49), (2048 + 37, 2048 + 47), (47, 46), (2048 + 47, 2048 + 46))):
            assert np.array_equal(run(a), run(b)), (a, b, act, f32)
        outs = []
        for shape in (0, 1, 2, 3):
            ffi.set_f16_mfma_shape(shape)
            try:
                outs.append(run(0))
            finally:
                ffi.set_f16_mfma_shape(3)
        assert all(np.array_equal(outs[0], o) for o in outs[1:])
