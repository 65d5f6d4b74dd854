"""The FPN top-down step of the fp16 path as one launch (UP2X residual mode of the persistent conv kernel, csrc/conv_mfma_f16.hip: lateral 1x1 conv with the
nearest-2x upsampled coarser level added in its epilogue, configs[4]) against the two launches it replaces -- the lateral conv, then
out = fp16(float(lat) + float(coarse[n, y >> 1, x >> 1])): BIT-identical (the lateral result is rounded to fp16 before the add, as the stored tensor is)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 100, 168, 512), (1, 200, 336, 256), (3, 25, 42, 1024), (2, 51, 85, 256), (1, 13, 21, 2048), (8, 50, 84, 1024)])
def test_fused_merge_equals_conv_then_upsample_add(ffi, shape):
    N, H, W, Cin = shape
    rng = np.random.default_rng(H * 131 + W + Cin)
    x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)
    w = (rng.standard_normal((256, 1, 1, Cin)) * (1.0 / Cin) ** 0.5).astype(np.float16).astype(np.float32)
    sc = np.ones(256, np.float32); sh = (rng.standard_normal(256) * 0.1).astype(np.float32)
    Hc, Wc = (H + 1) // 2, (W + 1) // 2
    coarse = rng.standard_normal((N, Hc, Wc, 256)).astype(np.float16)
    got = ffi.conv1x1_up2x_add_f16(x, w, sc, sh, coarse)
    lat = ffi.conv2d_f16(x, w, 1, 0, sc, sh, None, 0, 37)   # the same 192 x 256 persistent tile (one K order for a 1x1 anyway)
    yy = np.minimum(np.arange(H) >> 1, Hc - 1); xx = np.minimum(np.arange(W) >> 1, Wc - 1)
    ref = (lat.astype(np.float32) + coarse[:, yy][:, :, xx].astype(np.float32)).astype(np.float16)
    assert got.shape == ref.shape and got.dtype == np.float16
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), "merged != conv + upsample-add: %d of %d differ, max |d| %g" % (
        int((got != ref).sum()), got.size, float(np.abs(got.astype(np.float32) - ref.astype(np.float32)).max()))


@pytest.mark.parametrize("shape", [(2, 100, 168, 512), (2, 51, 85, 256), (1, 13, 21, 2048)])
def test_fused_merge_close_to_oracle(ffi, shape):
    """The merged FPN step against the ORACLE: lat = fp16(ora.conv2d 1x1 on the fp16 operands, fp32 accumulation), out = fp16(ora.upsample_nearest2x_add
    of the coarser level).  TOLERANCE (stated, as tests/test_conv_f16_gpu.py): lat within one fp16 ulp of the ordered chain (the f16 MFMA's own
    association) + 1e-3; the add of two fp16 values is exact in fp32 and rounded once, which can move the sum by one more ulp of the SUM's
    magnitude: |got - ref| <= 2^-10 (|lat| + |ref|) + 1e-3; >= 98 % exactly equal."""
    from oracle import ora
    N, H, W, Cin = shape
    rng = np.random.default_rng(H * 131 + W + Cin + 1)
    x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)
    w = (rng.standard_normal((256, 1, 1, Cin)) * (1.0 / Cin) ** 0.5).astype(np.float16).astype(np.float32)
    sc = np.ones(256, np.float32); sh = (rng.standard_normal(256) * 0.1).astype(np.float32)
    Hc, Wc = (H + 1) // 2, (W + 1) // 2
    coarse = rng.standard_normal((N, Hc, Wc, 256)).astype(np.float16)
    got = ffi.conv1x1_up2x_add_f16(x, w, sc, sh, coarse)
    lat = ora.conv2d(x.astype(np.float32), w, 1, 0, sc, sh, None, 0).astype(np.float16).astype(np.float32)
    ref = ora.upsample_nearest2x_add(coarse.astype(np.float32), lat).astype(np.float16)
    d = np.abs(got.astype(np.float32) - ref.astype(np.float32))
    assert got.shape == ref.shape and np.all(d <= (np.abs(lat) + np.abs(ref.astype(np.float32))) * 2.0 ** -10 + 1e-3), float(d.max())
    assert np.mean(got == ref) >= 0.98
