"""HIP implicit-GEMM conv vs the CPU oracle: BIT-EXACT (k-ordered fmaf chain on both sides)."""
import numpy as np
import pytest

from oracle import ora

pytestmark = pytest.mark.gpu


def _rand(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


CASES = [
    # N, H, W, Cin, Cout, R, stride, pad
    (2, 19, 23, 32, 48, 3, 1, 1),
    (1, 35, 35, 64, 64, 1, 1, 0),
    (2, 35, 33, 64, 128, 3, 2, 1),
    (1, 18, 18, 256, 243, 3, 1, 1),
    (3, 9, 9, 128, 12, 3, 1, 1),
    (1, 40, 56, 256, 256, 1, 2, 0),
    (1, 7, 7, 256, 1024, 7, 1, 0),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 12, 13, 14])
def test_conv_bit_exact(ffi, case, tile):
    N, H, W, Cin, Cout, R, stride, pad = case
    rng = np.random.default_rng(hash(case) % (2**32))
    x = _rand(rng, (N, H, W, Cin))
    w = _rand(rng, (Cout, R, R, Cin), (2.0 / (R * R * Cin)) ** 0.5)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    sh = _rand(rng, (Cout,), 0.1)
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - R) // stride + 1
    res = _rand(rng, (N, Ho, Wo, Cout))
    # act 3 = LeakyReLU(0.1), act 4 = LeakyReLU(0.1) then + residual (the DarkNet block of yolact_darknet53_config)
    for act, use_res in [(1, True), (0, False), (2, False), (3, False), (4, True)]:
        ref = ora.conv2d(x, w, stride, pad, sc, sh, res if use_res else None, act)
        got = ffi.conv2d(x, w, stride, pad, sc, sh, res if use_res else None, act, tile)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), "max abs diff %g" % np.max(np.abs(got - ref))


# Round 6: FIXED-TREE SPLIT-K (conv tile 15, the engines' opt-in `conv_split_k`): the output is ((p0 + p1) + p2) + p3 of four k-ordered partial chains over
# equal ranges of the K chunks -- bit-exact against the oracle's restatement of exactly that sum (ora.conv2d(..., ksplit=4)), and NOT the default chain
SPLIT_CASES = [
    # N, H, W, Cin, Cout, R, stride, pad
    (1, 35, 35, 256, 256, 3, 1, 1),    # Yolact res4 conv2 at bs 1: 72 chunks, two channel groups of 128 (a split range ends inside a tap's group)
    (1, 18, 18, 512, 512, 3, 1, 1),    # res5 conv2: 144 chunks, four channel groups
    (1, 35, 35, 1024, 256, 1, 1, 0),   # res4 conv1: 32 chunks, plain (r, s, c) order
    (1, 25, 42, 2048, 512, 1, 1, 0),   # Mask R-CNN res5 conv1 at bs 1
    (2, 9, 9, 320, 40, 3, 2, 1),       # 90 chunks: ranges of 23 / 23 / 23 / 21, ragged channel groups 128 + 128 + 64, stride 2, Cout tail
    (1, 5, 7, 96, 48, 3, 1, 1),        # 27 chunks (a layer the engines' rule would not split: the kernel must still be right)
    (1, 6, 6, 64, 32, 1, 1, 0),        # 2 chunks over four sets: two sets have nothing to do
    (3, 11, 13, 128, 72, 3, 1, 1),     # 36 chunks, rows past M in the last tile, three images
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_split_k_bit_exact_against_the_oracle_split(ffi, case):
    N, H, W, Cin, Cout, R, stride, pad = case
    rng = np.random.default_rng(hash(case) % (2**32))
    x = _rand(rng, (N, H, W, Cin))
    w = _rand(rng, (Cout, R, R, Cin), (2.0 / (R * R * Cin)) ** 0.5)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    sh = _rand(rng, (Cout,), 0.1)
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - R) // stride + 1
    res = _rand(rng, (N, Ho, Wo, Cout))
    differs = False
    for act, use_res in [(1, True), (0, False)]:
        ref = ora.conv2d(x, w, stride, pad, sc, sh, res if use_res else None, act, ksplit=4)
        got = ffi.conv2d(x, w, stride, pad, sc, sh, res if use_res else None, act, 15)
        assert np.array_equal(got, ref), "max abs diff %g" % np.max(np.abs(got - ref))
        one = ora.conv2d(x, w, stride, pad, sc, sh, res if use_res else None, act)
        assert np.allclose(got, one, rtol=0, atol=2e-5 * max(1.0, np.abs(one).max()))   # the same sum, another association
        differs |= not np.array_equal(got, one)
    if R * R * (Cin // 32) >= 8:
        assert differs   # (it IS another evaluation: a test that passed with the split ignored would prove nothing)


def test_conv_split_rule_matches_the_oracle_rule(ffi):
    """isegmi_conv_split_qualifies == oracle.ora.conv_split_qualifies over the layer shapes of both models at bs 1 / 2 / 8"""
    import ctypes as C
    shapes = [(n, h, w, ci, co, r, st) for n in (1, 2, 8) for (h, w) in ((35, 35), (18, 18), (69, 69), (138, 138), (50, 84), (25, 42), (100, 168), (7, 7))
              for (ci, co, r, st) in ((256, 256, 3, 1), (1024, 256, 1, 1), (256, 1024, 1, 1), (512, 512, 3, 1), (2048, 512, 1, 1), (512, 2048, 1, 1), (128, 128, 3, 1),
                                      (512, 128, 1, 1), (256, 1024, 7, 1), (64, 64, 3, 1), (1024, 512, 1, 2))]
    n_true = 0
    for (n, h, w, ci, co, r, st) in shapes:
        pad = r // 2
        if h + 2 * pad < r:
            continue
        d = ffi.make_conv_desc(n, h, w, ci, co, r, r, st, pad, 0, 0)
        ho, wo = (h + 2 * pad - r) // st + 1, (w + 2 * pad - r) // st + 1
        got = ffi.lib().isegmi_conv_split_qualifies(C.byref(d))
        assert got == int(ora.conv_split_qualifies(n * ho * wo, co, r, r, ci)), (n, h, w, ci, co, r, st)
        n_true += got
    assert 0 < n_true < len(shapes)


# Round 3: the K walk (128-channel groups outermost) with a ragged last group, and the banded tile walk over the XCDs with a last band narrower
# than the others (conv_set_band picks 2 / 3 / 6 / 11 / 4-wide bands for these shapes; a wrong tile decode leaves tiles uncomputed or computed twice)
WALK_CASES = [
    # N, H, W, Cin, Cout, R, stride, pad, tiles
    (1, 14, 14, 320, 96, 3, 1, 1, (0, 3, 4, 5, 6, 10)),      # channel groups 128 + 128 + 64
    (2, 9, 9, 640, 40, 3, 2, 1, (0, 3, 4, 5)),               # five groups, stride 2
    (2, 12, 12, 256, 160, 1, 1, 0, (0, 3, 4, 5, 6, 10)),     # 5 Cout tiles of 32 in bands of 3
    (4, 12, 12, 256, 352, 1, 1, 0, (0, 4, 5, 6, 10)),        # 11 in bands of 6
    (2, 20, 20, 256, 672, 1, 1, 0, (0, 4, 5, 10)),           # 21 in bands of 11
    (4, 56, 56, 512, 416, 3, 1, 1, (0, 10, 13)),             # 1372 64x64 tiles, 7 Cout tiles in bands of 4 + 3, four channel groups
]


@pytest.mark.parametrize("case", WALK_CASES)
def test_conv_k_groups_and_tile_bands_bit_exact(ffi, case):
    N, H, W, Cin, Cout, R, stride, pad, tiles = case
    rng = np.random.default_rng(hash(case[:8]) % (2**32))
    x = _rand(rng, (N, H, W, Cin))
    w = _rand(rng, (Cout, R, R, Cin), (2.0 / (R * R * Cin)) ** 0.5)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    sh = _rand(rng, (Cout,), 0.1)
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - R) // stride + 1
    res = _rand(rng, (N, Ho, Wo, Cout))
    ref = ora.conv2d(x, w, stride, pad, sc, sh, res, 1)
    for tile in tiles:
        got = ffi.conv2d(x, w, stride, pad, sc, sh, res, 1, tile)
        assert np.array_equal(got, ref), "tile %d: max abs diff %g" % (tile, np.max(np.abs(got - ref)))


def test_stem_bit_exact(ffi):
    rng = np.random.default_rng(7)
    x3 = _rand(rng, (2, 70, 62, 3), 50.0)
    x = np.concatenate([x3, np.zeros((2, 70, 62, 1), np.float32)], -1)
    w3 = _rand(rng, (64, 7, 7, 3), 0.05)
    w = np.concatenate([w3, np.zeros((64, 7, 7, 1), np.float32)], -1)
    sc = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    sh = _rand(rng, (64,), 0.1)
    for tile in (0, 1, 2, 3):
        ref = ora.conv2d(x, w, 2, 3, sc, sh, None, 1)
        got = ffi.conv2d(x, w, 2, 3, sc, sh, None, 1, tile)
        assert np.array_equal(got, ref), "max abs diff %g" % np.max(np.abs(got - ref))


def test_detmath_bit_exact(ffi):
    rng = np.random.default_rng(3)
    x = np.concatenate([_rand(rng, (200000,), 10.0), np.linspace(-100, 100, 4001, dtype=np.float32),
                        np.array([0.0, -0.0, 0.625, -0.625, 88.5, -87.4, 44.5], np.float32)])
    for fn in (0, 1, 2):
        assert np.array_equal(ffi.map_f32(x, fn), ora.map_f32(x, fn)), fn
    xp = np.abs(x) + 1e-6
    assert np.array_equal(ffi.map_f32(xp, 3), ora.map_f32(xp, 3))


def test_spatial_bit_exact(ffi):
    rng = np.random.default_rng(5)
    a = _rand(rng, (2, 19, 23, 32))
    assert np.array_equal(ffi.maxpool(a, 3, 2, 1), ora.maxpool(a, 3, 2, 1))
    assert np.array_equal(ffi.maxpool(a, 1, 2, 0), ora.maxpool(a, 1, 2, 0))
    add = _rand(rng, (2, 35, 41, 32))
    assert np.array_equal(ffi.resize_bilinear(a, 35, 41, add, 0), ora.resize_bilinear(a, 35, 41, add, 0))
    assert np.array_equal(ffi.resize_bilinear(a, 38, 46, None, 1), ora.resize_bilinear(a, 38, 46, None, 1))
    lat = _rand(rng, (2, 38, 46, 32))
    assert np.array_equal(ffi.upsample_nearest2x_add(a, lat), ora.upsample_nearest2x_add(a, lat))


HYBRID_CASES = [
    # shapes whose 64x64-tile grid is larger than the 256 CUs and does not divide over them: the hybrid launch really splits
    (1, 184, 184, 32, 64, 1, 1, 0),     # 529 m-tiles x 1: 512 v2 tiles + 17 tiles' rows on 32x32 blocks
    (1, 150, 150, 64, 128, 3, 1, 1),    # 352 x 2: 256 m-tile rows on v2, 96 on the small blocks (3x3 taps, padding)
    (2, 101, 97, 64, 72, 3, 2, 1),      # stride 2, Cout not a multiple of 32, M = 2 * 51 * 49 = 4998 (M % 32 != 0): 79 x 2 < 256 -> plain v2 fallback
    (3, 83, 79, 32, 200, 1, 1, 0),      # 308 x 4 = 1232: 256 m-tile rows main, M % 64 != 0 in the tail
]


@pytest.mark.parametrize("case", HYBRID_CASES)
@pytest.mark.parametrize("tile", [13, 14])
def test_conv_hybrid_split_bit_exact(ffi, case, tile):
    """Tiles 13 / 14: v2 tiles on the rows that fill the CUs a whole number of times + 32x32 blocks on the left-over rows, one launch."""
    N, H, W, Cin, Cout, R, stride, pad = case
    rng = np.random.default_rng(hash(case) % (2**32))
    x = _rand(rng, (N, H, W, Cin))
    w = _rand(rng, (Cout, R, R, Cin), (2.0 / (R * R * Cin)) ** 0.5)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    sh = _rand(rng, (Cout,), 0.1)
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - R) // stride + 1
    res = _rand(rng, (N, Ho, Wo, Cout))
    for act, use_res in [(1, True), (0, False)]:
        ref = ora.conv2d(x, w, stride, pad, sc, sh, res if use_res else None, act)
        got = ffi.conv2d(x, w, stride, pad, sc, sh, res if use_res else None, act, tile)
        assert np.array_equal(got, ref)


def _member(rng, N, H, W, Cin, Cout, R, stride=1, pad=None, act=1, res=False, bn=True):
    pad = R // 2 if pad is None else pad
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, R, R, Cin)) * (2.0 / (R * R * Cin)) ** 0.5).astype(np.float32)
    it = dict(x=x, w=w, stride=stride, pad=pad, act=act)
    if bn:
        it["scale"] = rng.uniform(0.5, 1.5, Cout).astype(np.float32); it["shift"] = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    if res:
        ho, wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        it["residual"] = rng.standard_normal((N, ho, wo, Cout)).astype(np.float32)
    return it


@pytest.mark.parametrize("case", ["heads", "small", "mixed", "one"])
def test_conv_group_bit_exact(ffi, case):
    """isegmi_op_conv2d_group (several independent convolutions in ONE launch: csrc/conv_mfma.hip conv_group_kernel) against the oracle, member by member,
    bit for bit -- shared-weight heads over five pyramid levels (large enough that the big levels run on 64 x 64 tiles and the small ones on 32 x 32
    blocks inside the same launch), a group small enough to run on 32 x 32 blocks only, members that differ in everything (1x1 / 3x3 / stride 2,
    residual, no BN, Cout <= 32, ragged Cout), and a group of one."""
    rng = np.random.default_rng({"heads": 1, "small": 2, "mixed": 3, "one": 4}[case])
    if case == "heads":
        w = (rng.standard_normal((96, 3, 3, 64)) * 0.06).astype(np.float32)
        items = []
        for hw in (69, 35, 18, 9, 5):
            it = _member(rng, 4, hw, hw, 64, 96, 3)
            it["w"] = w
            items.append(it)
    elif case == "small":
        items = [_member(rng, 1, 9, 9, 64, 64, 3), _member(rng, 1, 5, 5, 64, 351, 3, act=0, bn=False), _member(rng, 2, 7, 11, 32, 40, 1)]
    elif case == "mixed":
        items = [_member(rng, 2, 40, 56, 64, 128, 3, res=True), _member(rng, 1, 33, 31, 128, 24, 1, act=0), _member(rng, 2, 40, 56, 64, 64, 3, stride=2),
                 _member(rng, 3, 50, 50, 32, 70, 1, bn=False), _member(rng, 1, 64, 64, 256, 256, 3)]
    else:
        items = [_member(rng, 2, 30, 30, 64, 64, 3, res=True)]
    got = ffi.conv2d_group(items)
    for g, it in zip(got, items):
        ref = ora.conv2d(it["x"], it["w"], it["stride"], it["pad"], it.get("scale"), it.get("shift"), it.get("residual"), it["act"])
        assert g.shape == ref.shape and np.array_equal(g, ref), (case, it["x"].shape, it["w"].shape)
