"""Device front end: uint8 images go over PCIe, resize / normalise / pad run on the engine's stream.
The checker is the ORACLE's restatement of FastBaseTransform and build_transform + to_image_list (oracle/ora_ops.c:
ora_fast_base_transform, ora_build_transform; round 5 -- before that the device kernel was compared with the product's own numpy
transforms).  The host transforms of the product (isegmi/transforms.py) are compared with the same oracle in tests/test_oracle_cpu.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fetch_input(net, n, shape):
    from isegmi import _ffi
    out = np.empty((n,) + shape, np.float32)
    net.sync()
    _ffi.check(_ffi.lib().isegmi_d2h(out.ctypes.data_as(C.c_void_p), net.input_buffer(0).ptr, C.c_int64(out.nbytes)))
    return out


@pytest.mark.parametrize("hw", [(200, 200), (123, 171), (480, 640), (37, 29)])
def test_yolact_upload_u8_equals_oracle_transform(hw):
    from oracle import ora
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    net = Yolact(yolact_state_dict(1234), max_batch=3, input_size=200)
    imgs = rng.integers(0, 256, (3,) + hw + (3,), dtype=np.uint8)
    imgs[0, :3, :5] = 0; imgs[1, -2:, -2:] = 255   # saturated corners
    n = net.upload_u8(imgs)
    got = _fetch_input(net, n, (200, 200, 3))
    ref = ora.fast_base_transform(imgs, 200)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    net.close()


@pytest.mark.parametrize("hin,win,size", [(550, 550, 550), (480, 640, 550), (1080, 1920, 550), (33, 47, 700), (1, 1, 64), (2, 3, 64)])
def test_preprocess_op_equals_oracle_fast_base_transform(ffi, hin, win, size):
    """isegmi_op_preprocess_u8 through the C ABI against ora_fast_base_transform: the BASELINE size (550, identity and real resizes),
    im700, and degenerate one- and two-pixel sources (every tap clamps)"""
    from oracle import ora
    rng = np.random.default_rng(hin * 7 + win)
    x = rng.integers(0, 256, (2, hin, win, 3), dtype=np.uint8)
    got = ffi.preprocess_u8(x, size, size, size, size, ora.YOLACT_MEANS, ora.YOLACT_STD, True)
    assert np.array_equal(got, ora.fast_base_transform(x, size))
    got = ffi.preprocess_u8(x[:1], size, size, size, size, (0, 0, 0), (255, 255, 255), True)
    assert np.array_equal(got, ora.fast_base_transform(x[:1], size, darknet=True))


@pytest.mark.parametrize("hw", [(800, 1333), (1333, 800), (31, 65), (32, 64)])
def test_preprocess_op_equals_oracle_to_image_list(ffi, hw):
    """M1 at the BASELINE size (800 x 1333 -> 800 x 1344) and at sizes that do / do not need padding"""
    from oracle import ora
    rng = np.random.default_rng(hw[0])
    im = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    ref, rhw = ora.to_image_list([im])
    got = ffi.preprocess_u8(im[None], hw[0], hw[1], ref.shape[1], ref.shape[2], ora.PIXEL_MEAN, (1, 1, 1), False)
    assert np.array_equal(got, ref) and tuple(rhw[0]) == hw


def test_yolact_upload_u8_same_detections_as_host_path():
    from isegmi.transforms import yolact_transform
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact
    rng = np.random.default_rng(5)
    net = Yolact(yolact_state_dict(1234), max_batch=2, input_size=200)
    imgs = rng.integers(0, 256, (2, 240, 320, 3), dtype=np.uint8)
    host = net(np.concatenate([yolact_transform(im, 200) for im in imgs]))
    a = [(o["detection"] or {}).get("score") for o in host]
    n = net.upload_u8(imgs)
    net.forward_device(n); net.sync()
    cnt = net.fetch("det.count", n); score = net.fetch("det.score", n)
    for i in range(n):
        c = int(cnt[i])
        assert (a[i] is None and c == 0) or np.array_equal(a[i], score[i, :c])
    net.close()


def test_yolact_upload_u8_async_and_darknet_norm():
    from isegmi import _ffi
    from oracle import ora
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig
    rng = np.random.default_rng(9)
    net = Yolact(yolact_state_dict(1234), max_batch=2, input_size=200)
    pin = _ffi.PinnedBuffer((2, 150, 210, 3), np.uint8)
    pin.array[...] = rng.integers(0, 256, pin.array.shape, dtype=np.uint8)
    net.upload_u8_async(pin, 2, 150, 210)
    got = _fetch_input(net, 2, (200, 200, 3))
    ref = ora.fast_base_transform(pin.array, 200)
    assert np.array_equal(got, ref)
    net.close(); pin.free()
    dk = Yolact(yolact_state_dict(7, backbone="darknet53"), cfg=YolactConfig.darknet53(), max_batch=1, input_size=200)
    img = rng.integers(0, 256, (1, 90, 130, 3), dtype=np.uint8)
    dk.upload_u8(img)
    assert np.array_equal(_fetch_input(dk, 1, (200, 200, 3)), ora.fast_base_transform(img, 200, darknet=True))
    dk.close()


def test_maskrcnn_upload_u8_equals_oracle_to_image_list():
    from oracle import ora
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig
    from isegmi.weights import maskrcnn_state_dict
    rng = np.random.default_rng(3)
    ims = [rng.integers(0, 256, (200, 333, 3), dtype=np.uint8), rng.integers(0, 256, (256, 190, 3), dtype=np.uint8)]
    ref, hw = ora.to_image_list(ims)
    net = MaskRCNN(maskrcnn_state_dict(1234, 50), ref.shape[1], ref.shape[2], cfg=MaskRCNNConfig(depth=50), max_batch=2)
    n = net.upload_u8(ims)
    got = _fetch_input(net, n, ref.shape[1:])
    assert np.array_equal(got, ref) and np.array_equal(net._hw, hw)
    # and the forward sees the same proposals / detections as with the host-prepared batch
    net.forward_device(n); net.sync()
    s_dev = net.fetch("det.score", n).copy()
    net.upload(ref, hw); net.forward_device(n); net.sync()
    assert np.array_equal(s_dev, net.fetch("det.score", n))
    # images of one size go through ONE front-end launch for the batch (round 5): same result
    same = [rng.integers(0, 256, (200, 333, 3), dtype=np.uint8) for _ in range(2)]
    ref2, hw2 = ora.to_image_list(same)
    n = net.upload_u8(same, canvas=(ref.shape[1], ref.shape[2]))
    got2 = _fetch_input(net, n, ref.shape[1:])
    assert np.array_equal(got2[:, : ref2.shape[1], : ref2.shape[2]], ref2) and not got2[:, ref2.shape[1]:].any() and not got2[:, :, ref2.shape[2]:].any()
    net.close()


def test_maskrcnn_async_front_end_of_every_image_waits_for_its_upload():
    """A Mask R-CNN batch is ONE upload of the images back to back and one front-end launch PER IMAGE; every one of them has to run behind the
    H2D copy.  (Round 4 regression: the first launch marked the upload consumed, the second image's launch went to the main stream ahead of the
    copy and transformed whatever the staging buffer held before.)  Made deterministic: a 256 MB upload to a scratch buffer is queued on the copy
    stream first, so the batch's own copy is late; the staging buffer still holds the PREVIOUS batch."""
    from isegmi import _ffi
    from oracle import ora
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig
    from isegmi.weights import maskrcnn_state_dict
    import ctypes as C
    rng = np.random.default_rng(8)
    old = [rng.integers(0, 256, (200, 333, 3), dtype=np.uint8) for _ in range(2)]
    new = [rng.integers(0, 256, (200, 333, 3), dtype=np.uint8) for _ in range(2)]
    ref, hw = ora.to_image_list(new)
    net = MaskRCNN(maskrcnn_state_dict(1234, 50), ref.shape[1], ref.shape[2], cfg=MaskRCNNConfig(depth=50), max_batch=2)
    pin = _ffi.PinnedBuffer((2 * 200 * 333 * 3,), np.uint8)
    big_h = _ffi.PinnedBuffer((256 << 20,), np.uint8)
    big_d = _ffi.DeviceBuffer((256 << 20,), np.uint8)
    for rep in range(3):
        pin.array[...] = np.concatenate([im.reshape(-1) for im in old])
        net.upload_u8_async(pin, hw, 0); net.sync()                      # the staging buffer now holds the old batch
        pin.array[...] = np.concatenate([im.reshape(-1) for im in new])
        _ffi.check(_ffi.lib().isegmi_engine_upload_async(net._h, big_d.ptr, big_h.ptr, C.c_int64(big_h.nbytes)))   # keeps the copy stream busy
        net.upload_u8_async(pin, hw, 0)
        net.forward_device(2)                                             # (orders the main stream behind the front end, as in the product loop)
        net.sync()
        got = _fetch_input(net, 2, ref.shape[1:])
        assert np.array_equal(got[0], ref[0]), rep
        assert np.array_equal(got[1], ref[1]), "image 2 of the batch was transformed before its upload had landed (rep %d)" % rep
    pin.free(); big_h.free(); big_d.free()
    net.close()


def test_upload_u8_rejects_float_input():
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig
    from isegmi.weights import maskrcnn_state_dict, yolact_state_dict
    from isegmi.yolact import Yolact
    net = Yolact(yolact_state_dict(1234), max_batch=1, input_size=200)
    with pytest.raises(TypeError):
        net.upload_u8(np.zeros((1, 200, 200, 3), np.float32))
    net.close()
    m = MaskRCNN(maskrcnn_state_dict(1234, 50), 64, 64, cfg=MaskRCNNConfig(depth=50), max_batch=1)
    with pytest.raises(TypeError):
        m([np.zeros((64, 64, 3), np.float32)])  # image_hw forgotten: must not be truncated to bytes silently
    m.close()
