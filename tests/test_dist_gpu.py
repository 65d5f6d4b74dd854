"""RCCL path with world_size 1 on the GPU box (the only size a 1-GPU box allows): device-side record packing
equals the host-side layout, and the all-gather round-trips it.  N>1 is covered by the gloo CPU test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_world1_gather_matches_host_pack(ffi):
    from isegmi.dist import RcclGather, pack_records, record_bytes, unpack_records
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform
    size, n = 200, 2
    net = Yolact(yolact_state_dict(1234), max_batch=n, input_size=size)
    rng = np.random.default_rng(3)
    x = fast_base_transform(rng.uniform(0, 255, (n, size, size, 3)).astype(np.float32))
    net(x)
    g = RcclGather(0, 1, RcclGather.unique_id(), record_bytes(n))
    g.gather_from(net)
    got = g.fetch()
    assert got.shape == (1, record_bytes(n))
    host = pack_records(net.fetch("det.count", n), net.fetch("det.box", n), net.fetch("det.score", n), net.fetch("det.class", n),
                        net.fetch("det.coeff", n))
    assert np.array_equal(got[0], host)
    rec = unpack_records(got[0], n)
    assert rec["count"].sum() > 0 and np.array_equal(rec["score"], net.fetch("det.score", n))
    # with the prototypes attached
    ph = net.fetch("proto", n).shape[1:3]
    g2 = RcclGather(0, 1, RcclGather.unique_id(), record_bytes(n, proto_hw=ph))
    g2.gather_from(net, with_proto=True)
    rec2 = unpack_records(g2.fetch()[0], n, proto_hw=ph)
    assert np.array_equal(rec2["proto"], net.fetch("proto", n))
    g.close(); g2.close(); net.close()


def test_rccl_world1_maskrcnn_records(ffi):
    from isegmi.dist import RcclGather, maskrcnn_record_bytes, pack_maskrcnn_records, unpack_maskrcnn_records
    from isegmi.maskrcnn import MaskRCNN, prepare_images
    from isegmi.weights import maskrcnn_state_dict
    rng = np.random.default_rng(2)
    x, hw = prepare_images([rng.uniform(0, 255, (200, 230, 3)).astype(np.float32)])
    model = MaskRCNN(maskrcnn_state_dict(1234), x.shape[1], x.shape[2], max_batch=1)
    model(x, hw)
    g = RcclGather(0, 1, RcclGather.unique_id(), maskrcnn_record_bytes(1))
    g.gather_from(model)
    got = g.fetch()[0]
    host = pack_maskrcnn_records(model.fetch("det.count", 1), model.fetch("det.box", 1), model.fetch("det.score", 1),
                                 model.fetch("det.label", 1), model.fetch("det.mask28", 1))
    assert np.array_equal(got, host)
    rec = unpack_maskrcnn_records(got, 1)
    assert rec["count"][0] > 0 and np.array_equal(rec["mask28"], model.fetch("det.mask28", 1))
    g.close(); model.close()


def test_rccl_world1_maskrcnn_c4_records(ffi):
    """The C4 predictor's 14x14 masks travel in the same record layout (M = 14)."""
    import dataclasses
    from isegmi.dist import RcclGather, maskrcnn_record_bytes, pack_maskrcnn_records, unpack_maskrcnn_records
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
    from isegmi.weights import maskrcnn_c4_state_dict
    rng = np.random.default_rng(20261003)
    x, hw = prepare_images([rng.uniform(0, 255, (250, 340, 3)).astype(np.float32)], 16)
    cfg = dataclasses.replace(MaskRCNNConfig.c4(), RPN_POST_NMS_TOP_N_TEST=300)
    model = MaskRCNN(maskrcnn_c4_state_dict(1234), x.shape[1], x.shape[2], cfg=cfg, max_batch=1)
    model(x, hw)
    g = RcclGather(0, 1, RcclGather.unique_id(), maskrcnn_record_bytes(1, M=14))
    g.gather_from(model)
    got = g.fetch()[0]
    host = pack_maskrcnn_records(model.fetch("det.count", 1), model.fetch("det.box", 1), model.fetch("det.score", 1),
                                 model.fetch("det.label", 1), model.fetch("det.mask14", 1))
    assert np.array_equal(got, host)
    rec = unpack_maskrcnn_records(got, 1, M=14)
    assert rec["count"][0] > 0 and np.array_equal(rec["mask28"], model.fetch("det.mask14", 1))
    g.close(); model.close()
