"""Synthetic (seeded) weights in the upstream state-dict naming, and BN folding.

BASELINE.json's configs are all "random weights" (no network for the .pth files the reference
lists at README.md:209-221, README.md:266).  Generator per SURVEY.md 8(d): conv/FC
N(0, sqrt(2/fan_in)); BN weight U(0.5,1.5), bias N(0,0.1), mean N(0,0.1), var U(0.5,1.5); the last BN
of every bottleneck is damped (x0.25) so the 16 residual adds do not blow activations up; the
predictor layers carry recorded gains/biases so a controlled number of detections pass the
score thresholds (otherwise NMS / mask stages would see nothing).
"""
import numpy as np


def _conv(rng, cout, cin, k, gain=1.0):
    std = gain * np.sqrt(2.0 / (cin * k * k))
    return (rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32)


def _bn(rng, sd, prefix, c, wscale=1.0):
    sd[prefix + ".weight"] = (rng.uniform(0.5, 1.5, c) * wscale).astype(np.float32)
    sd[prefix + ".bias"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    sd[prefix + ".running_mean"] = (rng.standard_normal(c) * 0.1).astype(np.float32)
    sd[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)


def _conv_bias(rng, sd, name, cout, cin, k, gain=1.0, bias_std=0.01):
    sd[name + ".weight"] = _conv(rng, cout, cin, k, gain)
    sd[name + ".bias"] = (rng.standard_normal(cout) * bias_std).astype(np.float32)


def resnet_state_dict(rng, sd, prefix, blocks=(3, 4, 6, 3), layer_fmt="layers.%d.%d", stem_conv="conv1", stem_bn="bn1"):
    sd[prefix + stem_conv + ".weight"] = _conv(rng, 64, 3, 7)
    _bn(rng, sd, prefix + stem_bn, 64)
    inpl = 64
    for li, nb in enumerate(blocks):
        planes = 64 << li
        for b in range(nb):
            nm = prefix + layer_fmt % (li, b)
            sd[nm + ".conv1.weight"] = _conv(rng, planes, inpl, 1); _bn(rng, sd, nm + ".bn1", planes)
            sd[nm + ".conv2.weight"] = _conv(rng, planes, planes, 3); _bn(rng, sd, nm + ".bn2", planes)
            sd[nm + ".conv3.weight"] = _conv(rng, planes * 4, planes, 1); _bn(rng, sd, nm + ".bn3", planes * 4, 0.25)
            if b == 0:
                sd[nm + ".downsample.0.weight"] = _conv(rng, planes * 4, inpl, 1)
                _bn(rng, sd, nm + ".downsample.1", planes * 4)
            inpl = planes * 4


# Recorded calibration of the Yolact predictor (see module docstring): found with the CPU oracle on
# the seeded uniform(0,255) 550x550 image so that roughly 100 detections per image survive.
YOLACT_CONF_GAIN = 0.056
YOLACT_BG_BIAS = 10.0
YOLACT_LOC_GAIN = 0.015
YOLACT_MASK_GAIN = 0.04
YOLACT_PROTO_GAIN = 0.06


def darknet53_state_dict(rng, sd, prefix="backbone."):
    """dbolya/yolact DarkNetBackbone([1, 2, 8, 8, 4]) names: _preconv.{0,1}; layers.L.0.{0,1} = the stride-2 3x3 + BN that opens
    a layer; layers.L.B.{conv1,bn1,conv2,bn2} = DarkNetBlock B (1x1 C -> C/2, 3x3 C/2 -> C, shortcut after the activation)."""
    sd[prefix + "_preconv.0.weight"] = _conv(rng, 32, 3, 3); _bn(rng, sd, prefix + "_preconv.1", 32)
    cin = 32
    for li, nb in enumerate((1, 2, 8, 8, 4)):
        ch = 32 << li
        nm = prefix + "layers.%d" % li
        sd[nm + ".0.0.weight"] = _conv(rng, ch * 2, cin, 3); _bn(rng, sd, nm + ".0.1", ch * 2)
        cin = ch * 2
        for b in range(1, nb + 1):
            sd["%s.%d.conv1.weight" % (nm, b)] = _conv(rng, ch, cin, 1); _bn(rng, sd, "%s.%d.bn1" % (nm, b), ch)
            sd["%s.%d.conv2.weight" % (nm, b)] = _conv(rng, cin, ch, 3); _bn(rng, sd, "%s.%d.bn2" % (nm, b), cin, 0.35)


def dcn_blocks(depth, dcn_layers, dcn_interval):
    """(layer, block) pairs whose 3x3 is a DCNv2 (dbolya/yolact ResNetBackbone._make_layer: the first block of a layer when
    dcn_layers >= blocks, block i > 0 when i + dcn_layers >= blocks and i % dcn_interval == 0)."""
    out = set()
    for li, nb in enumerate((3, 4, 23 if depth == 101 else 6, 3)):
        dl = dcn_layers[li]
        for b in range(nb):
            if (b == 0 and dl >= nb) or (b > 0 and b + dl >= nb and b % dcn_interval == 0):
                out.add((li, b))
    return out


YOLACT_DCN_OFFSET_STD = 0.6   # synthetic conv_offset_mask outputs: offsets of about +-1 pixel, mask logits around 0


def yolact_state_dict(seed=1234, depth=50, num_priors=3, dcn_layers=(0, 0, 0, 0), dcn_interval=1, maskiou=False, backbone="resnet"):
    """dbolya/yolact state-dict names; depth 50 = yolact_resnet50, 101 = yolact_base / yolact_im700.
    YOLACT++ (yolact_plus_*): num_priors=9, DCNv2 3x3s per dcn_layers / dcn_interval (conv2.weight/.bias and
    conv2.conv_offset_mask.weight/.bias), maskiou=True adds maskiou_net.{0,2,4,6,8,10}."""
    rng = np.random.default_rng(seed)
    sd = {}
    if backbone == "darknet53":  # yolact_darknet53_config: C3/C4/C5 have 256/512/1024 channels
        darknet53_state_dict(rng, sd, "backbone.")
        dcn_layers = (0, 0, 0, 0)
    else:
        resnet_state_dict(rng, sd, "backbone.", blocks=(3, 4, 23 if depth == 101 else 6, 3))
    for li, b in sorted(dcn_blocks(depth, dcn_layers, dcn_interval)):
        nm = "backbone.layers.%d.%d.conv2" % (li, b)
        planes = 64 << li
        sd[nm + ".bias"] = (rng.standard_normal(planes) * 0.05).astype(np.float32)
        # conv1 outputs (post BN+ReLU) are O(1): a 3x3 x planes fan-in with this std gives offsets / logits of about OFFSET_STD
        sd[nm + ".conv_offset_mask.weight"] = (rng.standard_normal((27, planes, 3, 3)) * (YOLACT_DCN_OFFSET_STD / np.sqrt(9.0 * planes))).astype(np.float32)
        sd[nm + ".conv_offset_mask.bias"] = (rng.standard_normal(27) * 0.1).astype(np.float32)
    for i, cin in enumerate((1024, 512, 256) if backbone == "darknet53" else (2048, 1024, 512)):
        _conv_bias(rng, sd, "fpn.lat_layers.%d" % i, 256, cin, 1)
    for i in range(3):
        _conv_bias(rng, sd, "fpn.pred_layers.%d" % i, 256, 256, 3)
    for i in range(2):
        _conv_bias(rng, sd, "fpn.downsample_layers.%d" % i, 256, 256, 3)
    for i in (0, 2, 4, 8):
        _conv_bias(rng, sd, "proto_net.%d" % i, 256, 256, 3)
    _conv_bias(rng, sd, "proto_net.10", 32, 256, 1, gain=YOLACT_PROTO_GAIN)
    _conv_bias(rng, sd, "prediction_layers.0.upfeature.0", 256, 256, 3)
    A = num_priors
    _conv_bias(rng, sd, "prediction_layers.0.bbox_layer", A * 4, 256, 3, gain=YOLACT_LOC_GAIN)
    _conv_bias(rng, sd, "prediction_layers.0.conf_layer", A * 81, 256, 3, gain=YOLACT_CONF_GAIN)
    _conv_bias(rng, sd, "prediction_layers.0.mask_layer", A * 32, 256, 3, gain=YOLACT_MASK_GAIN)
    b = sd["prediction_layers.0.conf_layer.bias"].reshape(A, 81)
    b[:, 0] += YOLACT_BG_BIAS
    if maskiou:  # FastMaskIoUNet: 1 -> 8 -> 16 -> 32 -> 64 -> 128 (3x3 stride 2, no padding, ReLU) -> 80 (1x1, ReLU) -> global max
        cin = 1
        for i, cout in enumerate((8, 16, 32, 64, 128)):
            _conv_bias(rng, sd, "maskiou_net.%d" % (2 * i), cout, cin, 3, bias_std=0.05)
            cin = cout
        _conv_bias(rng, sd, "maskiou_net.10", 80, 128, 1, bias_std=0.05)
        sd["maskiou_net.10.bias"] += np.float32(0.25)  # most class maps clear the final ReLU: non-degenerate IoU predictions
    return sd


def to_krsc(w_oihw):
    return np.ascontiguousarray(np.transpose(np.asarray(w_oihw, np.float32), (0, 2, 3, 1)))


def fold_batchnorm(sd, prefix, eps=1e-5):
    """nn.BatchNorm2d eval (Yolact): scale = w/sqrt(var+eps); shift = b - mean*scale."""
    w = sd[prefix + ".weight"].astype(np.float32); b = sd[prefix + ".bias"].astype(np.float32)
    m = sd[prefix + ".running_mean"].astype(np.float32); v = sd[prefix + ".running_var"].astype(np.float32)
    scale = (w / np.sqrt(v + np.float32(eps))).astype(np.float32)
    return scale, (b - m * scale).astype(np.float32)


def fold_frozen_batchnorm(sd, prefix, eps=0.0):
    """FrozenBatchNorm2d (maskrcnn-benchmark): scale = w*rsqrt(var) (no eps); shift = b - mean*scale.
    eps > 0 is the other side of the App. A.1 fork (FrozenBatchNorm2d variants that add an eps inside the rsqrt, e.g. detectron2's 1e-5):
    scale = w*rsqrt(var + eps)."""
    w = sd[prefix + ".weight"].astype(np.float32); b = sd[prefix + ".bias"].astype(np.float32)
    m = sd[prefix + ".running_mean"].astype(np.float32); v = sd[prefix + ".running_var"].astype(np.float32)
    if eps:
        v = (v + np.float32(eps)).astype(np.float32)
    scale = (w * (np.float32(1.0) / np.sqrt(v))).astype(np.float32)
    return scale, (b - m * scale).astype(np.float32)


# Recorded calibration of the Mask R-CNN predictors (found with the CPU oracle on the seeded image):
# enough RPN proposals survive NMS and roughly 100 detections per image pass SCORE_THRESH 0.05.
MRCNN_RPN_CLS_GAIN = 0.0012
MRCNN_RPN_BBOX_GAIN = 0.0001
MRCNN_CLS_GAIN = 0.0008
MRCNN_BG_BIAS = 4.5
MRCNN_BBOX_GAIN = 0.0002
MRCNN_MASK_LOGIT_GAIN = 0.0005


def maskrcnn_state_dict(seed=1234, depth=50):
    """maskrcnn-benchmark e2e_mask_rcnn_R_50/101_FPN state-dict names."""
    rng = np.random.default_rng(seed)
    sd = {}
    blocks = (3, 4, 23 if depth == 101 else 6, 3)
    # ResNet body: same tensors as torchvision but under backbone.body.{stem,layerN}
    tmp = {}
    resnet_state_dict(rng, tmp, "", blocks=blocks)
    for k, v in tmp.items():
        if k.startswith("conv1.") or k.startswith("bn1."):
            sd["backbone.body.stem." + k] = v
        else:  # layers.L.B.xxx -> layer{L+1}.B.xxx
            parts = k.split(".")
            sd["backbone.body.layer%d.%s" % (int(parts[1]) + 1, ".".join(parts[2:]))] = v
    for i, cin in enumerate((256, 512, 1024, 2048), 1):
        _conv_bias(rng, sd, "backbone.fpn.fpn_inner%d" % i, 256, cin, 1)
        _conv_bias(rng, sd, "backbone.fpn.fpn_layer%d" % i, 256, 256, 3)
    _conv_bias(rng, sd, "rpn.head.conv", 256, 256, 3)
    _conv_bias(rng, sd, "rpn.head.cls_logits", 3, 256, 1, gain=MRCNN_RPN_CLS_GAIN)
    _conv_bias(rng, sd, "rpn.head.bbox_pred", 12, 256, 1, gain=MRCNN_RPN_BBOX_GAIN)

    def fc(name, cout, cin, gain=1.0):
        sd[name + ".weight"] = (rng.standard_normal((cout, cin)) * gain * np.sqrt(2.0 / cin)).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(cout) * 0.01).astype(np.float32)
    fc("roi_heads.box.feature_extractor.fc6", 1024, 256 * 7 * 7)
    fc("roi_heads.box.feature_extractor.fc7", 1024, 1024)
    fc("roi_heads.box.predictor.cls_score", 81, 1024, MRCNN_CLS_GAIN)
    fc("roi_heads.box.predictor.bbox_pred", 324, 1024, MRCNN_BBOX_GAIN)
    sd["roi_heads.box.predictor.cls_score.bias"][0] += MRCNN_BG_BIAS
    for i in range(1, 5):
        _conv_bias(rng, sd, "roi_heads.mask.feature_extractor.mask_fcn%d" % i, 256, 256, 3)
    sd["roi_heads.mask.predictor.conv5_mask.weight"] = (rng.standard_normal((256, 256, 2, 2)) * np.sqrt(2.0 / 256)).astype(np.float32)
    sd["roi_heads.mask.predictor.conv5_mask.bias"] = (rng.standard_normal(256) * 0.01).astype(np.float32)
    _conv_bias(rng, sd, "roi_heads.mask.predictor.mask_fcn_logits", 81, 256, 1, gain=MRCNN_MASK_LOGIT_GAIN)
    return sd


# Recorded calibration of the R-50-C4 heads on the seeded uniform(0,255) image (same intent as above: about a thousand
# proposals out of the 6000 pre-NMS candidates and about a hundred detections above 0.05).
C4_RPN_CLS_GAIN = 0.0012
C4_RPN_BBOX_GAIN = 0.0001
C4_CLS_GAIN = 0.0011
C4_BG_BIAS = 3.6
C4_BBOX_GAIN = 0.0002
C4_MASK_LOGIT_GAIN = 0.0005


def maskrcnn_c4_state_dict(seed=1234):
    """maskrcnn-benchmark e2e_mask_rcnn_R_50_C4_1x state-dict names: backbone.body.{stem,layer1..3}; rpn.head on 1024 channels with
    15 anchors; roi_heads.box.feature_extractor.head.layer4 (conv5 head, shared with the mask branch); FastRCNNPredictor on 2048;
    MaskRCNNC4Predictor (ConvTranspose 2048 -> 256, 1x1 -> 81)."""
    rng = np.random.default_rng(seed)
    sd = {}
    tmp = {}
    resnet_state_dict(rng, tmp, "", blocks=(3, 4, 6, 3))
    for k, v in tmp.items():
        if k.startswith("conv1.") or k.startswith("bn1."):
            sd["backbone.body.stem." + k] = v
            continue
        parts = k.split(".")
        li = int(parts[1]) + 1
        rest = ".".join(parts[2:])
        if li <= 3:
            sd["backbone.body.layer%d.%s" % (li, rest)] = v
        else:
            sd["roi_heads.box.feature_extractor.head.layer4." + rest] = v
            sd["roi_heads.mask.feature_extractor.head.layer4." + rest] = v  # SHARE_BOX_FEATURE_EXTRACTOR: the same module
    _conv_bias(rng, sd, "rpn.head.conv", 1024, 1024, 3)
    _conv_bias(rng, sd, "rpn.head.cls_logits", 15, 1024, 1, gain=C4_RPN_CLS_GAIN)
    _conv_bias(rng, sd, "rpn.head.bbox_pred", 60, 1024, 1, gain=C4_RPN_BBOX_GAIN)

    def fc(name, cout, cin, gain=1.0):
        sd[name + ".weight"] = (rng.standard_normal((cout, cin)) * gain * np.sqrt(2.0 / cin)).astype(np.float32)
        sd[name + ".bias"] = (rng.standard_normal(cout) * 0.01).astype(np.float32)
    fc("roi_heads.box.predictor.cls_score", 81, 2048, C4_CLS_GAIN)
    fc("roi_heads.box.predictor.bbox_pred", 324, 2048, C4_BBOX_GAIN)
    sd["roi_heads.box.predictor.cls_score.bias"][0] += C4_BG_BIAS
    sd["roi_heads.mask.predictor.conv5_mask.weight"] = (rng.standard_normal((2048, 256, 2, 2)) * np.sqrt(2.0 / 2048)).astype(np.float32)
    sd["roi_heads.mask.predictor.conv5_mask.bias"] = (rng.standard_normal(256) * 0.01).astype(np.float32)
    _conv_bias(rng, sd, "roi_heads.mask.predictor.mask_fcn_logits", 81, 256, 1, gain=C4_MASK_LOGIT_GAIN)
    return sd
