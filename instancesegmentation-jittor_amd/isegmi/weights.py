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


def yolact_state_dict(seed=1234):
    rng = np.random.default_rng(seed)
    sd = {}
    resnet_state_dict(rng, sd, "backbone.")
    for i, cin in enumerate((2048, 1024, 512)):
        _conv_bias(rng, sd, "fpn.lat_layers.%d" % i, 256, cin, 1)
    for i in range(3):
        _conv_bias(rng, sd, "fpn.pred_layers.%d" % i, 256, 256, 3)
    for i in range(2):
        _conv_bias(rng, sd, "fpn.downsample_layers.%d" % i, 256, 256, 3)
    for i in (0, 2, 4, 8):
        _conv_bias(rng, sd, "proto_net.%d" % i, 256, 256, 3)
    _conv_bias(rng, sd, "proto_net.10", 32, 256, 1, gain=YOLACT_PROTO_GAIN)
    _conv_bias(rng, sd, "prediction_layers.0.upfeature.0", 256, 256, 3)
    _conv_bias(rng, sd, "prediction_layers.0.bbox_layer", 12, 256, 3, gain=YOLACT_LOC_GAIN)
    _conv_bias(rng, sd, "prediction_layers.0.conf_layer", 243, 256, 3, gain=YOLACT_CONF_GAIN)
    _conv_bias(rng, sd, "prediction_layers.0.mask_layer", 96, 256, 3, gain=YOLACT_MASK_GAIN)
    b = sd["prediction_layers.0.conf_layer.bias"].reshape(3, 81)
    b[:, 0] += YOLACT_BG_BIAS
    return sd


def to_krsc(w_oihw):
    return np.ascontiguousarray(np.transpose(np.asarray(w_oihw, np.float32), (0, 2, 3, 1)))


def fold_batchnorm(sd, prefix, eps=1e-5):
    """nn.BatchNorm2d eval (Yolact): scale = w/sqrt(var+eps); shift = b - mean*scale."""
    w = sd[prefix + ".weight"].astype(np.float32); b = sd[prefix + ".bias"].astype(np.float32)
    m = sd[prefix + ".running_mean"].astype(np.float32); v = sd[prefix + ".running_var"].astype(np.float32)
    scale = (w / np.sqrt(v + np.float32(eps))).astype(np.float32)
    return scale, (b - m * scale).astype(np.float32)


def fold_frozen_batchnorm(sd, prefix):
    """FrozenBatchNorm2d (maskrcnn-benchmark): scale = w*rsqrt(var) (no eps); shift = b - mean*scale."""
    w = sd[prefix + ".weight"].astype(np.float32); b = sd[prefix + ".bias"].astype(np.float32)
    m = sd[prefix + ".running_mean"].astype(np.float32); v = sd[prefix + ".running_var"].astype(np.float32)
    scale = (w * (np.float32(1.0) / np.sqrt(v))).astype(np.float32)
    return scale, (b - m * scale).astype(np.float32)
