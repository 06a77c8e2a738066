"""Oracle (test infrastructure only): ResNet-50 / DenseNet-121 YOLO backbones.

Functional torch-CPU fp32 restatement over a flat ``{state_dict key: tensor}``
parameter dict, following
  * backbones/OriginResNet.py:69-107 (Bottleneck), :112-195 (ResNet, head
    ``layer6`` 1x1 2048->B*5+C, ``bn_end``, sigmoid, NHWC permute), :220-231;
  * backbones/OriginDenseNet.py:19-54 (_DenseLayer/_DenseBlock/_Transition),
    :57-129 (DenseNet), :149-164 (densenet121: S=7 -> blocks (6,12,24,16,16),
    S=14 -> (6,12,24,16)).
The conv/BN/pool arithmetic itself lives in PyTorch ATen (not in the
reference tree); this oracle pins it to torch 2.10 CPU fp32 ``F.conv2d`` /
``F.batch_norm`` / ``F.max_pool2d`` / ``F.avg_pool2d``.

The key order of ``*_param_shapes`` is the reference's ``state_dict()`` order
(checked against tests/golden/state_dict_keys.json).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

RESNET50_LAYERS = (3, 4, 6, 3)


# ----------------------------------------------------------------------------
# parameter inventories
# ----------------------------------------------------------------------------
def _bn_entries(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)
    d[name + ".running_mean"] = (c,)
    d[name + ".running_var"] = (c,)
    d[name + ".num_batches_tracked"] = ()


def resnet50_param_shapes(S=7, B=2, C=20):
    """OriginResNet.py:112-134, :155-171 -> OrderedDict key -> shape."""
    d = OrderedDict()
    d["conv1.weight"] = (64, 3, 7, 7)
    _bn_entries(d, "bn1", 64)
    inpl = 64
    stages = [("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2), ("layer4", 512, 3, 2)]
    if S == 7:
        stages.append(("layer5", 512, 3, 2))                     # :131-132
    for name, planes, blocks, stride in stages:
        for i in range(blocks):
            p = "%s.%d" % (name, i)
            d[p + ".conv1.weight"] = (planes, inpl, 1, 1)
            _bn_entries(d, p + ".bn1", planes)
            d[p + ".conv2.weight"] = (planes, planes, 3, 3)
            _bn_entries(d, p + ".bn2", planes)
            d[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
            _bn_entries(d, p + ".bn3", planes * 4)
            if i == 0 and (stride != 1 or inpl != planes * 4):    # :159
                d[p + ".downsample.0.weight"] = (planes * 4, inpl, 1, 1)
                _bn_entries(d, p + ".downsample.1", planes * 4)
            inpl = planes * 4
    d["layer6.weight"] = (B * 5 + C, 2048, 1, 1)                  # :133
    _bn_entries(d, "bn_end", B * 5 + C)
    return d


def densenet121_block_config(S=7):
    return (6, 12, 24, 16, 16) if S == 7 else (6, 12, 24, 16)    # OriginDenseNet.py:159-161


def densenet121_param_shapes(S=7, B=2, C=20, growth=32, bn_size=4, init=64):
    """OriginDenseNet.py:76-102."""
    d = OrderedDict()
    d["features.conv0.weight"] = (init, 3, 7, 7)
    _bn_entries(d, "features.norm0", init)
    nf = init
    cfg = densenet121_block_config(S)
    for bi, nl in enumerate(cfg):
        for li in range(nl):
            p = "features.denseblock%d.denselayer%d" % (bi + 1, li + 1)
            cin = nf + li * growth
            _bn_entries(d, p + ".norm1", cin)
            d[p + ".conv1.weight"] = (bn_size * growth, cin, 1, 1)
            _bn_entries(d, p + ".norm2", bn_size * growth)
            d[p + ".conv2.weight"] = (growth, bn_size * growth, 3, 3)
        nf += nl * growth
        if bi != len(cfg) - 1:
            p = "features.transition%d" % (bi + 1)
            _bn_entries(d, p + ".norm", nf)
            d[p + ".conv.weight"] = (nf // 2, nf, 1, 1)
            nf //= 2
    _bn_entries(d, "features.norm5", nf)
    d["layer6.weight"] = (B * 5 + C, 1024, 1, 1)                   # :101
    _bn_entries(d, "bn_end", B * 5 + C)
    return d


def init_params(shapes, kind, seed=0):
    """Seeded reference-style init (NOT the reference's RNG stream).

    resnet: kaiming normal fan_out/relu (OriginResNet.py:138-143);
    densenet: kaiming normal fan_in, a=0 (OriginDenseNet.py:105-110);
    BN weight 1, bias 0, running_mean 0, running_var 1.
    """
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            P[k] = torch.zeros((), dtype=torch.long)
        elif len(shp) == 4:
            fan = shp[0] * shp[2] * shp[3] if kind == "resnet" else shp[1] * shp[2] * shp[3]
            P[k] = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan)
        elif k.endswith("running_var") or k.endswith(".weight"):
            P[k] = torch.ones(shp)
        else:
            P[k] = torch.zeros(shp)
    return P


# ----------------------------------------------------------------------------
# functional forward
# ----------------------------------------------------------------------------
# ``q`` (optional) is a storage quantiser applied wherever the HIP path stores a tensor in bf16
# (input image, weights, raw conv outputs, post-activation tensors).  With q=None this is the plain
# fp32 restatement; with q=bf16_ste the oracle emulates the bf16 storage points so whole-network
# comparisons are not dominated by rounding noise amplified through ~60 small-batch BatchNorms.
def bf16_ste(t):
    """bf16 round-trip with a straight-through gradient."""
    return t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()


def _q(t, q):
    return t if q is None else q(t)


def _bn(x, P, name, training, momentum=0.1, eps=1e-5):
    rm, rv = P[name + ".running_mean"], P[name + ".running_var"]
    return F.batch_norm(x, rm, rv, P[name + ".weight"], P[name + ".bias"], training, momentum, eps)


def bottleneck(x, P, p, stride, training=True, q=None, qconv=None):
    """OriginResNet.py:87-107 (stride on the 3x3, :79).  ``qconv`` (optional) replaces each convolution:
    ``qconv(conv_fn, x, w_master, w_stored)`` -- used to emulate a forward GEMM that reads lower-precision copies of its
    operands while the backward keeps the stored ones (oracle/fp8.py::fp8_forward_ste)."""
    w = lambda k: _q(P[p + k], q)

    def conv(inp, wk, **kw):
        if qconv is not None:
            return qconv(lambda a, b: F.conv2d(a, b, **kw), inp, P[p + wk], w(wk))
        return F.conv2d(inp, w(wk), **kw)
    out = _q(F.relu(_bn(_q(conv(x, ".conv1.weight"), q), P, p + ".bn1", training)), q)
    out = _q(F.relu(_bn(_q(conv(out, ".conv2.weight", stride=stride, padding=1), q), P, p + ".bn2", training)), q)
    out = _bn(_q(conv(out, ".conv3.weight"), q), P, p + ".bn3", training)
    if (p + ".downsample.0.weight") in P:
        idt = _bn(_q(conv(x, ".downsample.0.weight", stride=stride), q), P, p + ".downsample.1", training)
    else:
        idt = x
    return _q(F.relu(out + idt), q)


def resnet50_forward(x, P, S=7, training=True, q=None, qconv=None):
    """OriginResNet.py:173-195.  x [N,3,H,W] -> [N,H/64 or H/32, ., B*5+C].  ``qconv``: see ``bottleneck`` (the stem and
    the head are left alone)."""
    x = _q(F.conv2d(_q(x, q), _q(P["conv1.weight"], q), stride=2, padding=3), q)
    x = _q(F.relu(_bn(x, P, "bn1", training)), q)
    x = F.max_pool2d(x, 3, 2, 1)
    stages = [("layer1", 3, 1), ("layer2", 4, 2), ("layer3", 6, 2), ("layer4", 3, 2)]
    if S == 7:
        stages.append(("layer5", 3, 2))
    for name, blocks, stride in stages:
        for i in range(blocks):
            x = bottleneck(x, P, "%s.%d" % (name, i), stride if i == 0 else 1, training, q, qconv)
    x = _q(F.conv2d(x, _q(P["layer6.weight"], q)), q)
    x = _bn(x, P, "bn_end", training)
    return torch.sigmoid(x).permute(0, 2, 3, 1)


def dense_layer(x, P, p, training=True, q=None):
    """OriginDenseNet.py:19-36: BN-ReLU-1x1(->128)-BN-ReLU-3x3(->32), cat."""
    h = _q(F.conv2d(_q(F.relu(_bn(x, P, p + ".norm1", training)), q), _q(P[p + ".conv1.weight"], q)), q)
    h = _q(F.conv2d(_q(F.relu(_bn(h, P, p + ".norm2", training)), q), _q(P[p + ".conv2.weight"], q), padding=1), q)
    return torch.cat([x, h], 1)


def transition(x, P, p, training=True, q=None):
    """OriginDenseNet.py:47-54."""
    h = _q(F.conv2d(_q(F.relu(_bn(x, P, p + ".norm", training)), q), _q(P[p + ".conv.weight"], q)), q)
    return _q(F.avg_pool2d(h, 2, 2), q)


def densenet121_forward(x, P, S=7, training=True, q=None):
    """OriginDenseNet.py:114-129."""
    x = _q(F.conv2d(_q(x, q), _q(P["features.conv0.weight"], q), stride=2, padding=3), q)
    x = _q(F.relu(_bn(x, P, "features.norm0", training)), q)
    x = F.max_pool2d(x, 3, 2, 1)
    cfg = densenet121_block_config(S)
    for bi, nl in enumerate(cfg):
        for li in range(nl):
            x = dense_layer(x, P, "features.denseblock%d.denselayer%d" % (bi + 1, li + 1), training, q)
        if bi != len(cfg) - 1:
            x = transition(x, P, "features.transition%d" % (bi + 1), training, q)
    x = _q(F.relu(_bn(x, P, "features.norm5", training)), q)
    x = _q(F.conv2d(x, _q(P["layer6.weight"], q)), q)
    x = _bn(x, P, "bn_end", training)
    return torch.sigmoid(x).permute(0, 2, 3, 1)
