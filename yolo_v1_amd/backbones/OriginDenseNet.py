"""Drop-in for the reference's ``backbones/OriginDenseNet.py`` on MI355X.

``densenet121(pretrained=False, S=7)`` keeps the reference's state_dict keys
(``features.{conv0,norm0,denseblock{k}.denselayer{i}.{norm1,conv1,norm2,conv2},
transition{k}.{norm,conv},norm5}``, ``layer6``, ``bn_end`` -- OriginDenseNet.py:76-102) and the
block configuration quirk S=7 -> (6,12,24,16,16), S=14 -> (6,12,24,16) (:159-161).

MI355X-first differences to how the reference executes it:
  * a dense block owns ONE NHWC buffer of its final width; each layer's 3x3 conv writes its 32
    new channels straight into its slice, so ``torch.cat`` (:36), which re-copies the growing map
    in every layer (O(L^2) traffic), does not exist;
  * batch statistics of a channel do not depend on which BatchNorm reads it, so they are
    computed ONCE when the channel is produced (conv epilogue) and kept in a per-block table that
    every later ``norm1`` / transition ``norm`` / ``norm5`` finalises with its own gamma/beta;
  * the backward accumulates the gradient of the shared buffer in place (one gradient buffer per
    block), walking the layers in reverse.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _lib, ops
from ..engine import ConvParam, HipBackbone, make_bn

__all__ = ['DenseNet', 'densenet121']


class _DenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = make_bn(cin)
        self.conv1 = ConvParam(cin, bn_size * growth, 1)
        self.norm2 = make_bn(bn_size * growth)
        self.conv2 = ConvParam(bn_size * growth, growth, 3, 1, 1)


class _DenseBlock(nn.Module):
    def __init__(self, num_layers, cin, bn_size, growth):
        super().__init__()
        for i in range(num_layers):
            self.add_module('denselayer%d' % (i + 1), _DenseLayer(cin + i * growth, growth, bn_size))

    def layers(self):
        return list(self.children())


class _Transition(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = make_bn(cin)
        self.conv = ConvParam(cin, cout, 1)


class DenseNet(HipBackbone):
    def __init__(self, growth_rate=32, block_config=(6, 12, 24, 16), num_init_features=64, bn_size=4, B=2, S=7,
                 num_classes=20):
        super().__init__()
        self.growth = growth_rate
        self.out_channels = B * 5 + num_classes
        feats = OrderedDict()
        feats['conv0'] = ConvParam(3, num_init_features, 7, 2, 3)
        feats['norm0'] = make_bn(num_init_features)
        nf = num_init_features
        self._plan = []                         # ("block", name, nf_in, n_layers) / ("trans", name, cin)
        for i, nl in enumerate(block_config):
            feats['denseblock%d' % (i + 1)] = _DenseBlock(nl, nf, bn_size, growth_rate)
            self._plan.append(("block", 'denseblock%d' % (i + 1), nf, nl))
            nf += nl * growth_rate
            if i != len(block_config) - 1:
                feats['transition%d' % (i + 1)] = _Transition(nf, nf // 2)
                self._plan.append(("trans", 'transition%d' % (i + 1), nf))
                nf //= 2
        feats['norm5'] = make_bn(nf)
        self.features = nn.Sequential(feats)
        self.layer6 = ConvParam(nf, self.out_channels, 1)          # nf == 1024 for densenet121, the reference hard-wires it (:101)
        self.bn_end = make_bn(self.out_channels)
        for m in self.modules():                                    # :105-110
            if isinstance(m, ConvParam):
                nn.init.kaiming_normal_(m.weight)

    def layer_forward(self, layer, buf, table, cin, norm, train):
        """One _DenseLayer (OriginDenseNet.py:19-36) on the block buffer: reads channels [0, cin) of ``buf``, writes its
        ``growth`` new channels at [cin, cin+growth) in place (the reference's torch.cat, :36) and merges their batch
        statistics into ``table`` (the shared [1][2][Ctot] sum / sum-of-squares rows every later norm1 reads)."""
        dev = buf.t.device
        N, h, w = buf.N, buf.H, buf.W
        w1, w2 = self.cw(layer.conv1), self.cw(layer.conv2)
        xin = buf.window(0, cin)
        t1 = ops.new_act(N, h, w, cin, dev)
        st1 = norm(table, buf.npix, layer.norm1, cin, apply=(xin, t1))
        t2 = ops.new_act(N, h, w, w1.Opad, dev)
        if not train and self.fused_eval:
            # eval(): norm2 + ReLU ride in the 1x1 convolution's epilogue (norm1 acts on the concat input
            # and stays a separate pass)
            y1 = st2 = None
            ops.conv_fwd_bn_act(t1, w1, t2, ops.bn_eval_state(layer.norm2), relu=True)
        else:
            y1 = ops.new_act(N, h, w, w1.Opad, dev)
            st2 = norm(ops.conv_fwd(t1, w1, y1, train), y1.npix, layer.norm2, apply=(y1, t2))
        stats = ops.conv_fwd(t2, w2, buf.window(cin, self.growth), train)
        if train:
            # the new features' statistic rows join the shared table inside the NEXT BatchNorm's finalize launch (the next
            # layer's norm1, the transition's norm or norm5 -- each reads the table over at least these channels)
            self._pending_seg = (stats, cin)
        return (layer, cin, st1, t1, y1, st2, t2)

    def transition_forward(self, tr, buf, table, norm):
        """_Transition (OriginDenseNet.py:47-54) up to the 1x1 convolution; the caller pools ``yc`` into the next block."""
        dev = buf.t.device
        cin = buf.C
        wc = self.cw(tr.conv)
        t = ops.new_act(buf.N, buf.H, buf.W, cin, dev)
        st = norm(table, buf.npix, tr.norm, cin, apply=(buf, t))
        yc = ops.new_act(buf.N, buf.H, buf.W, wc.Opad, dev)
        ops.conv_fwd(t, wc, yc, False)
        return ("trans", tr, buf, st, t, yc)

    def layer_backward(self, lrec, buf, G, grads, side, K=None, owed=False):
        """Backward of one _DenseLayer: ``G`` holds the gradient of the block buffer; the layer's own slice is complete
        (every later layer has added to it), its input gradient is accumulated into G[..., :cin].
        ``K`` ([2][Ctot] fp32, ops.BN_DEFERRED): norm1's backward is deferred -- conv1's data gradient adds scale * masked
        gradient to G in its epilogue; the mean terms (affine in x per channel, coefficients in K) are subtracted by the NEXT
        data gradient into G -- every layer covers all channels below it -- and, for this layer's own 32-channel slice, which
        no later launch touches, here, right before it is consumed (``owed``: K holds something)."""
        (layer, cin, st1, t1, y1, st2, t2) = lrec
        dev = G.t.device
        N = G.N
        w1, w2 = self.cw(layer.conv1), self.cw(layer.conv2)
        dy2 = G.window(cin, self.growth)
        if K is not None and owed:
            ops.bn_deferred_fix(dy2, buf.window(cin, self.growth), K[:, cin:cin + self.growth])
        mk = side.mark()
        dt2 = ops.new_act(N, t2.H, t2.W, t2.C, dev)
        part2 = ops.conv_dgrad_bn_sums(dy2, w2, dt2, y1, st2)     # norm2's reduction pass inside conv2's data gradient
        if part2 is None:
            ops.conv_dgrad(dy2, w2, dt2)
        grads[layer.conv2.weight] = ops.conv_wgrad(t2, dy2, w2, side, after=mk)
        dy1 = ops.new_act(N, y1.H, y1.W, y1.C, dev)
        if part2 is None:
            grads[layer.norm2.weight], grads[layer.norm2.bias] = ops.bn_backward(dt2, y1, st2, layer.norm2, dy1, 2)
        else:
            grads[layer.norm2.weight], grads[layer.norm2.bias] = ops.bn_backward_from_sums(dt2, y1, st2, layer.norm2, dy1, part2)
        mk = side.mark()
        if K is not None:
            # K[:, :cin] holds what the BatchNorm handled by the previous launch into G (the layer above, or the transition)
            # still owes these channels: subtracted by this launch, then replaced by this layer's own coefficients
            pend = ops.BN_DEFERRED_PENDING
            part = ops.conv_dgrad_bn_deferred(dy1, w1, G.window(0, cin), buf.window(0, cin), st1, accumulate=True,
                                              pending=K[:, :cin] if (owed and pend) else None)
            grads[layer.conv1.weight] = ops.conv_wgrad(t1, dy1, w1, side, after=mk)
            grads[layer.norm1.weight], grads[layer.norm1.bias] = ops.bn_bwd_finalize_deferred(
                part, buf.npix, layer.norm1, st1, K[:, :cin], accumulate=owed and not pend)
        else:
            dt1 = ops.new_act(N, t1.H, t1.W, cin, dev)
            ops.conv_dgrad(dy1, w1, dt1)
            grads[layer.conv1.weight] = ops.conv_wgrad(t1, dy1, w1, side, after=mk)
            grads[layer.norm1.weight], grads[layer.norm1.bias] = ops.bn_backward(
                dt1, buf.window(0, cin), st1, layer.norm1, G.window(0, cin), 2, accumulate=True)
        self._emit(grads, list(layer.parameters()))

    def transition_backward(self, trec, g_first, grads, side, K=None):
        """Backward of one _Transition: ``g_first`` is the gradient of the pooled tensor; returns the gradient of the
        preceding block buffer.  ``K``: the preceding block's deferred-correction table (see layer_backward) -- the
        transition's BatchNorm is the first to write into it."""
        _, tr, buf, st, t, yc = trec
        dev = g_first.t.device
        N = buf.N
        wc = self.cw(tr.conv)
        dyc = ops.new_act(N, yc.H, yc.W, yc.C, dev)
        ops.avgpool_bwd(g_first, dyc)
        mk = side.mark()
        G = ops.new_act(N, buf.H, buf.W, buf.C, dev)
        if K is not None:
            part = ops.conv_dgrad_bn_deferred(dyc, wc, G, buf, st, accumulate=False)
            grads[tr.conv.weight] = ops.conv_wgrad(t, dyc, wc, side, after=mk)
            grads[tr.norm.weight], grads[tr.norm.bias] = ops.bn_bwd_finalize_deferred(part, buf.npix, tr.norm, st, K,
                                                                                      accumulate=False)
        else:
            dt = ops.new_act(N, t.H, t.W, t.C, dev)
            ops.conv_dgrad(dyc, wc, dt)
            grads[tr.conv.weight] = ops.conv_wgrad(t, dyc, wc, side, after=mk)
            grads[tr.norm.weight], grads[tr.norm.bias] = ops.bn_backward(dt, buf, st, tr.norm, G, 2)
        self._emit(grads, list(tr.parameters()))
        return G

    # ------------------------------------------------------------------ forward executor
    def _run_forward(self, images, train, save):
        dev = images.device
        N, _, H, W = images.shape
        if H % 64 or W % 64:
            raise _lib.Yv1Error("input height/width must be multiples of 64, got %dx%d" % (H, W))
        F = self.features
        self.refresh_all_weights()
        bns = []

        self._pending_seg = None
        self._cur_table = None

        def norm(stats, count, bn, C=None, apply=None):
            """BatchNorm statistics -> BNState; with ``apply`` = (x, z) also z = relu(bn(x)) -- in training mode finalize and
            apply are ONE launch (ops.BN_FUSED)."""
            if train:
                bns.append(bn)
                seg = None
                on_table = stats is self._cur_table
                if on_table and self._pending_seg is not None:  # the block's shared table: merge what is owed
                    seg, self._pending_seg = self._pending_seg, None
                if apply is not None and ops.BN_FUSED:
                    if on_table:
                        return ops.bn_finalize_merged_apply(stats, count, bn, C or bn.num_features, apply[0], apply[1], seg=seg)
                    return ops.bn_finalize_apply(stats, count, bn, apply[0], apply[1], relu=True, C=C)[0]
                st = ops.bn_finalize(stats, count, bn, C, seg=seg)
            else:
                st = ops.bn_eval_state(bn)
            if apply is not None:
                ops.bn_apply(apply[0], st, apply[1], relu=True)
            return st

        w0 = self.cw(F.conv0, stem=True)
        xp = ops.pack_input(images)
        y0 = ops.new_act(N, H // 2, W // 2, 64, dev)
        s0 = norm(ops.stem_fwd(xp, w0, y0, H, W), y0.npix, F.norm0)
        rec = {"stem": [xp, y0, s0, None, H, W, None], "stages": []}

        h, w = H // 4, W // 4
        buf = table = None
        pending_pool = ("max", y0)      # norm0 + relu0 + pool0 run as one launch: the BatchNorm output is never stored
        for item in self._plan:
            if item[0] == "block":
                _, name, nf, nl = item
                ctot = nf + nl * self.growth
                buf = ops.new_act(N, h, w, ctot, dev)
                first = buf.window(0, nf)
                if pending_pool[0] == "max":
                    rec["stem"][6] = ops.bn_act_maxpool_fwd(pending_pool[1], s0, first, relu=True, want_index=save)
                else:
                    ops.avgpool_fwd(pending_pool[1], first)
                table = None
                if train:
                    table = torch.empty((1, 2, ctot), dtype=torch.float32, device=dev)
                    self._cur_table = table
                    ops.stats_merge(ops.bn_stats(first), table[0], 0)
                lrecs = []
                for li, layer in enumerate(getattr(F, name).layers()):
                    lrec = self.layer_forward(layer, buf, table, nf + li * self.growth, norm, train)
                    if save:
                        lrecs.append(lrec)
                rec["stages"].append(("block", buf, lrecs, nf))
            else:
                _, name, cin = item
                trec = self.transition_forward(getattr(F, name), buf, table, norm)
                yc = trec[5]
                rec["stages"].append(trec)
                pending_pool = ("avg", yc)
                h, w = h // 2, w // 2

        t5 = ops.new_act(N, h, w, buf.C, dev)
        st5 = norm(table, buf.npix, F.norm5, buf.C, apply=(buf, t5))
        wh = self.cw(self.layer6)
        yh = ops.new_act(N, h, w, wh.Opad, dev)
        sh = norm(ops.conv_fwd(t5, wh, yh, train), yh.npix, self.bn_end, self.out_channels)
        pred = ops.head_fwd(yh, sh, self.out_channels)
        rec["head"] = (buf, st5, t5, yh, sh, pred)
        if train:
            self._bump_counters(bns)
        return pred, (rec if save else None)

    # ------------------------------------------------------------------ backward executor
    def _run_backward(self, rec, gpred):
        grads = {}
        F = self.features
        buf, st5, t5, yh, sh, pred = rec["head"]
        dev = pred.device
        N = yh.N
        wh = self.cw(self.layer6)
        # weight gradients run on a side stream (see OriginResNet._run_backward)
        side = ops.SideStream(dev, enabled=self._grad_ready_hook is None and self.wgrad_side_stream)
        dyh = ops.new_act(N, yh.H, yh.W, wh.Opad, dev)
        grads[self.bn_end.weight], grads[self.bn_end.bias] = ops.head_bwd(gpred, pred, yh, sh, self.bn_end, dyh)
        # every data gradient is enqueued AHEAD of the weight gradient that shares its dy (SideStream.mark): the side
        # stream's workgroups would otherwise fill the CUs first and the dgrad -- the critical path -- wait behind them
        # (trace: 40 us per dense layer, 4 ms per step)
        mk = side.mark()
        dt5 = ops.new_act(N, t5.H, t5.W, t5.C, dev)
        ops.conv_dgrad(dyh, wh, dt5)
        grads[self.layer6.weight] = ops.conv_wgrad(t5, dyh, wh, side, after=mk)
        G = ops.new_act(N, buf.H, buf.W, buf.C, dev)         # gradient of the last block's feature buffer
        grads[F.norm5.weight], grads[F.norm5.bias] = ops.bn_backward(dt5, buf, st5, F.norm5, G, 2)
        self._emit(grads, [self.bn_end.weight, self.bn_end.bias, self.layer6.weight, F.norm5.weight, F.norm5.bias])

        # deferred BatchNorm backward (ops.BN_DEFERRED): K = per-block [2][Ctot] table of the affine corrections the norm1s /
        # the transition norm owe the gradient of each feature channel.  The LAST block's buffer gradient comes from norm5's
        # ordinary backward (its convolution has 32 padded output channels: K of the GEMM too short for the ring kernels),
        # so its table starts at zero.
        deferred = ops.BN_DEFERRED
        K, owed = None, False
        for stage in reversed(rec["stages"]):
            if stage[0] == "block":
                _, buf, lrecs, nf = stage
                if deferred and K is None:
                    K = torch.empty((2, buf.C), dtype=torch.float32, device=dev)     # written before it is read (owed)
                for lrec in reversed(lrecs):
                    self.layer_backward(lrec, buf, G, grads, side, K, owed)
                    owed = owed or K is not None
                g_first = G.window(0, nf)                     # gradient w.r.t. the pooled tensor that opened the block
                if K is not None and owed:
                    ops.bn_deferred_fix(g_first, buf.window(0, nf), K[:, :nf])
                K, owed = None, False
            else:
                if deferred:
                    K = torch.empty((2, stage[2].C), dtype=torch.float32, device=dev)
                    owed = True
                G = self.transition_backward(stage, g_first, grads, side, K)

        xp, y0, s0, z0, H, W, pidx = rec["stem"]
        w0 = self.cw(F.conv0, stem=True)
        # max-pool backward gathered inside the BatchNorm backward: the pool-input gradient is never materialised
        dy0 = ops.new_act(N, y0.H, y0.W, 64, dev)
        grads[F.norm0.weight], grads[F.norm0.bias] = ops.bn_backward(g_first, y0, s0, F.norm0, dy0, 2, pool_idx=pidx)
        grads[F.conv0.weight] = ops.stem_wgrad(xp, dy0, w0, H, W, side)
        side.join()
        self._emit(grads, [F.norm0.weight, F.norm0.bias, F.conv0.weight])
        return grads


def densenet121(pretrained=False, S=7, **kwargs):
    """DenseNet-121 backbone + YOLO head (OriginDenseNet.py:149-164)."""
    if S not in [7, 14]:
        print('S musk be 7x7 or 14x14')
        exit()
    if pretrained:
        raise _lib.Yv1Error("pretrained=True would download from download.pytorch.org; load a state_dict instead")
    cfg = (6, 12, 24, 16, 16) if S == 7 else (6, 12, 24, 16)
    return DenseNet(num_init_features=64, growth_rate=32, block_config=cfg, **kwargs)
