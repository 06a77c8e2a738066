"""Drop-in for the reference's ``backbones/OriginResNet.py`` on MI355X.

``resnet50(pretrained=False, S=7)`` returns an ``nn.Module`` with the reference's state_dict keys
and OIHW weight shapes (conv1, bn1, layer{1..5}.{i}.conv{1,2,3} / bn{1,2,3} / downsample.{0,1},
layer6, bn_end -- OriginResNet.py:112-134,:155-171) whose forward maps fp32 NCHW images to
``[N, S, S, B*5+C]`` sigmoid outputs (:173-195).  Nothing of it runs through ATen: the module
tree only holds parameters; forward and backward are explicit HIP launch sequences over NHWC
bf16 activations (csrc/conv.hip, wgrad.hip, elementwise.hip), recorded as one autograd node.

Per Bottleneck (OriginResNet.py:87-107), forward:
    y1 = conv1x1(x)            + BN statistics in the conv epilogue
    z1 = relu(bn1(y1))
    y2 = conv3x3(z1, stride)   (stride on the 3x3, :79)
    z2 = relu(bn2(y2))
    y3 = conv1x1(z2)
    yd = conv1x1(x, stride)    projection shortcut when the shape changes (:159-163)
    out = relu(bn3(y3) + (bn_d(yd) | x))      one fused elementwise kernel
"""
import torch
import torch.nn as nn

from .. import _lib, ops
from ..engine import ConvParam, HipBackbone, make_bn

__all__ = ['ResNet', 'resnet50']

EXPANSION = 4


class Bottleneck(nn.Module):
    """Parameter container for one bottleneck (keys conv1,bn1,conv2,bn2,conv3,bn3[,downsample.0/.1])."""

    def __init__(self, inplanes, planes, stride=1, project=False):
        super().__init__()
        self.conv1 = ConvParam(inplanes, planes, 1)
        self.bn1 = make_bn(planes)
        self.conv2 = ConvParam(planes, planes, 3, stride, 1)
        self.bn2 = make_bn(planes)
        self.conv3 = ConvParam(planes, planes * EXPANSION, 1)
        self.bn3 = make_bn(planes * EXPANSION)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(ConvParam(inplanes, planes * EXPANSION, 1, stride), make_bn(planes * EXPANSION))
        self.stride = stride


class ResNet(HipBackbone):
    # (state_dict name, planes, stride); block counts come from ``layers``
    STAGES = (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2))

    def __init__(self, layers=(3, 4, 6, 3), S=7, B=2, num_classes=20):
        super().__init__()
        self.S7 = S == 7
        self.out_channels = B * 5 + num_classes
        self.conv1 = ConvParam(3, 64, 7, 2, 3)
        self.bn1 = make_bn(64)
        inpl = 64
        stages = [(n, p, s, layers[i]) for i, (n, p, s) in enumerate(self.STAGES)]
        if self.S7:
            stages.append(("layer5", 512, 2, layers[3]))     # extra stride-2 stage for the 7x7 grid (:131-132)
        self._stage_names = [s[0] for s in stages]
        for name, planes, stride, nblocks in stages:
            blocks = []
            for i in range(nblocks):
                st = stride if i == 0 else 1
                blocks.append(Bottleneck(inpl, planes, st, project=(i == 0 and (st != 1 or inpl != planes * EXPANSION))))
                inpl = planes * EXPANSION
            setattr(self, name, nn.Sequential(*blocks))
        self.layer6 = ConvParam(inpl, self.out_channels, 1)      # head, :133
        self.bn_end = make_bn(self.out_channels)
        for m in self.modules():                                  # init as :138-143
            if isinstance(m, ConvParam):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    # ------------------------------------------------------------------ forward executor
    def _blocks(self):
        for name in self._stage_names:
            for blk in getattr(self, name):
                yield blk

    def _run_forward_eval(self, images):
        """eval() mode: BatchNorm uses its running statistics, so it folds into per-channel scale/shift and rides in
        the convolution epilogue together with the residual add and the ReLU -- one launch per convolution, no
        stand-alone BatchNorm pass, no raw conv outputs kept (there is no backward in this mode)."""
        dev = images.device
        N, _, H, W = images.shape
        w0 = self.cw(self.conv1, stem=True)
        xp = ops.pack_input(images)
        z0 = ops.new_act(N, H // 2, W // 2, 64, dev)
        ops.stem_fwd_bn_act(xp, w0, z0, H, W, ops.bn_eval_state(self.bn1), relu=True)
        x = ops.new_act(N, H // 4, W // 4, 64, dev)
        ops.maxpool_fwd(z0, x)
        for blk in self._blocks():
            w1, w2, w3 = self.cw(blk.conv1), self.cw(blk.conv2), self.cw(blk.conv3)
            planes = blk.conv1.out_channels
            z1 = ops.new_act(N, x.H, x.W, planes, dev)
            ops.conv_fwd_bn_act(x, w1, z1, ops.bn_eval_state(blk.bn1))
            h2, w2_ = ops.conv_out_hw(x.H, x.W, 3, blk.stride, 1)
            z2 = ops.new_act(N, h2, w2_, planes, dev)
            ops.conv_fwd_bn_act(z1, w2, z2, ops.bn_eval_state(blk.bn2))
            if blk.downsample is not None:
                res = ops.new_act(N, h2, w2_, planes * EXPANSION, dev)
                ops.conv_fwd_bn_act(x, self.cw(blk.downsample[0]), res, ops.bn_eval_state(blk.downsample[1]), relu=False)
            else:
                res = x
            out = ops.new_act(N, h2, w2_, planes * EXPANSION, dev)
            ops.conv_fwd_bn_act(z2, w3, out, ops.bn_eval_state(blk.bn3), relu=True, residual=res)
            x = out
        wh = self.cw(self.layer6)
        yh = ops.new_act(N, x.H, x.W, wh.Opad, dev)
        ops.conv_fwd(x, wh, yh, False)
        return ops.head_fwd(yh, ops.bn_eval_state(self.bn_end), self.out_channels)

    def block_forward(self, blk, x, x8, norm, conv, q8, save, norm_apply=None):
        """One Bottleneck (OriginResNet.py:87-107).  ``norm(stats, count, bn)`` -> BNState, ``conv(x, x8, ConvParam, y)``
        -> statistic partials, ``q8(act)`` -> e4m3 twin or None; ``norm_apply`` (training, ops.BN_FUSED): BatchNorm finalize
        and apply in ONE launch -- same signature family as ops.bn_finalize_apply.  Returns (out, out8, record for
        block_backward)."""
        dev = x.t.device
        N = x.N
        planes, cout = blk.conv1.out_channels, blk.conv3.out_channels
        y1 = ops.new_act(N, x.H, x.W, planes, dev)
        z1 = ops.new_act(N, x.H, x.W, planes, dev)
        z1_8 = q8(z1)
        p1 = conv(x, x8, blk.conv1, y1)
        if norm_apply is not None:
            s1, _, _ = norm_apply(p1, blk.bn1, y1, z1, z8=z1_8)
        else:
            s1 = norm(p1, y1.npix, blk.bn1)
            ops.bn_apply(y1, s1, z1, relu=True, z8=z1_8)
        h2, w2_ = ops.conv_out_hw(x.H, x.W, 3, blk.stride, 1)
        y2 = ops.new_act(N, h2, w2_, planes, dev)
        z2 = ops.new_act(N, h2, w2_, planes, dev)
        z2_8 = q8(z2)
        p2 = conv(z1, z1_8, blk.conv2, y2)
        if norm_apply is not None:
            s2, _, _ = norm_apply(p2, blk.bn2, y2, z2, z8=z2_8)
        else:
            s2 = norm(p2, y2.npix, blk.bn2)
            ops.bn_apply(y2, s2, z2, relu=True, z8=z2_8)
        y3 = ops.new_act(N, h2, w2_, cout, dev)
        p3 = conv(z2, z2_8, blk.conv3, y3)
        out = ops.new_act(N, h2, w2_, cout, dev)
        out8 = q8(out)
        yd = sd = None
        if blk.downsample is not None:
            yd = ops.new_act(N, h2, w2_, cout, dev)
            pd = conv(x, x8, blk.downsample[0], yd)
            if norm_apply is not None:            # bn3, the downsample BatchNorm and the block's closing add + ReLU: one launch
                s3, omask, sd = norm_apply(p3, blk.bn3, y3, out, residual=yd, res_stats=pd, res_bn=blk.downsample[1],
                                           want_mask=save, z8=out8)
            else:
                s3 = norm(p3, y3.npix, blk.bn3)
                sd = norm(pd, yd.npix, blk.downsample[1])
                omask = ops.bn_apply(y3, s3, out, relu=True, residual=yd, res_state=sd, want_mask=save, z8=out8)
        elif norm_apply is not None:
            s3, omask, _ = norm_apply(p3, blk.bn3, y3, out, residual=x, want_mask=save, z8=out8)
        else:
            s3 = norm(p3, y3.npix, blk.bn3)
            omask = ops.bn_apply(y3, s3, out, relu=True, residual=x, want_mask=save, z8=out8)
        return out, out8, (blk, x, y1, s1, z1, y2, s2, z2, y3, s3, yd, sd, out, omask)

    def _run_forward(self, images, train, save):
        dev = images.device
        N, _, H, W = images.shape
        if H % 64 or W % 64:
            raise _lib.Yv1Error("input height/width must be multiples of 64, got %dx%d" % (H, W))
        self.refresh_all_weights()
        if not train and self.fused_eval:
            return self._run_forward_eval(images), None
        bns = []

        def norm(stats, count, bn, C=None):
            if train:
                bns.append(bn)
                return ops.bn_finalize(stats, count, bn, C)
            return ops.bn_eval_state(bn)

        def norm_apply(stats, bn, y, z, residual=None, res_stats=None, res_bn=None, want_mask=False, z8=None):
            bns.append(bn)
            if res_bn is not None:
                bns.append(res_bn)
            return ops.bn_finalize_apply(stats, y.npix, bn, y, z, relu=True, residual=residual, res_stats=res_stats,
                                         res_bn=res_bn, want_mask=want_mask, z8=z8)

        fused = norm_apply if (train and ops.BN_FUSED) else None
        rec = {"blocks": []}
        # fp8 forward GEMMs (training, BASELINE config 5): every Bottleneck convolution reads e4m3 copies of its input
        # and weights; the copies are written by the BN-apply that produces the activation.  Everything the backward
        # reads (raw conv outputs, bf16 activations, bf16 weights) is unchanged.
        f8 = bool(train and self.fp8_forward)
        if f8:
            self.refresh_all_weights_fp8()

        def conv(xa, xa8, cp, ya):
            if f8:
                return ops.conv_fwd_fp8(xa8, self.cw8(cp), ya, train)
            return ops.conv_fwd(xa, self.cw(cp), ya, train)

        def q8(act):
            return ops.Fp8Act(act.N, act.H, act.W, act.C, dev) if f8 else None

        # stem: 7x7/2 conv -> BN -> ReLU -> maxpool 3x3/2                       (:174-177)
        w0 = self.cw(self.conv1, stem=True)
        xp = ops.pack_input(images)
        y0 = ops.new_act(N, H // 2, W // 2, 64, dev)
        s0 = norm(ops.stem_fwd(xp, w0, y0, H, W), y0.npix, self.bn1)
        x = ops.new_act(N, H // 4, W // 4, 64, dev)
        pidx = ops.bn_act_maxpool_fwd(y0, s0, x, relu=True, want_index=save)   # BN + ReLU + pool: one launch, z0 never stored
        x8 = ops.quantize_fp8(x) if f8 else None
        rec["stem"] = (xp, y0, s0, None, H, W, pidx)

        for blk in self._blocks():
            x, x8, brec = self.block_forward(blk, x, x8, norm, conv, q8, save, fused)
            if save:
                rec["blocks"].append(brec)

        # head: 1x1 conv -> bn_end -> sigmoid, already NHWC                      (:186-189)
        wh = self.cw(self.layer6)
        yh = ops.new_act(N, x.H, x.W, wh.Opad, dev)
        sh = norm(ops.conv_fwd(x, wh, yh, train), yh.npix, self.bn_end, self.out_channels)
        pred = ops.head_fwd(yh, sh, self.out_channels)
        rec["head"] = (x, yh, sh, pred)
        if train:
            self._bump_counters(bns)
        return pred, (rec if save else None)

    def _bn3_algebra_ok(self, brec):
        """Identity-shortcut Bottleneck whose BatchNorm-3 backward can run as algebra (ops.bn3_algebra_backward)."""
        blk, x, y1 = brec[0], brec[1], brec[2]
        p = blk.conv1.out_channels
        return (brec[13] is not None and ops.BN3_ALGEBRA_MAX_P > 0 and p % 64 == 0 and p <= ops.BN3_ALGEBRA_MAX_P and
                (brec[10] is None or (ops.BN3_ALGEBRA_PROJ and x.C % 64 == 0)))

    def block_backward(self, brec, g, grads, side, g_sum=None, below=None):
        """Backward of one Bottleneck: ``g`` is the gradient of the block output; fills ``grads`` and returns the gradient of
        the block input.  ``g_sum``: ``g`` arrives already ReLU-masked with its per-tile column sums (the block above stored
        it that way): BatchNorm-3 + conv3 run as algebra.  ``below``: the record of the block whose output gradient this
        block produces, when THAT block wants it masked + summed; returns (g_in, sums or None)."""
        (blk, x, y1, s1, z1, y2, s2, z2, y3, s3, yd, sd, out, omask) = brec
        dev = g.t.device
        N = x.N
        w1, w2, w3 = self.cw(blk.conv1), self.cw(blk.conv2), self.cw(blk.conv3)
        g_in = ops.new_act(N, x.H, x.W, x.C, dev)
        algebra = g_sum is not None
        dz2 = ops.new_act(N, z2.H, z2.W, z2.C, dev)
        part2 = None
        if algebra:
            # bn3's reduce / finalize / apply passes and conv3's ordinary dgrad + wgrad, as four GEMM-side steps
            (grads[blk.bn3.weight], grads[blk.bn3.bias], grads[blk.conv3.weight]) = ops.bn3_algebra_backward(
                g, g_sum, z2, w3, s3, blk.bn3, blk.conv3.weight, dz2, side)
            if yd is not None:
                wd = self.cw(blk.downsample[0])
                bnd = blk.downsample[1]
        else:
            dy3 = ops.new_act(N, y3.H, y3.W, y3.C, dev)
            if yd is not None:
                wd = self.cw(blk.downsample[0])
                bnd = blk.downsample[1]
                dyd = ops.new_act(N, yd.H, yd.W, yd.C, dev)
                # bn3 and the downsample BatchNorm receive the same masked gradient: one reduction + one apply pass for both
                if self.bn_dual:
                    (grads[blk.bn3.weight], grads[blk.bn3.bias]), (grads[bnd.weight], grads[bnd.bias]) = ops.bn_backward_dual(
                        g, omask, (y3, s3, blk.bn3, dy3), (yd, sd, bnd, dyd))
                else:
                    grads[blk.bn3.weight], grads[blk.bn3.bias] = ops.bn_backward(g, y3, s3, blk.bn3, dy3, 3, z=omask)
                    grads[bnd.weight], grads[bnd.bias] = ops.bn_backward(g, yd, sd, bnd, dyd, 3, z=omask)
            else:
                # identity shortcut: its contribution to g_in (g where the block output was positive) is added in
                # the epilogue of conv1's dgrad below
                grads[blk.bn3.weight], grads[blk.bn3.bias] = ops.bn_backward(g, y3, s3, blk.bn3, dy3, 3, z=omask)
            # the data gradient (critical path) is enqueued BEFORE the weight gradient that reads the same dy: launched
            # the other way round, the side stream's wgrad workgroups fill the CUs first and the dgrad waits behind
            # them (50 us per layer in the trace); this way the wgrad runs beside the bandwidth-bound BN kernels.
            # Inside a captured hipGraph the order decides more: the HIP runtime hands a node's FIRST captured successor the
            # node's own queue and every further successor another one, so a weight gradient captured before the next
            # main-chain kernel pushes the main chain onto a new queue -- after four such forks (the projection blocks) it
            # wrapped around onto the weight gradients' queue and layer2's backward ran serialized with them
            mk = side.mark()
            # bn2's reduction pass rides in conv3's data gradient where that shape has the epilogue (ops.conv_dgrad_bn_sums)
            part2 = ops.conv_dgrad_bn_sums(dy3, w3, dz2, y2, s2) if self.bn_sums_conv3 else None
            if part2 is None:
                ops.conv_dgrad(dy3, w3, dz2)
            if yd is not None:
                grads[blk.downsample[0].weight] = ops.conv_wgrad(x, dyd, wd, side, after=mk)
            grads[blk.conv3.weight] = ops.conv_wgrad(z2, dy3, w3, side, after=mk)
        dy2 = ops.new_act(N, y2.H, y2.W, y2.C, dev)
        if part2 is None:
            grads[blk.bn2.weight], grads[blk.bn2.bias] = ops.bn_backward(dz2, y2, s2, blk.bn2, dy2, 2)
        else:
            grads[blk.bn2.weight], grads[blk.bn2.bias] = ops.bn_backward_from_sums(dz2, y2, s2, blk.bn2, dy2, part2)
        mk = side.mark()
        dz1 = ops.new_act(N, z1.H, z1.W, z1.C, dev)
        part1 = ops.conv_dgrad_bn_sums(dy2, w2, dz1, y1, s1) if self.bn_sums_conv2 else None   # (engine.py: measured +0.5 %)
        if part1 is None:
            ops.conv_dgrad(dy2, w2, dz1)
        grads[blk.conv2.weight] = ops.conv_wgrad(z1, dy2, w2, side, after=mk)
        dy1 = ops.new_act(N, y1.H, y1.W, y1.C, dev)
        if part1 is None:
            grads[blk.bn1.weight], grads[blk.bn1.bias] = ops.bn_backward(dz1, y1, s1, blk.bn1, dy1, 2)
        else:
            grads[blk.bn1.weight], grads[blk.bn1.bias] = ops.bn_backward_from_sums(dz1, y1, s1, blk.bn1, dy1, part1)
        mk = side.mark()
        sums = None
        if yd is not None and algebra:
            # projection block under the algebra: conv1's data gradient first, then the downsample BatchNorm + convolution
            # backward as algebra on the same masked gradient, scatter-accumulated into g_in (x at the strided pixels, dense)
            bmask = below[13] if below is not None else None
            s_a = ops.conv_dgrad_out(dy1, w1, g_in, False, bmask) if below is not None else ops.conv_dgrad(dy1, w1, g_in)
            xs = x if wd.stride == 1 else ops.subsample2(x)
            res = ops.bn3_algebra_backward(g, g_sum, xs, wd, sd, bnd, blk.downsample[0].weight, g_in, side, stride=wd.stride,
                                           accumulate=True, out_mask=bmask)
            grads[bnd.weight], grads[bnd.bias], grads[blk.downsample[0].weight] = res[0], res[1], res[2]
            if below is not None:
                sums = torch.cat([s_a, res[3]], 0)
        elif yd is not None and below is not None:
            # the block below runs its bn3 backward as algebra: both data gradients store its output gradient masked by ITS
            # ReLU mask and report the column sums of what they added
            ra, rb = ops.dgrad_gsum_rows(dy1, w1), ops.dgrad_gsum_rows(dyd, wd)
            sums = torch.empty((ra + rb, g_in.C), dtype=torch.float32, device=dev)
            ops.conv_dgrad_out(dy1, w1, g_in, False, below[13], gsum=sums[:ra])
            ops.conv_dgrad_out(dyd, wd, g_in, True, below[13], gsum=sums[ra:])
        elif yd is not None:
            ops.conv_dgrad(dy1, w1, g_in, accumulate=False)
            ops.conv_dgrad(dyd, wd, g_in, accumulate=True)      # strided 1x1: scatter-accumulate
        elif below is not None or algebra:
            # g is added unmasked-or-premasked; the block BELOW gets its output gradient masked + summed when it asks for it
            sums = ops.conv_dgrad_add_masked_out(dy1, w1, g_in, g, None if algebra else omask,
                                                 out_mask=below[13] if below is not None else None,
                                                 want_sum=below is not None)
        else:
            ops.conv_dgrad_add_masked(dy1, w1, g_in, g, omask)
        grads[blk.conv1.weight] = ops.conv_wgrad(x, dy1, w1, side, after=mk)
        self._emit(grads, list(blk.parameters()))
        return (g_in, sums) if (below is not None or algebra or g_sum is not None) else g_in

    # ------------------------------------------------------------------ backward executor
    def _run_backward(self, rec, gpred):
        grads = {}
        x, yh, sh, pred = rec["head"]
        dev = pred.device
        N = yh.N
        # weight gradients overlap with the dgrad / BN chain on a side stream, unless a gradient-ready hook
        # (overlapped RCCL all-reduce issued from the main stream) needs them in main-stream order
        side = ops.SideStream(dev, enabled=self._grad_ready_hook is None and self.wgrad_side_stream)
        # phase boundaries (see HipBackbone.set_phase_boundary): after layer4 -- head + layer5 + layer4 = 79 % of the
        # gradient bytes after ~10 % of the backward time -- and after layer3 (another 17 %, ~45 % of the time); what
        # is left for the last, un-overlapped collective is layer2 + layer1 + stem = 1.5 M parameters (6 MB)
        boundary_blks = [self.layer4[0], self.layer3[0]][:self.phase_boundaries]
        wh = self.cw(self.layer6)
        dyh = ops.new_act(N, yh.H, yh.W, wh.Opad, dev)
        dg, db = ops.head_bwd(gpred, pred, yh, sh, self.bn_end, dyh)
        grads[self.bn_end.weight], grads[self.bn_end.bias] = dg, db
        mk = side.mark()
        g = ops.new_act(N, x.H, x.W, x.C, dev)
        ops.conv_dgrad(dyh, wh, g)                  # main chain first: it keeps the capture's queue (see block_backward)
        grads[self.layer6.weight] = ops.conv_wgrad(x, dyh, wh, side, after=mk)
        self._emit(grads, [self.bn_end.weight, self.bn_end.bias, self.layer6.weight])

        # The side stream gets little of the chip while the main stream's kernels run and finishes its queue alone after
        # them (trace: the main stream idles ~4 ms at the end of the backward).  The weight gradients of the LAST blocks of
        # the backward (layer1, large maps: HBM-bound, ~120 us each) therefore go onto the main stream, which would
        # otherwise wait for them anyway: both queues then end together.
        inline = ops.SideStream(dev, enabled=False)
        nblk = len(rec["blocks"])
        blocks = rec["blocks"]
        g_sum = None
        for bi, brec in enumerate(reversed(blocks)):
            side.wide = nblk - bi <= self.wgrad_wide_tail
            fi = nblk - 1 - bi                                        # forward index of this block
            # the block below (processed next) runs its bn3 backward as algebra when it is an eligible identity block AND
            # this block's conv1 data gradient is the one that produces its output gradient (identity shortcut here)
            below = blocks[fi - 1] if (fi > 0 and self._bn3_algebra_ok(blocks[fi - 1])) else None
            res = self.block_backward(brec, g, grads, side if nblk - bi > self.wgrad_main_tail else inline,
                                      g_sum=g_sum, below=below)
            g, g_sum = res if isinstance(res, tuple) else (res, None)
            if self._phase_boundary is not None and any(brec[0] is b for b in boundary_blks):
                side.join()
                self._phase_boundary(grads)

        xp, y0, s0, z0, H, W, pidx = rec["stem"]
        w0 = self.cw(self.conv1, stem=True)
        # max-pool backward gathered inside the BatchNorm backward: the pool-input gradient is never materialised
        dy0 = ops.new_act(N, y0.H, y0.W, 64, dev)
        grads[self.bn1.weight], grads[self.bn1.bias] = ops.bn_backward(g, y0, s0, self.bn1, dy0, 2, pool_idx=pidx)
        grads[self.conv1.weight] = ops.stem_wgrad(xp, dy0, w0, H, W, side if self.wgrad_main_tail == 0 else None)
        side.join()
        self._emit(grads, [self.bn1.weight, self.bn1.bias, self.conv1.weight])
        return grads


def resnet50(pretrained=False, S=7, **kwargs):
    """ResNet-50 backbone + YOLO head (OriginResNet.py:220-231).  ``pretrained`` needs a network
    fetch in the reference (:229-230) and is not available offline: load weights by name with
    ``load_state_dict`` instead (train.py:61-67)."""
    if S not in [7, 14]:
        print('S musk be 7x7 or 14x14')      # same idiom as the reference (:225-227)
        exit()
    if pretrained:
        raise _lib.Yv1Error("pretrained=True would download from download.pytorch.org; load a state_dict instead")
    return ResNet((3, 4, 6, 3), S=S, **kwargs)
