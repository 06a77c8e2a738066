"""Micro-bench of the BatchNorm kernels at the ResNet-50 (N=64, 448^2) tensor shapes: GB/s achieved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd import ops
from yolo_v1_amd._lib import lib, ptr, stream_ptr, check
DEV = "cuda:0"
N = 64
SHAPES = [(256, 112), (64, 112), (512, 56), (128, 56), (1024, 28), (256, 28), (2048, 14), (512, 14), (2048, 7)]

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

print("%-14s %10s %10s %10s %10s | GB/s apply reduce3 bwdapply3 reduce2" % ("C@H", "apply", "reduce m3", "bwdapp m3", "reduce m2"))
for C, H in SHAPES:
    y = ops.Act(torch.randn(N, H, H, C, device=DEV).to(torch.bfloat16))
    res = ops.Act(torch.randn(N, H, H, C, device=DEV).to(torch.bfloat16))
    dz = ops.Act(torch.randn(N, H, H, C, device=DEV).to(torch.bfloat16))
    z = ops.new_act(N, H, H, C, DEV)
    dy = ops.new_act(N, H, H, C, DEV)
    dres = ops.new_act(N, H, H, C, DEV)
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    st = ops.bn_finalize(ops.bn_stats(y), y.npix, bn)
    mask = ops.bn_apply(y, st, z, relu=True, residual=res, want_mask=True)
    L = lib(); s = stream_ptr(DEV)
    rows = L.yv1_bn_reduce_rows(y.npix, C)
    part = torch.empty(rows * 2 * C, dtype=torch.float32, device=DEV)
    gb = torch.empty((5, C), dtype=torch.float32, device=DEV)
    t_apply = timeit(lambda: ops.bn_apply(y, st, z, relu=True, residual=res))
    def red(mode, zz, zld):
        check(L.yv1_bn_bwd_reduce(dz.p, dz.ld, zz, zld, y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
                                  y.npix, C, mode, ptr(part), s), "r")
    t_r3 = timeit(lambda: red(3, mask.p, mask.ld))
    t_r2 = timeit(lambda: red(2, None, 0))
    check(L.yv1_bn_bwd_finalize(ptr(part), rows, C, float(y.npix), ptr(bn.weight), ptr(st.invstd), ptr(gb[0]), ptr(gb[1]), ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), s), "f")
    def app():
        check(L.yv1_bn_bwd_apply(dz.p, dz.ld, mask.p, mask.ld, y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
                                 ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), y.npix, C, 3, dy.p, dy.ld, dres.p, dres.ld, 0, s), "a")
    t_a3 = timeit(app)
    def app2():
        check(L.yv1_bn_bwd_apply(dz.p, dz.ld, mask.p, mask.ld, y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
                                 ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), y.npix, C, 3, dy.p, dy.ld, None, 0, 0, s), "a")
    t_a3 = timeit(app2)
    T = y.npix * C * 2 / 1e3     # KB per tensor
    print("%4d@%3d rows%5d %9.1f %10.1f %10.1f %10.1f | %5.0f %5.0f %5.0f %5.0f" % (
        C, H, rows, t_apply, t_r3, t_a3, t_r2, 3 * T / t_apply, 2.06 * T / t_r3, 4.06 * T / t_a3, 2 * T / t_r2))
