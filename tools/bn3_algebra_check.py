"""Checks the identities of DESIGN.md section 7 ("bn3's backward moved into the 1x1 GEMMs next to it") on the CPU: the
BatchNorm-3 backward of a Bottleneck expressed through T = g^T z2, G = z2^T z2 and column sums, against the ordinary
reduce / apply / dgrad / wgrad sequence -- exactly (fp64) and with the bf16 roundings either path would have on the GPU.
python tools/bn3_algebra_check.py [M p]"""
import sys
import torch
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = int(sys.argv[2]) if len(sys.argv) > 2 else 64
C = 4 * p
bf = lambda t: t.to(torch.bfloat16).to(t.dtype)

def run(dt, emulate):
    q = bf if emulate else (lambda t: t)
    z2 = q(torch.relu(torch.randn(M, p, dtype=dt) * 0.8 + 0.2))              # conv3's input (post-ReLU bn2 output)
    W3 = q(torch.randn(C, p, dtype=dt) * (2.0 / p) ** 0.5)                   # [4p][p]
    y3 = q(z2 @ W3.t())                                                       # conv3 output as stored
    mu, var = y3.mean(0), y3.var(0, unbiased=False)
    inv = (var + 1e-5).rsqrt()
    gamma = torch.rand(C, dtype=dt) + 0.5
    g = q(torch.randn(M, C, dtype=dt) * 1e-3)
    mask = torch.rand(M, C) > 0.45
    gm = torch.where(mask, g, torch.zeros_like(g))                            # masked gradient (bf16 values)
    xhat = (y3 - mu) * inv
    # ---- the ordinary sequence (what the kernels do today)
    sg, sgx = gm.sum(0), (gm * xhat).sum(0)
    k1 = gamma * inv; k2 = k1 * sg / M; k3 = k1 * sgx / M
    dy3 = q(k1 * gm - k2 - xhat * k3)                                         # rounded to bf16 by the apply pass
    dz2_ref = q(dy3 @ W3)
    dW3_ref = dy3.t() @ z2
    # ---- the algebraic form
    T = gm.t() @ z2                                                           # [4p][p]: the weight-gradient GEMM fed g
    G = z2.t() @ z2                                                           # [p][p]
    sz = z2.sum(0)
    sgy = (W3 * T).sum(1)                                                     # sum_m g*y3 per channel: row dots, no pass over y3
    sgx2 = inv * (sgy - mu * sg)
    k3b = k1 * sgx2 / M
    d = k3b * inv
    dW3 = k1[:, None] * T - d[:, None] * (W3 @ G - mu[:, None] * sz[None, :]) - k2[:, None] * sz[None, :]
    Wp = q(k1[:, None] * W3)                                                  # diag(k1) W3, stored bf16
    A = q(W3.t() @ (d[:, None] * W3))                                         # [p][p], stored bf16
    crow = ((-k2 + d * mu)[:, None] * W3).sum(0)
    dz2 = q(gm @ Wp - z2 @ A + crow)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    return dict(sgx=rel(sgx2, sgx), dW3=rel(dW3, dW3_ref), dz2=rel(dz2, dz2_ref),
                dz2_vs_exact=None)

for name, dt, em in (("fp64, no rounding (identity check)", torch.float64, False),
                     ("fp32 accumulation, bf16 storage on both paths", torch.float32, True)):
    r = run(dt, em)
    print("%-48s  sum g*xhat %.2e   dW3 %.2e   dz2 %.2e   (relative L2 against the ordinary sequence)" % (name, r["sgx"], r["dW3"], r["dz2"]))
