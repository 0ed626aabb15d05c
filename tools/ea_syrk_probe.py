#!/usr/bin/env python3
"""Feasibility probe (GPU): ExpectedAttention's query statistics from the HIDDEN states -- mu_q = W mu_x, Sigma_q(head) = W_h Sigma_x W_h^T
 -- with library GEMMs: how fast does the GEMM library run X^T X (4096 x S by S x 4096, the whole square or only the
blocks of its upper triangle) next to the q_proj it would replace, and does torch give a float32 result from bf16 operands?"""
import sys
import time

import torch

dev = "cuda:0"
S, Hd = 131068, 4096
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn((S, Hd), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
W = (torch.randn((Hd, Hd), generator=g, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("q_proj X W^T (bf16 out):", round(timeit(lambda: X @ W.T), 3), "ms")
try:
    t = timeit(lambda: torch.mm(X.T, X, out_dtype=torch.float32))
    print("X^T X full, float32 out:", round(t, 3), "ms")
    f32 = True
except Exception as e:  # noqa: BLE001
    print("torch.mm(out_dtype=float32) unsupported:", repr(e)[:200])
    f32 = False
print("X^T X full, bf16 out:", round(timeit(lambda: X.T @ X), 3), "ms")
mm = (lambda a, b: torch.mm(a, b, out_dtype=torch.float32)) if f32 else (lambda a, b: a @ b)
for nb in (2, 4, 8):
    bs = Hd // nb
    blocks = [(i, j) for i in range(nb) for j in range(i, nb)]

    def tri():
        return [mm(X[:, i * bs:(i + 1) * bs].T, X[:, j * bs:(j + 1) * bs]) for i, j in blocks]

    print(f"X^T X upper block triangle, {nb} x {nb} blocks of {bs} ({len(blocks)} GEMMs, {len(blocks) / nb / nb:.3f} of the flops):", round(timeit(tri), 3), "ms")
# chunked over tokens (16 chunks of 8192) with float32 accumulation outside: what a cancellation-safe version would do
if f32:
    acc = torch.zeros((Hd, Hd), device=dev, dtype=torch.float32)

    def chunked():
        acc.zero_()
        for c in range(0, S, 8192):
            acc.add_(torch.mm(X[c:c + 8192].T, X[c:c + 8192], out_dtype=torch.float32))

    print("X^T X full in 16 token chunks + float32 accumulate:", round(timeit(chunked), 3), "ms")
    # precision of the raw second moment with a large mean (|mean| = 10 sigma on some channels)
    Xm = (X.float() + 10.0 * (torch.arange(Hd, device=dev) % 64 == 0)).to(torch.bfloat16)
    M2 = torch.mm(Xm.T, Xm, out_dtype=torch.float32)
    mu = Xm.float().mean(0, dtype=torch.float64)
    ref = (Xm[:, :256].double().T @ Xm[:, :256].double()) / S - torch.outer(mu[:256], mu[:256])
    got = M2[:256, :256].double() / S - torch.outer(mu[:256], mu[:256])
    sig = ref.diagonal().sqrt()
    print("covariance error with |mean| = 10 sigma channels, relative to sigma_i sigma_j:", float(((got - ref).abs() / torch.outer(sig, sig)).max()))
# the projections: T_h = W_h Sigma W_h^T for 32 heads
Sig = torch.randn((Hd, Hd), device=dev, dtype=torch.float32)
Wh = W.view(32, 128, Hd)
print("W Sigma (float32 GEMM 4096 x 4096 x 4096):", round(timeit(lambda: W.float() @ Sig), 3), "ms")
hi = Sig.to(torch.bfloat16)
lo = (Sig - hi.float()).to(torch.bfloat16)
print("W Sigma as two bf16 GEMMs (hi + lo):", round(timeit(lambda: (mm(W, hi) + mm(W, lo)) if f32 else (W @ hi + W @ lo)), 3), "ms")
print("column mean of X (torch):", round(timeit(lambda: X.float().mean(0)), 3), "ms;  (X.sum via ones GEMM):", round(timeit(lambda: torch.ones((1, S), device=dev, dtype=torch.bfloat16) @ X), 3), "ms")
