"""dW = dY^T X for the Update operator's Linear layers in training (K = 18 000 edge rows, 384 x 384 output): hipBLASLt's plain
GEMM against a split-K formulation as a batched GEMM over row chunks.  python tools/ubench_dw_gemm.py"""
import torch, time
dev = torch.device("cuda", 0)
K, I, O = 18000, 384, 384
X = torch.randn(K, I, device=dev); dY = torch.randn(K, O, device=dev)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
ref = dY.t() @ X
print(f"dY^T @ X plain: {timed(lambda: dY.t() @ X):.1f} us")
for S in (4, 8, 16, 32, 64, 125):
    if K % S: continue
    def f():
        return torch.bmm(dY.view(S, K // S, O).transpose(1, 2), X.view(S, K // S, I)).sum(0)
    err = (f() - ref).abs().max().item() / ref.abs().max().item()
    print(f"split-K as bmm, S={S:3d}: {timed(f):.1f} us (rel diff {err:.1e})")
W = torch.randn(O, I, device=dev)
print(f"forward X @ W^T: {timed(lambda: X @ W.t()):.1f} us; dX = dY @ W: {timed(lambda: dY @ W):.1f} us; bias grad dY.sum(0): {timed(lambda: dY.sum(0)):.1f} us")
