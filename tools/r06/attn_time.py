import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ddnm_amd import ops
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for B, T, C in ((8, 256, 512), (8, 64, 512), (1, 256, 512)):
    qkv = torch.randn(B, T, 3 * C, device="cuda")
    S = torch.empty(B, T, T, device="cuda"); o3 = torch.empty(B, T, C, device="cuda"); f = qkv.view(-1)
    def three():
        ops.bgemm(f[0:], f[C:], S, T, T, C, lda=3 * C, ldb=3 * C, ldc=T, transb=True, batch=B, sA=(T * 3 * C, 0), sB=(T * 3 * C, 0), sC=(T * T, 0))
        ops.softmax_rows_(S, B * T, T, T, C ** -0.5)
        ops.bgemm(S, f[2 * C:], o3, T, C, T, lda=T, ldb=3 * C, ldc=C, transb=False, batch=B, sA=(T * T, 0), sB=(T * 3 * C, 0), sC=(T * C, 0))
    out = torch.empty(B, T, C, device="cuda")
    print(f"B={B} T={T} C={C}: three launches {t(three):.1f} us, fused {t(lambda: ops.attn_fused(qkv, B, T, C, 64.0, 64.0, C ** -0.5, out=out)):.1f} us")
