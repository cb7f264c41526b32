"""Run the attention shapes of the 64x64 / 32x32 levels in isolation (for rocprofv3 --pmc); GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
for (b, heads, nq, n0, n1, n1b, d) in [(2, 8, 4096, 4096, 4096, 1, 40), (1, 8, 4096, 4096, 0, 0, 40), (2, 8, 1024, 1024, 1024, 1, 80), (2, 8, 4096, 77, 0, 0, 40)]:
    c = heads * d
    q = torch.randn(b, nq, c, device=dev).to(F16); k0 = torch.randn(b, n0, c, device=dev).to(F16)
    ld0 = (n0 + 7) // 8 * 8
    vt0 = torch.zeros(b, c, ld0, dtype=F16, device=dev); vt0[:, :, :n0] = torch.randn(b, c, n0, device=dev).to(F16)
    kw = {}
    if n1:
        k1 = torch.randn(1, n1, c, device=dev).to(F16); vt1 = torch.randn(1, c, n1, device=dev).to(F16)
        kw = dict(k1=k1, vt1=vt1, n1=n1, ld_k1=c, ld_vt1=n1, k1_bs=0, vt1_bs=0, n1_batches=n1b)
    out = torch.empty(b, nq, c, dtype=F16, device=dev)
    for _ in range(3):
        ops.attention(q, k0, vt0, out, batch=b, heads=heads, nq=nq, d=d, n0=n0, ld_q=c, ld_k0=c, ld_vt0=ld0, ld_out=c,
                      q_bs=nq * c, k0_bs=n0 * c, vt0_bs=c * ld0, out_bs=nq * c, **kw)
    torch.cuda.synchronize()
