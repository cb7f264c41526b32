"""Self + bank attention launches of the 32x32 (d = 80) and 16x16 (d = 160) levels for rocprofv3 --pmc (GPU box only): the one-frame
shapes (2 and 3 samples, one of them reading the bank) and the 8-frame shape (16 samples, 8 reading), three launches each in this order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402

from magicdance_amd import ops   # noqa: E402

dev = torch.device("cuda:0")
F16 = torch.float16
heads = 8
for d, nq in ((80, 1024), (160, 256)):
    for b, n1b in ((2, 1), (3, 1), (16, 8)):
        c = heads * d
        q = torch.randn(b, nq, c, device=dev).to(F16)
        k0 = torch.randn(b, nq, c, device=dev).to(F16)
        vt0 = torch.randn(b, c, nq, device=dev).to(F16)
        k1 = torch.randn(1, nq, c, device=dev).to(F16)
        vt1 = torch.randn(1, c, nq, device=dev).to(F16)
        out = torch.empty(b, nq, c, dtype=F16, device=dev)
        for _ in range(3):
            ops.attention(q, k0, vt0, out, batch=b, heads=heads, nq=nq, d=d, n0=nq, ld_q=c, ld_k0=c, ld_vt0=nq, ld_out=c, q_bs=nq * c,
                          k0_bs=nq * c, vt0_bs=c * nq, out_bs=nq * c, k1=k1, vt1=vt1, n1=nq, ld_k1=c, ld_vt1=nq, k1_bs=0, vt1_bs=0,
                          n1_batches=n1b, q_prescaled=True)
        torch.cuda.synchronize()
