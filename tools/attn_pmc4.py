"""One d = 40 self + bank attention launch set per loop form (MD_ATTN_V = 0, 1) at 16 samples, for rocprofv3 --pmc; GPU box only.
NOTE: MD_ATTN_V existed only on the day of the run (gpurun r4i); both 'forms' are the shipped kernel now.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
b, heads, nq, n0, n1, n1b, d = 16, 8, 4096, 4096, 4096, 8, 40
c = heads * d
q = torch.randn(b, nq, c, device=dev).to(F16); k0 = torch.randn(b, n0, c, device=dev).to(F16)
vt0 = torch.randn(b, c, n0, device=dev).to(F16)
k1 = torch.randn(1, n1, c, device=dev).to(F16); vt1 = torch.randn(1, c, n1, device=dev).to(F16)
out = torch.empty(b, nq, c, dtype=F16, device=dev)
for v in (0, 1):
    os.environ["MD_ATTN_V"] = str(v)
    for _ in range(3):
        ops.attention(q, k0, vt0, out, batch=b, heads=heads, nq=nq, d=d, n0=n0, ld_q=c, ld_k0=c, ld_vt0=n0, ld_out=c, q_bs=nq * c,
                      k0_bs=n0 * c, vt0_bs=c * n0, out_bs=nq * c, k1=k1, vt1=vt1, n1=n1, ld_k1=c, ld_vt1=n1, k1_bs=0, vt1_bs=0, n1_batches=n1b)
    torch.cuda.synchronize()
