"""Stress: back-to-back launches of the kernels of one transformer block, looking for run-to-run differences (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
torch.manual_seed(0)
heads, d, n = 8, 40, 4096
c = heads * d
ws = torch.zeros(128 << 20, dtype=torch.uint8, device=dev)
x = torch.randn(1, n, c, device=dev).to(F16)
wqkv = (torch.randn(3 * c, c, device=dev) * 0.05).to(F16)
wo = (torch.randn(c, c, device=dev) * 0.05).to(F16)
ctxk = torch.randn(1, 77, c, device=dev).to(F16); ctxv = torch.zeros(1, c, 80, dtype=F16, device=dev); ctxv[:, :, :77] = torch.randn(1, c, 77, device=dev).to(F16)
qk = torch.empty(1, n, 2 * c, dtype=F16, device=dev); vt = torch.empty(1, c, n, dtype=F16, device=dev)
a1 = torch.empty(1, n, c, dtype=F16, device=dev); o1 = torch.empty(1, n, c, dtype=F16, device=dev); a2 = torch.empty(1, n, c, dtype=F16, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
cfg = int(os.environ.get("CFG", "19"))
def block():
    ops.igemm(x, wqkv, 3 * c, batch=1, hin=1, win=n, hout=1, wout=n, c0=c, out=qk, ld_out=2 * c, out_t=vt, n_tr_begin=2 * c, ld_t=n, ws=ws, force_cfg=cfg, force_splitk=1)
    if mode in ("all", "self"):
        ops.attention(qk, qk[:, :, c:], vt, a1, batch=1, heads=heads, nq=n, d=d, n0=n, ld_q=2 * c, ld_k0=2 * c, ld_vt0=n, ld_out=c, q_bs=n * 2 * c, k0_bs=n * 2 * c, vt0_bs=c * n, out_bs=n * c)
    ops.igemm(a1 if mode in ("all", "self") else x, wo, c, batch=1, hin=1, win=n, hout=1, wout=n, c0=c, res=x, ld_res=c, out=o1, ws=ws, force_cfg=cfg, force_splitk=1)
    if mode in ("all", "cross"):
        ops.attention(o1, ctxk, ctxv, a2, batch=1, heads=heads, nq=n, d=d, n0=77, ld_q=c, ld_k0=c, ld_vt0=80, ld_out=c, q_bs=n * c, k0_bs=0, vt0_bs=0, out_bs=n * c)
block(); torch.cuda.synchronize()
ref = (a1.clone(), o1.clone(), a2.clone())
bad = [0, 0, 0]
N = 300
for it in range(N):
    block()
    if it % 10 == 9:
        torch.cuda.synchronize()
    cur = (a1, o1, a2)
    for i in range(3):
        if not torch.equal(ref[i], cur[i]):
            bad[i] += 1
torch.cuda.synchronize()
print(os.environ.get("TAG", ""), mode, "cfg", cfg, "mismatching iterations (self-attn, out-proj, cross-attn):", bad, "of", N, flush=True)
