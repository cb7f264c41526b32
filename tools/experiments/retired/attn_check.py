"""v2 vs v1 attention on the real layer shapes: max diff, NaN check, run-to-run determinism (GPU box only)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) < 2:
    for v in ("1", "2"):
        env = dict(os.environ, MD_ATTN_V=("1" if v == "1" else "0"))
        subprocess.check_call([sys.executable, __file__, v], env=env)
    a, b = torch.load("/tmp/attn_v1.pt"), torch.load("/tmp/attn_v2.pt")
    for k in a:
        d = (a[k].float() - b[k].float()).abs().max().item()
        print(k, "v1-v2 maxdiff", d, "finite", bool(torch.isfinite(b[k]).all()), flush=True)
    sys.exit(0)
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
res = {}
torch.manual_seed(0)
SH = [(1, 8, 4096, 4096, 4096, 1, 40), (1, 8, 1024, 1024, 1024, 1, 80), (1, 8, 256, 256, 256, 1, 160), (1, 8, 64, 64, 64, 1, 160), (1, 8, 4096, 77, 0, 0, 40), (1, 8, 1024, 77, 0, 0, 80), (1, 8, 256, 77, 0, 0, 160), (2, 8, 4096, 4096, 4096, 1, 40), (1, 8, 4096, 4096, 0, 0, 40), (2, 8, 1024, 1024, 1024, 1, 80), (1, 8, 1024, 1024, 0, 0, 80),
      (2, 8, 256, 256, 256, 1, 160), (2, 8, 64, 64, 64, 1, 160), (1, 8, 64, 64, 0, 0, 160), (2, 8, 4096, 77, 0, 0, 40), (2, 8, 64, 77, 0, 0, 160),
      (2, 8, 16, 16, 16, 1, 160), (2, 8, 4, 4, 4, 1, 160), (2, 8, 1, 1, 1, 1, 160), (1, 8, 1, 77, 0, 0, 160)]
for (b, heads, nq, n0, n1, n1b, d) in SH:
    c = heads * d
    q = torch.randn(b, nq, c, device=dev).to(F16)
    qk = torch.randn(b, n0, 2 * c, device=dev).to(F16)   # K as a strided view (column offset c), like the engine
    k0 = qk[:, :, c:]
    ld0 = (n0 + 7) // 8 * 8
    vt0 = torch.zeros(b, c, ld0, dtype=F16, device=dev); vt0[:, :, :n0] = torch.randn(b, c, n0, device=dev).to(F16)
    kw = {}
    if n1:
        ld1 = (n1 + 7) // 8 * 8
        k1 = torch.randn(1, n1, c, device=dev).to(F16); vt1 = torch.zeros(1, c, ld1, dtype=F16, device=dev); vt1[:, :, :n1] = torch.randn(1, c, n1, device=dev).to(F16)
        kw = dict(k1=k1, vt1=vt1, n1=n1, ld_k1=c, ld_vt1=ld1, k1_bs=0, vt1_bs=0, n1_batches=n1b)
    outs = []
    for rep in range(12):
        out = torch.empty(b, nq, c, dtype=F16, device=dev)
        ops.attention(q, k0, vt0, out, batch=b, heads=heads, nq=nq, d=d, n0=n0, ld_q=c, ld_k0=2 * c, ld_vt0=ld0, ld_out=c,
                      q_bs=nq * c, k0_bs=n0 * 2 * c, vt0_bs=c * ld0, out_bs=nq * c, **kw)
        torch.cuda.synchronize()
        outs.append(out)
    det = all(torch.equal(outs[0], o) for o in outs[1:])
    key = f"B{b} nq{nq} n0{n0} n1{n1} d{d}"
    print(sys.argv[1], key, "deterministic", det, "finite", bool(torch.isfinite(outs[0]).all()), flush=True)
    res[key] = outs[0].cpu()
torch.save(res, f"/tmp/attn_v{sys.argv[1]}.pt")
