"""What are the wrong values of a LayerNorm-folded md_igemm made of?  The q projection shape (64x64 level, cfg 15) run until a few
runs disagree; for every element that disagrees: the two values, the fp32 value, and the contribution of every 32-wide k-step of the
contraction (is the wrong value the right one minus one k-step?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from magicdance_amd import ops   # noqa: E402

dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale   # noqa: E731
b, tok, c, n, cfg = 2, 4096, 320, 320, int(sys.argv[1]) if len(sys.argv) > 1 else 15
x = rnd(b, tok, c).to(dev, F16)
w = rnd(n, c, scale=c ** -0.5).to(dev, F16)
s1, s0 = rnd(n).to(dev, F32), rnd(n).to(dev, F32)
ws = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
out = torch.zeros(b, tok, n, dtype=F16, device=dev)
runs = []
for r in range(12):
    out.zero_()
    ops.igemm(x, w, n, batch=b, hin=1, win=tok, hout=1, wout=tok, c0=c, out=out, ld_out=n, ws=ws, force_cfg=cfg, ln=(s1, s0, 1e-5))
    torch.cuda.synchronize()
    runs.append(out.clone())
st = torch.stack(runs).float()                       # [R, b, tok, n]
maj = st.median(0).values
xf, wf = x.float(), w.float()
mu = xf.mean(-1, keepdim=True)
rstd = torch.rsqrt(xf.var(-1, unbiased=False, keepdim=True) + 1e-5)
accf = xf @ wf.t()
ref = rstd * (accf - mu * s1) + s0
print(f"cfg {cfg}: majority vs fp32 reference max |diff| {float((maj - ref).abs().max()):.3e}")
shown = 0
for r in range(len(runs)):
    bad = (st[r] != maj).nonzero()
    if bad.numel() == 0:
        continue
    print(f"run {r}: {bad.shape[0]} elements differ from the majority value; rows {sorted(set(bad[:, 1].tolist()))[:20]} columns {sorted(set(bad[:, 2].tolist()))}")
    for bi, ti, ni in bad[:3].tolist():
        v_bad, v_good, v_ref = float(st[r, bi, ti, ni]), float(maj[bi, ti, ni]), float(ref[bi, ti, ni])
        steps = [float((xf[bi, ti, 32 * q:32 * q + 32] * wf[ni, 32 * q:32 * q + 32]).sum()) * float(rstd[bi, ti, 0]) for q in range(c // 32)]
        print(f"   [{bi},{ti},{ni}] wrong {v_bad:+.5f} right {v_good:+.5f} fp32 {v_ref:+.5f}  right-wrong {v_good - v_bad:+.5f}; rstd x k-step contributions "
              + " ".join(f"{s:+.4f}" for s in steps) + f"; rstd*mu*s1 {float(rstd[bi, ti, 0] * mu[bi, ti, 0] * s1[ni]):+.5f} s0 {float(s0[ni]):+.5f} rstd {float(rstd[bi, ti, 0]):.4f}")
    shown += 1
    if shown >= 4:
        break
