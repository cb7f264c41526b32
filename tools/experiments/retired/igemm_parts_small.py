"""Component timing of SMALL step-shaped launches (GPU box only, library built with -DMD_IGEMM_DEBUG):
MD_IGEMM_DEBUG=<mask> python tools/igemm_parts_small.py   (mask bits: 1 no MFMA, 2 no LDS reads+MFMA, 4 no k-loop loads)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
mask = int(os.environ.get("MD_IGEMM_DEBUG", "0"))
ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
for (b, h, w, cin, n, ks, cfg, sp) in [(2, 16, 16, 1280, 1280, 1, 15, 1), (2, 64, 64, 320, 320, 1, 27, 1), (2, 8, 8, 1280, 1280, 3, 27, 16),
                                       (2, 32, 32, 640, 640, 1, 15, 1), (2, 64, 64, 320, 320, 3, 27, 1), (1, 8, 8, 64, 64, 1, 15, 1)]:
    m, k = b * h * w, ks * ks * cin
    x = torch.randn(b, h * w, cin, device=dev).to(F16)
    wts = [(torch.randn(n, k, device=dev) * k ** -0.5).to(F16) for _ in range(12)]   # cold weights, like a real step
    out = torch.empty(b, h * w, n, dtype=F16, device=dev)
    bias = torch.randn(n, device=dev)
    def run(i):
        ops.igemm(x, wts[i % 12], n, batch=b, hin=h, win=w, hout=h, wout=w, c0=cin, ksize=ks, bias=bias, out=out, ws=ws, force_cfg=cfg, force_splitk=sp)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(0); s.synchronize()
        g = ops.Graph(); g.begin()
        for i in range(24): run(i)
        g.end(); g.launch(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.launch(); e1.record(s); s.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 24
        g.destroy()
    print(f"dbg={mask} M={m} N={n} K={k} ks={ks} cfg={cfg} split={sp}: {us:6.1f} us per launch (incl. reduce)", flush=True)
