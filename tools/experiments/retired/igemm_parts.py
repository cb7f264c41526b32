"""Component timing of the igemm k-loop (GPU box only): run one big shape with MD_IGEMM_DEBUG masks (set per process)
and print us per launch + clocks per 64-deep k-tile per CU.  Needs a library built with the debug hooks:
  MD_EXTRA_FLAGS=-DMD_IGEMM_DEBUG bash magicdance_amd/csrc/build.sh ;  MD_IGEMM_DEBUG=<mask> python tools/igemm_parts.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0")
F16 = torch.float16
mask = int(os.environ.get("MD_IGEMM_DEBUG", "0"))
SHAPES = [(65536, 640, 5760, 3), (65536, 1280, 1280, 1), (8192, 320, 2880, 3)][:int(os.environ.get("PARTS_SHAPES", "3"))]
CFGS = [int(c) for c in os.environ.get("PARTS_CFGS", "12,25,34,35,36,37").split(",")]
for (m, n, k, ks) in SHAPES:
    side = int((m // 16) ** 0.5) if ks == 3 else 1
    b = 16
    hw = m // b
    h = int(hw ** 0.5)
    cin = k // (ks * ks)
    x = torch.randn(b, hw, cin, device=dev).to(F16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(F16)
    out = torch.empty(b, hw, n, dtype=F16, device=dev)
    for cfg in CFGS:
        def run():
            ops.igemm(x, w, n, batch=b, hin=h, win=h, hout=h, wout=h, c0=cin, ksize=ks, out=out, force_cfg=cfg, force_splitk=1)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run(); s.synchronize()
            g = ops.Graph(); g.begin()
            for _ in range(10): run()
            g.end(); g.launch(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.launch(); e1.record(s); s.synchronize()
            us = e0.elapsed_time(e1) * 100.0
            g.destroy()
        bm, bn = {12: (128, 128), 13: (128, 64), 15: (64, 64), 20: (128, 128), 25: (128, 160), 34: (128, 128), 35: (128, 160), 36: (256, 128), 37: (256, 160)}[cfg]
        tiles = ((m + bm - 1) // bm) * ((n + bn - 1) // bn) * (k // 64)
        clk = us * 1e-6 * 2.4e9 / (tiles / 256.0)
        print(f"dbg={mask} M={m} N={n} K={k} ks={ks} cfg={cfg}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF  {clk:7.0f} clk per k-tile per CU "
              f"({(bm + bn) * 128 / clk:5.1f} B/clk)", flush=True)
