"""Row-major vs tiled weight storage (md_igemm_params.w_tiled) on cold weights: every replayed launch reads a different copy of W
(>= 320 MB of copies in rotation, beyond the 256 MB Infinity Cache), the launcher's own tile choice.  GPU box only.
usage: python tools/wtile_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops

dev = torch.device("cuda:0")
F16 = torch.float16
SHAPES = [  # (B, h, w, cin, cout, ksize)
    (3, 8, 8, 1280, 1280, 3), (2, 8, 8, 2560, 1280, 3), (2, 8, 8, 1280, 1280, 3), (3, 16, 16, 1280, 1280, 3),
    (2, 16, 16, 1280, 1280, 3), (2, 16, 16, 2560, 1280, 3), (2, 16, 16, 1920, 1280, 3), (2, 16, 16, 1280, 1280, 1),
    (2, 16, 16, 5120, 1280, 1), (2, 16, 16, 1280, 10240, 1), (2, 16, 16, 1280, 3840, 1), (3, 32, 32, 640, 640, 3),
    (2, 32, 32, 640, 640, 3), (2, 32, 32, 1920, 640, 3), (2, 32, 32, 640, 640, 1), (2, 32, 32, 2560, 640, 1), (2, 32, 32, 640, 5120, 1),
    (3, 64, 64, 320, 320, 3), (2, 64, 64, 320, 320, 3), (2, 64, 64, 960, 320, 3), (2, 64, 64, 320, 320, 1), (2, 64, 64, 320, 2560, 1),
    (16, 64, 64, 320, 320, 3), (16, 16, 16, 1280, 1280, 3),
]
ws = torch.zeros(512 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
REPS = 12
for (b, h, w, cin, cout, k) in SHAPES:
    M, K = b * h * w, k * k * cin
    x = torch.randn(b, h * w, cin, device=dev).to(F16)
    ncopy = max(2, min(REPS, int((320 << 20) // max(1, cout * K * 2)) + 1))
    wts = [(torch.randn(cout, K, device=dev) * 0.02).to(F16) for _ in range(ncopy)]
    wtl = [ops.tile_weights(t, k) for t in wts]
    bias = torch.randn(cout, device=dev)
    out = torch.empty(b, h * w, cout, dtype=F16, device=dev)
    res = {}
    for tiled in (False, True, False, True):
        def run(i):
            ops.igemm(x, (wtl if tiled else wts)[i % ncopy], cout, batch=b, hin=h, win=w, hout=h, wout=w, c0=cin, ksize=k, bias=bias,
                      out=out, ws=ws, w_tiled=tiled)
        with torch.cuda.stream(side):
            run(0)
            side.synchronize()
            if tiled and (False not in res or True not in res):
                ref = out.clone()
                ops.igemm(x, wts[0], cout, batch=b, hin=h, win=w, hout=h, wout=w, c0=cin, ksize=k, bias=bias, out=out, ws=ws)
                side.synchronize()
                assert torch.equal(ref, out), "tiled != row-major"
            g = ops.Graph()
            g.begin()
            for i in range(REPS):
                run(i)
            g.end()
            g.launch()
            side.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                g.launch()
                e1.record(side)
                side.synchronize()
                best = min(best, e0.elapsed_time(e1) / REPS * 1e3)
            g.destroy()
        res.setdefault(tiled, []).append(best)
    r, t = min(res[False]), min(res[True])
    print("M=%-6d N=%-5d K=%-6d ks=%d  row-major %7.1f us (%5.2f TB/s of W)   tiled %7.1f us (%5.2f TB/s)   %+5.1f %%" % (
        M, cout, K, k, r, cout * K * 2 / r * 1e-6, t, cout * K * 2 / t * 1e-6, (r / t - 1) * 100), flush=True)
