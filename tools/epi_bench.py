"""Epilogue-bound md_igemm shapes, graph-timed with the launcher's own (tuned) tile choice (GPU box only).
usage: MD_IGEMM_STAGE=0|1 python tools/epi_bench.py   -- one line per shape: us, TFLOP/s, GB/s of the algorithmic bytes.
Shapes: (batch, h, w, cin, cout, ksize, residual stream terms [0 | 1 | 2], per-sample bias) -- the 1x1 / linear layers with a
5..20-tile k-loop per output tile are the ones whose epilogue dominates."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops

dev = torch.device("cuda:0")
F16 = torch.float16
SHAPES = [
    (2, 64, 64, 320, 320, 1, 2, False), (2, 64, 64, 1280, 320, 1, 2, False), (2, 64, 64, 320, 320, 1, 0, False),
    (2, 32, 32, 640, 640, 1, 2, False), (2, 32, 32, 2560, 640, 1, 2, False),
    (2, 16, 16, 1280, 1280, 1, 2, False), (2, 16, 16, 5120, 1280, 1, 2, False), (2, 8, 8, 1280, 1280, 1, 2, False),
    (24, 64, 64, 320, 320, 1, 2, False), (24, 64, 64, 1280, 320, 1, 2, False), (16, 64, 64, 320, 320, 1, 2, False),
    (24, 32, 32, 640, 640, 1, 2, False), (24, 32, 32, 2560, 640, 1, 2, False), (24, 16, 16, 1280, 1280, 1, 2, False),
    (2, 64, 64, 320, 320, 3, 0, True), (2, 64, 64, 320, 320, 3, 1, False), (2, 32, 32, 640, 640, 3, 1, False),
    (2, 16, 16, 1280, 1280, 3, 1, False), (24, 64, 64, 320, 320, 3, 0, True), (24, 64, 64, 320, 320, 3, 1, False),
    (24, 32, 32, 640, 640, 3, 1, False), (24, 64, 64, 640, 320, 3, 0, True),
    # GEGLU feed-forward input projections (cout = 8 cin packed a | gate rows, output [M][4 cin]): residual terms = -1
    (2, 64, 64, 320, 2560, 1, -1, False), (2, 32, 32, 640, 5120, 1, -1, False), (2, 16, 16, 1280, 10240, 1, -1, False),
    (24, 64, 64, 320, 2560, 1, -1, False), (16, 64, 64, 320, 2560, 1, -1, False), (24, 32, 32, 640, 5120, 1, -1, False),
    (24, 16, 16, 1280, 10240, 1, -1, False),
]
ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
REPS = 20
print("# MD_IGEMM_STAGE=%s" % os.environ.get("MD_IGEMM_STAGE", "(default 1)"))
for (b, h, w, cin, cout, k, nres, pbias) in SHAPES:
    x = torch.randn(b, h * w, cin, device=dev).to(F16)
    wt = (torch.randn(cout, k * k * cin, device=dev) * 0.02).to(F16)
    bias = torch.randn(b if pbias else 1, cout, device=dev)
    geglu = nres < 0
    nres = max(nres, 0)
    out = torch.empty(b, h * w, cout // 2 if geglu else cout, dtype=F16, device=dev)
    out_lo = torch.empty_like(out) if nres == 2 else None
    res = torch.randn(b, h * w, cout, device=dev).to(F16) if nres else None
    res_lo = (torch.randn(b, h * w, cout, device=dev) * 1e-3).to(F16) if nres == 2 else None
    M, K = b * h * w, k * k * cin
    flops = 2.0 * M * cout * K
    byts = 2.0 * (M * cin + cout * K + M * out.shape[-1] * (1 + (2 if nres == 2 else nres) + (1 if nres == 2 else 0)))

    def run():
        ops.igemm(x, wt, cout, batch=b, hin=h, win=w, hout=h, wout=w, c0=cin, ksize=k, bias=bias,
                  bias_batch_stride=cout if pbias else 0, res=res, ld_res=cout if nres else 0, res_lo=res_lo, out_lo=out_lo,
                  out=out, ld_out=out.shape[-1], ws=ws, act=2 if geglu else 0)
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
        side.synchronize()
        g = ops.Graph()
        g.begin()
        for _ in range(REPS):
            run()
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
    print("M=%-6d N=%-5d K=%-6d ks=%d res=%d pb=%d %s %8.1f us  %7.1f TFLOP/s  %7.0f GB/s" % (M, cout, K, k, nres, int(pbias), "geglu" if geglu else "     ", best,
                                                                                      flops / best * 1e-6, byts / best * 1e-3), flush=True)
