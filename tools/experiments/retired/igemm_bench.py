"""Micro-benchmark of md_igemm tile configs / split-K on the layer shapes of one DDIM step (GPU box only).
usage: python tools/igemm_bench.py [out.tsv]   -- prints TFLOP/s per (shape, cfg, split)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops

dev = torch.device("cuda:0")
F16 = torch.float16
# (B, H, W, Cin, Cout, ksize)  -> M = B*H*W
SHAPES = [
    (2, 64, 64, 320, 320, 3), (1, 64, 64, 320, 320, 3), (2, 64, 64, 640, 320, 3), (2, 64, 64, 960, 320, 3),
    (2, 32, 32, 640, 640, 3), (1, 32, 32, 640, 640, 3), (2, 32, 32, 1920, 640, 3),
    (2, 16, 16, 1280, 1280, 3), (1, 16, 16, 1280, 1280, 3), (2, 16, 16, 2560, 1280, 3),
    (2, 8, 8, 1280, 1280, 3), (1, 8, 8, 1280, 1280, 3), (2, 8, 8, 2560, 1280, 3),
    (2, 64, 64, 320, 320, 1), (1, 64, 64, 320, 320, 1), (2, 64, 64, 320, 2560, 1), (2, 64, 64, 1280, 320, 1), (2, 64, 64, 320, 960, 1),
    (2, 32, 32, 640, 640, 1), (1, 32, 32, 640, 640, 1), (2, 32, 32, 640, 5120, 1), (2, 32, 32, 2560, 640, 1),
    (2, 16, 16, 1280, 1280, 1), (1, 16, 16, 1280, 1280, 1), (2, 16, 16, 1280, 10240, 1), (2, 16, 16, 5120, 1280, 1), (2, 16, 16, 1280, 3840, 1),
    (2, 8, 8, 1280, 1280, 1), (1, 8, 8, 1280, 1280, 1),
]
CFGS = [4, 5, 7, 12, 13, 14, 15, 16, 17, 19]
SPLITS = [1, 2, 4, 8]
ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
REPS = 20
out_f = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
for (b, h, w, cin, cout, k) in SHAPES:
    x = torch.randn(b, h * w, cin, device=dev).to(F16)
    wt = (torch.randn(cout, k * k * cin, device=dev) * 0.02).to(F16)
    bias = torch.randn(cout, device=dev)
    out = torch.empty(b, h * w, cout, dtype=F16, device=dev)
    M, K = b * h * w, k * k * cin
    flops = 2.0 * M * cout * K
    res = []
    for cfg, sp in itertools.product(CFGS, SPLITS):
        if sp > 1 and (M > 2048 or K // 64 // sp < 4):
            continue
        def run():
            ops.igemm(x, wt, cout, batch=b, hin=h, win=w, hout=h, wout=w, c0=cin, ksize=k, bias=bias, out=out, ws=ws,
                      force_cfg=cfg, force_splitk=sp)
        try:
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
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                g.launch()
                e1.record(side)
                side.synchronize()
                us = e0.elapsed_time(e1) / REPS * 1e3
                g.destroy()
        except Exception as ex:  # noqa
            print("ERR", cfg, sp, ex)
            us = float("nan")
        res.append((us, cfg, sp))
    res.sort()
    line = f"M={M} N={cout} K={K} ks={k}: " + "  ".join(f"c{c}/s{s}:{u:.1f}us({flops / u / 1e6:.0f}TF)" for u, c, s in res[:6])
    worst = res[-1]
    line += f"  | worst c{worst[1]}/s{worst[2]}:{worst[0]:.1f}us"
    print(line, flush=True)
    if out_f:
        for u, c, s in res:
            out_f.write(f"{M}\t{cout}\t{K}\t{k}\t{c}\t{s}\t{u:.2f}\t{flops / u / 1e6:.1f}\n")
