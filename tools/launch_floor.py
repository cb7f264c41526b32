"""Per-launch floor of a dependent kernel chain inside a captured HIP graph (GPU box only): what one more launch costs a DDIM step
whatever it computes.  (a) 200 x md_counter_add (one thread of work), (b) 200 x md_igemm with ONE 64-deep k-tile on the 64x64 / 320
channel geometry (512 workgroups: launch + dispatch + prologue + first-tile latency + epilogue), (c) the same with 5 and 45 k-tiles."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from magicdance_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
F16 = torch.float16
side = torch.cuda.Stream()
REPS = 200


def timed(fn, label):
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        g = ops.Graph()
        g.begin()
        for _ in range(REPS):
            fn()
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
    print(f"{label}: {us:.2f} us per launch", flush=True)


cnt = torch.zeros(1, dtype=torch.int32, device=dev)
timed(lambda: ops.counter_add(cnt, 1), "md_counter_add chain")
b, h, n = 2, 64, 320
for ks, cin in ((1, 64), (1, 320), (3, 320)):
    x = torch.randn(b, h * h, cin, device=dev).to(F16)
    w = (torch.randn(n, ks * ks * cin, device=dev) * 0.05).to(F16)
    y = torch.empty(b, h * h, n, dtype=F16, device=dev)
    bias = torch.randn(n, device=dev)
    timed(lambda: ops.igemm(x, w, n, batch=b, hin=h, win=h, hout=h, wout=h, c0=cin, ksize=ks, bias=bias, out=y, force_cfg=27),
          f"md_igemm M=8192 N=320 K={ks * ks * cin} ({ks * ks * cin // 64} k-tiles, 64x80 tiles, 512 workgroups)")
x = torch.randn(b, h * h, 320, device=dev).to(F16)
gma, bta = torch.ones(320, device=dev), torch.zeros(320, device=dev)
o = torch.empty_like(x)
ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
timed(lambda: ops.groupnorm(x, gma, bta, o, ws, batch=b, hw=h * h, c0=320, silu=True), "md_groupnorm 2 x 64x64 x 320 (stats + apply = 2 launches)")
x2 = torch.randn(b, 256, 1280, device=dev).to(F16)
g2, b2 = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
o2 = torch.empty_like(x2)
timed(lambda: ops.groupnorm(x2, g2, b2, o2, ws, batch=b, hw=256, c0=1280, silu=True), "md_groupnorm 2 x 16x16 x 1280 (single launch)")

# split-K: slabs + deterministic reduce kernel (the 16x16 / 8x8 levels of a one-frame step)
ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
for (bb, hh, cin, nn, cfg, sp) in ((2, 16, 1280, 1280, 25, 16), (3, 8, 1280, 1280, 28, 8)):
    xs = torch.randn(bb, hh * hh, cin, device=dev).to(F16)
    wsk = (torch.randn(nn, 9 * cin, device=dev) * 0.01).to(F16)
    ys = torch.empty(bb, hh * hh, nn, dtype=F16, device=dev)
    ylo = torch.empty_like(ys)
    rs = torch.randn(bb, hh * hh, nn, device=dev).to(F16)
    bs = torch.randn(nn, device=dev)
    timed(lambda: ops.igemm(xs, wsk, nn, batch=bb, hin=hh, win=hh, hout=hh, wout=hh, c0=cin, ksize=3, bias=bs, res=rs, ld_res=nn, res_lo=rs,
                            out=ys, out_lo=ylo, ws=ws, force_cfg=cfg, force_splitk=sp),
          f"md_igemm 3x3 M={bb * hh * hh} N={nn} K={9 * cin} split {sp} + reduce (two-term residual)")
