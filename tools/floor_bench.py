"""Per-launch floor of dependent kernels inside a HIP graph (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
F16 = torch.float16

def timeit(fn, reps=200, label=""):
    with torch.cuda.stream(side):
        fn(); side.synchronize()
        g = ops.Graph(); g.begin()
        for _ in range(reps): fn()
        g.end(); g.launch(); side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side); g.launch(); e1.record(side); side.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        g.destroy()
    print(f"{label}: {us:.2f} us/launch", flush=True)

counter = torch.zeros(1, dtype=torch.int32, device=dev)
timeit(lambda: ops.counter_add(counter, 1), label="counter_add (1 wave)")
t = torch.tensor([1.0, 2.0], device=dev); out = torch.empty(2, 320, device=dev)
timeit(lambda: ops.timestep_embedding(t, out, 2, 320), label="timestep_embedding")
a = torch.randn(2, 4096, 320, device=dev).to(F16); b = torch.randn_like(a); o = torch.empty_like(a)
timeit(lambda: ops.add_f16(a, b, o, a.numel()), label="add_f16 5.2MB")
g_, b_ = torch.ones(320, device=dev), torch.zeros(320, device=dev)
x = torch.randn(4096, 320, device=dev).to(F16); y = torch.empty_like(x)
timeit(lambda: ops.layernorm(x, g_, b_, y, 4096, 320), label="layernorm 4096x320")
x2 = torch.randn(256, 1280, device=dev).to(F16); y2 = torch.empty_like(x2)
g2, b2 = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
timeit(lambda: ops.layernorm(x2, g2, b2, y2, 256, 1280), label="layernorm 256x1280")
ws = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
xg = torch.randn(2, 4096, 320, device=dev).to(F16); yg = torch.empty_like(xg)
timeit(lambda: ops.groupnorm(xg, g_, b_, yg, ws, batch=2, hw=4096, c0=320, silu=True), label="groupnorm 2x64x64x320 (2 kernels)")
xg2 = torch.randn(2, 64, 1280, device=dev).to(F16); yg2 = torch.empty_like(xg2)
timeit(lambda: ops.groupnorm(xg2, g2, b2, yg2, ws, batch=2, hw=64, c0=1280, silu=True), label="groupnorm 2x8x8x1280 (2 kernels)")
for (m, n, k) in [(64, 64, 64), (64, 1280, 64), (64, 1280, 1280), (4096, 320, 64)]:
    xx = torch.randn(1, m, k, device=dev).to(F16); ww = torch.randn(n, k, device=dev).to(F16); oo = torch.empty(1, m, n, dtype=F16, device=dev)
    for cfg in (15, 12):
        timeit(lambda: ops.igemm(xx, ww, n, batch=1, hin=1, win=m, hout=1, wout=m, c0=k, out=oo, force_cfg=cfg, force_splitk=1),
               label=f"igemm M={m} N={n} K={k} cfg={cfg}")
