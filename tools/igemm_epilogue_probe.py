"""Short-K GEMMs of an 8-frame step (GPU box only): how much of their time is the epilogue's global traffic?  M = 65536 tokens,
K = 320; variants: plain store, + residual, + two-term residual (res_lo / out_lo), fp32 out, GEGLU; graph-timed, rotating outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from magicdance_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
F16 = torch.float16
side = torch.cuda.Stream()
REPS = 10


def timed(fn):
    with torch.cuda.stream(side):
        fn(0)
        side.synchronize()
        g = ops.Graph()
        g.begin()
        for i in range(REPS):
            fn(i)
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
    return us


M = 65536
for (N, K, act, cfgs) in ((320, 320, 0, (24, 25, 27, 12)), (640, 640, 0, (24, 25, 12)), (2560, 320, 2, (14, 12)), (320, 1280, 0, (25, 24))):
    x = [torch.randn(1, M, K, device=dev).to(F16) for _ in range(4)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(F16)
    bias = torch.randn(N, device=dev)
    nout = N // 2 if act == 2 else N
    outs = [torch.empty(1, M, nout, dtype=F16, device=dev) for _ in range(4)]
    los = [torch.empty(1, M, nout, dtype=F16, device=dev) for _ in range(4)]
    res = [torch.randn(1, M, nout, device=dev).to(F16) for _ in range(4)]
    rlo = [(torch.randn(1, M, nout, device=dev) * 1e-4).to(F16) for _ in range(4)]
    o32 = [torch.empty(1, M, nout, dtype=torch.float32, device=dev) for _ in range(2)]
    for cfg in cfgs:
        kw = dict(batch=1, hin=1, win=M, hout=1, wout=M, c0=K, bias=bias, act=act, ld_out=nout, force_cfg=cfg)
        variants = {"plain": lambda i: ops.igemm(x[i % 4], w, N, out=outs[i % 4], **kw)}
        if act != 2:
            variants["res"] = lambda i: ops.igemm(x[i % 4], w, N, out=outs[i % 4], res=res[i % 4], ld_res=nout, **kw)
            variants["res2"] = lambda i: ops.igemm(x[i % 4], w, N, out=outs[i % 4], res=res[i % 4], ld_res=nout, res_lo=rlo[i % 4],
                                                   out_lo=los[i % 4], **kw)
            variants["f32"] = lambda i: ops.igemm(x[i % 4], w, N, out=o32[i % 2], out_f32=True, **kw)
        line = f"M={M} N={N} K={K} act={act} cfg={cfg}:"
        for name, fn in variants.items():
            us = timed(fn)
            byts = M * K * 2 + M * nout * 2 * {"plain": 1, "res": 2, "res2": 4, "f32": 2}[name]
            line += f"  {name} {us:6.1f} us ({2.0 * M * N * K / us / 1e6:4.0f} TF, {byts / us / 1e3:5.0f} GB/s)"
        print(line, flush=True)
