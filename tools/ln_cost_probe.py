"""What does the folded LayerNorm cost a GEMM?  The same md_igemm shape with and without ``ln`` (row statistics by v_dot2 in the k-loop +
rank-1 correction in the epilogue), graph-timed, tuned tile choice.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16
side = torch.cuda.Stream()
ws = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
SHAPES = [(16, 4096, 320, 960, 0), (16, 4096, 320, 2560, 2), (16, 1024, 640, 1920, 0), (16, 1024, 640, 5120, 2), (16, 256, 1280, 3840, 0),
          (16, 256, 1280, 10240, 2), (2, 4096, 320, 960, 0), (2, 4096, 320, 2560, 2), (2, 1024, 640, 5120, 2), (2, 256, 1280, 10240, 2)]
for (b, n, c, nout, act) in SHAPES:
    x = torch.randn(b, n, c, device=dev).to(F16)
    w = (torch.randn(nout, c, device=dev) * 0.02).to(F16)
    s1, s0 = w.float().sum(1).contiguous(), torch.randn(nout, device=dev)
    out = torch.empty(b, n, nout // 2 if act == 2 else nout, dtype=F16, device=dev)
    res = {}
    for tag, ln in (("plain", None), ("ln", (s1, s0, 1e-5))):
        def run():
            ops.igemm(x, w, nout, batch=b, hin=1, win=n, hout=1, wout=n, c0=c, bias=None if ln else s0, act=act, out=out, ld_out=out.shape[-1], ws=ws, ln=ln)
        with torch.cuda.stream(side):
            run(); side.synchronize()
            g = ops.Graph(); g.begin()
            for _ in range(20): run()
            g.end(); g.launch(); side.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side); g.launch(); e1.record(side); side.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            g.destroy()
        res[tag] = best
    fl = 2.0 * b * n * c * nout
    print(f"M={b * n:6d} K={c:5d} N={nout:6d} act={act}: plain {res['plain']:7.1f} us ({fl / res['plain'] / 1e6:5.0f} TF)   ln {res['ln']:7.1f} us ({fl / res['ln'] / 1e6:5.0f} TF)   ln / plain {res['ln'] / res['plain']:.3f}", flush=True)
