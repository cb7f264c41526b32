"""md_ff_block against the three md_igemm launches it replaces (to_out + residual, folded-LN GEGLU projection, feed-forward output +
residual), event-timed on the GPU box.  usage: python tools/ffblock_bench.py > gpurun_out/ffblock_bench.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from magicdance_amd import ops, _lib  # noqa: E402
if os.environ.get("MD_HIP_LIB"):   # an experiment build of the library (ablation variants of ffblock.hip)
    _lib.LIB_PATH = os.environ["MD_HIP_LIB"]
import test_gpu_ffblock as T  # noqa: E402

dev = torch.device("cuda:0")
F16 = torch.float16


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


shapes = [(320, 2, 4096), (320, 3, 4096), (320, 16, 4096), (320, 24, 4096), (640, 2, 1024), (640, 3, 1024), (640, 16, 1024), (640, 24, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for c, b, n in shapes:
    m = b * n
    pk = T.make_params(c, 20, dev)["packed"]
    x16 = (T._rand((m, c), 1, dev) + 0.3).to(F16)
    lo16 = T._rand((m, c), 3, dev, 1e-3).to(F16)
    att16 = T._rand((m, c), 2, dev).to(F16)
    out, out_lo = torch.empty((m, c), dtype=F16, device=dev), torch.empty((m, c), dtype=F16, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    t2, t2_lo = torch.empty_like(out), torch.empty_like(out)
    hid = torch.empty((m, 4 * c), dtype=F16, device=dev)

    def unfused_head():
        ops.igemm(att16, pk["wo"], c, batch=b, hin=1, win=n, hout=1, wout=n, c0=c, bias=pk["bo"], res=x16, ld_res=c, res_lo=lo16, out=t2,
                  out_lo=t2_lo, ws=ws, w_tiled=True)

    def unfused_ff():
        ops.igemm(t2, pk["w1"], 8 * c, batch=b, hin=1, win=n, hout=1, wout=n, c0=c, out=hid, ld_out=4 * c, act=ops.MD_ACT_GEGLU,
                  ln=(pk["s1"], pk["s0"], 1e-5), ws=ws, w_tiled=True)
        ops.igemm(hid, pk["w2"], c, batch=b, hin=1, win=n, hout=1, wout=n, c0=4 * c, bias=pk["b2"], res=t2, ld_res=c, res_lo=t2_lo, out=out,
                  out_lo=out_lo, ws=ws, w_tiled=True)

    def unfused_all():
        unfused_head()
        unfused_ff()

    def fused(bm, head):
        kw = dict(attn=att16, wo=pk["wo"], bo=pk["bo"]) if head else {}
        return lambda: ops.ff_block(x16, out, m=m, c=c, w1=pk["w1"], s1=pk["s1"], s0=pk["s0"], w2=pk["w2"], b2=pk["b2"], x_lo=lo16,
                                    out_lo=out_lo, force_bm=bm, **kw)

    flops_ff, flops_head = 2.0 * m * 12 * c * c, 2.0 * m * c * c
    u_all, u_ff = timed(unfused_all), timed(unfused_ff)
    line = [f"C={c} M={m:6d}: md_igemm x3 {u_all:7.1f} us ({(flops_ff + flops_head) / u_all / 1e6:5.0f} TF)  x2 (no to_out) {u_ff:7.1f} us"]
    for bm in ((32, 1032, 2032, 3032, 64, 1064, 2064, 128) if c == 320 else (32, 64)):
        try:
            fh, fn = timed(fused(bm, 1)), timed(fused(bm, 0))
            line.append(f"bm{bm}: head {fh:7.1f} us ({(flops_ff + flops_head) / fh / 1e6:5.0f} TF) ff-only {fn:7.1f} us")
        except Exception as e:  # noqa: BLE001
            line.append(f"bm{bm}: {type(e).__name__}")
    print(" | ".join(line), flush=True)
