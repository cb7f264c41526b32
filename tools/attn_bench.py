"""Graph-timed attention microbench (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16; side = torch.cuda.Stream()
SHAPES = [(2, 8, 4096, 4096, 4096, 1, 40), (1, 8, 4096, 4096, 0, 0, 40), (2, 8, 1024, 1024, 1024, 1, 80),
                                       (1, 8, 1024, 1024, 0, 0, 80), (2, 8, 256, 256, 256, 1, 160), (2, 8, 4096, 77, 0, 0, 40), (2, 8, 1024, 77, 0, 0, 80),
          (16, 8, 4096, 4096, 4096, 8, 40), (16, 8, 1024, 1024, 1024, 8, 80), (16, 8, 256, 256, 256, 8, 160)]
if os.environ.get("ATTN_BENCH_ONLY"):   # e.g. "0" or "0,2": PMC runs profile a single shape
    SHAPES = [SHAPES[int(i)] for i in os.environ["ATTN_BENCH_ONLY"].split(",")]
REPS = int(os.environ.get("ATTN_BENCH_REPS", "10"))
for (b, heads, nq, n0, n1, n1b, d) in SHAPES:
    c = heads * d
    q = torch.randn(b, nq, c, device=dev).to(F16); k0 = torch.randn(b, n0, c, device=dev).to(F16)
    ld0 = (n0 + 7) // 8 * 8
    vt0 = torch.zeros(b, c, ld0, dtype=F16, device=dev); vt0[:, :, :n0] = torch.randn(b, c, n0, device=dev).to(F16)
    kw = {}
    if n1:
        k1 = torch.randn(1, n1, c, device=dev).to(F16); vt1 = torch.randn(1, c, n1, device=dev).to(F16)
        kw = dict(k1=k1, vt1=vt1, n1=n1, ld_k1=c, ld_vt1=n1, k1_bs=0, vt1_bs=0, n1_batches=n1b)
    out = torch.empty(b, nq, c, dtype=F16, device=dev)
    def run():
        ops.attention(q, k0, vt0, out, batch=b, heads=heads, nq=nq, d=d, n0=n0, ld_q=c, ld_k0=c, ld_vt0=ld0, ld_out=c,
                      q_bs=nq * c, k0_bs=n0 * c, vt0_bs=c * ld0, out_bs=nq * c, **kw)
    with torch.cuda.stream(side):
        run(); side.synchronize()
        g = ops.Graph(); g.begin()
        for _ in range(REPS): run()
        g.end(); g.launch(); side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side); g.launch(); e1.record(side); side.synchronize()
        us = e0.elapsed_time(e1) / REPS * 1e3
    nkv = n0 + n1 * n1b / b
    print(f"B={b} nq={nq} n0={n0} n1={n1} d={d}: {us:.1f} us  {4.0 * b * heads * nq * nkv * d / us / 1e6:.0f} TF", flush=True)
