"""Do two torch streams overlap on this runtime?  A chain of small-grid kernels (one-head attention: 32 workgroups of a 256-CU chip)
on one stream, alone and next to the same chain on a second stream; then the step-like chain next to a chain of full-grid kernels
(the table pass's situation).  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magicdance_amd import ops
dev = torch.device("cuda:0"); F16 = torch.float16


def attn_args(b, heads, n):
    c = heads * 40
    q = torch.randn(b, n, c, device=dev).to(F16); k = torch.randn(b, n, c, device=dev).to(F16); vt = torch.randn(b, c, n, device=dev).to(F16)
    out = torch.empty(b, n, c, dtype=F16, device=dev)
    return lambda: ops.attention(q, k, vt, out, batch=b, heads=heads, nq=n, d=40, n0=n, ld_q=c, ld_k0=c, ld_vt0=n, ld_out=c, q_bs=n * c,
                                 k0_bs=n * c, vt0_bs=c * n, out_bs=n * c)


small_a, small_b = attn_args(1, 1, 4096), attn_args(1, 1, 4096)     # 64 workgroups of 4 waves each
big = attn_args(8, 8, 4096)                                          # 2048 workgroups
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("streams", s1.cuda_stream, s2.cuda_stream, "GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))


def timed(jobs, reps=40):
    """jobs: list of (stream, fn); every fn is launched reps times on its stream; wall time until both streams are done"""
    for st, fn in jobs:
        with torch.cuda.stream(st):
            fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for st, fn in jobs:
            with torch.cuda.stream(st):
                fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for name, jobs in (("small chain alone", [(s1, small_a)]), ("two small chains, two streams", [(s1, small_a), (s2, small_b)]),
                   ("two small chains, ONE stream", [(s1, small_a), (s1, small_b)]), ("big chain alone", [(s2, big)]),
                   ("small chain + big chain, two streams", [(s1, small_a), (s2, big)]),
                   ("small chain + big chain, ONE stream", [(s1, small_a), (s1, big)])):
    ts = [timed(jobs) for _ in range(3)]
    print(f"{name:42s} {min(ts):8.2f} ms", flush=True)
