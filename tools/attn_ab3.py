"""Round-4 attention loop on the other production shapes (64-row query blocks at d = 40, d = 80): V=0 vs V=1, interleaved.  GPU box only.
NOTE: MD_ATTN_V existed only on the day of the run (gpurun r4k).
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_ab import *   # noqa: F401,F403
SH = {"d40 B=1 self": (1, 8, 4096, 4096, 0, 0, 40), "d40 B=1 bank": (1, 8, 4096, 4096, 4096, 1, 40), "d80 B=2": (2, 8, 1024, 1024, 1024, 1, 80),
      "d80 B=3": (3, 8, 1024, 1024, 1024, 1, 80), "d80 B=16": (16, 8, 1024, 1024, 1024, 8, 80), "d80 B=24": (24, 8, 1024, 1024, 1024, 8, 80),
      "d40 768^2 B=2": (2, 8, 9216, 9216, 9216, 1, 40), "d40 B=3": (3, 8, 4096, 4096, 4096, 1, 40)}
for name, shape in SH.items():
    run, out, ten = make(shape, seed=2)
    if shape[0] <= 3 and shape[2] <= 4096:
        ref = reference(shape, ten)
        for v in (0, 1):
            setv(v); out.zero_(); run(); torch.cuda.synchronize()
            print(f"  {name} V={v}: max abs err {float((out.float() - ref).abs().max()):.3e}", flush=True)
        del ref
    res = {0: [], 1: []}
    for rnd in range(3):
        for v in (0, 1):
            setv(v); res[v].append(time_us(run))
    print(f"  {name}: V=0 {min(res[0]):.1f} us ({tf(shape, min(res[0])):.0f} TF)   V=1 {min(res[1]):.1f} us ({tf(shape, min(res[1])):.0f} TF)   ratio {min(res[0]) / min(res[1]):.3f}", flush=True)
setv(0)
