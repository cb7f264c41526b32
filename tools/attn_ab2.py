"""Round-4 attention: polynomial-exp2 variants of the round-4 loop (MD_ATTN_ABL 128 = all score pairs, 256 = every second pair)
against the v_exp_f32 form -- accuracy at the production shape, then interleaved timing.  GPU box only.
NOTE: MD_ATTN_ABL (128 / 256 = polynomial exp2) existed only on the day of the run (gpurun r4j); the variants were removed.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_ab import *   # noqa: F401,F403
variants = [(1, 0), (1, 128), (1, 256)]
print("== accuracy (max abs / rms error against fp32 softmax attention)", flush=True)
for name, spike in (("b2", False), ("b2", True)):
    run, out, ten = make(SHAPES[name], seed=1, spike=spike)
    ref = reference(SHAPES[name], ten)
    for v, abl in variants:
        setv(v, abl); out.zero_(); run(); torch.cuda.synchronize()
        e = out.float() - ref
        print(f"  {name} spike={spike} V={v} ABL={abl}: max abs {float(e.abs().max()):.3e} rms {float(e.pow(2).mean().sqrt()):.3e} (ref max {float(ref.abs().max()):.3f} rms {float(ref.pow(2).mean().sqrt()):.3e})", flush=True)
print("== timing", flush=True)
for name in ("b2", "b16"):
    run, out, ten = make(SHAPES[name])
    res = {k: [] for k in variants}
    for rnd in range(3):
        for k in variants:
            setv(*k); res[k].append(time_us(run))
    for k in variants:
        print(f"  {name} V={k[0]} ABL={k[1]}: min {min(res[k]):.1f} us ({tf(SHAPES[name], min(res[k])):.0f} TF)  all {[round(x, 1) for x in res[k]]}", flush=True)
setv(0)
