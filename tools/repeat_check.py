"""Run-to-run determinism of the whole product path (GPU box only): N full samplings (table pass overlapped with the loop,
captured step graph, decode) of the same inputs must be bit-identical.  usage: python tools/repeat_check.py [N] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from magicdance_amd import synthetic, parallel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fpg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
model = bench.build_model(dev, 64)
inp = synthetic.synth_inputs((64, 64), frames=fpg, seed=0, device=dev)
runner = parallel.FrameShardedSampler(model)
x_T = inp["x_T"].repeat(fpg, 1, 1, 1)
ref = None
bad = 0
for i in range(n):
    img = runner.sample(inp["pose"], inp["ctx"], inp["ref"], x_T, ddim_steps=50, scale=7.0, decode=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(img).all()), f"run {i}: non-finite output"
    if ref is None:
        ref = img.clone()
    elif not torch.equal(img, ref):
        bad += 1
        print(f"run {i}: MISMATCH max-abs diff {float((img - ref).abs().max()):.3e}", flush=True)
print(f"{n} runs of {fpg} frame(s): {bad} mismatching the first run", flush=True)
sys.exit(1 if bad else 0)
