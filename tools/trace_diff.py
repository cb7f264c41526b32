"""Debug: checksum after every engine op over repeated identical passes; report the first op whose output differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from magicdance_amd import synthetic, engine
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
model = bench.build_model(dev, 64)
inp = synthetic.synth_inputs((64, 64), frames=1, seed=0, device=dev)
c = {"c_concat": [inp["pose"]], "c_crossattn": [inp["ctx"]], "image_control": [inp["ref"]], "wonoise": True, "overlap_sampling": False}
t = torch.full((1,), 981, dtype=torch.long, device=dev)
sync = os.environ.get("TRACE_SYNC", "1") == "1"
traces = []
model.apply_model(inp["x_T"], t, c, inp["ref"])  # warm caches
for rep in range(int(os.environ.get("REPS", "10"))):
    engine._TRACE = []
    model.apply_model(inp["x_T"], t, c, inp["ref"])
    model.apply_model(inp["x_T"], t, c, None, uc=True)
    traces.append(engine._TRACE)
engine._TRACE = None
ref = traces[0]
nbad = 0
for r, tr in enumerate(traces[1:], 1):
    for i, (a, b) in enumerate(zip(ref, tr)):
        if a != b:
            nbad += 1
            print(f"rep {r}: first diff at op {i}/{len(ref)}: {a[0]} | {a[1]:.6f} vs {b[1]:.6f}; prev op: {ref[i-1][0]}", flush=True)
            break
print("differing reps:", nbad, "of", len(traces) - 1)
