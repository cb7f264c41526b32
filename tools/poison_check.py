"""Debug: run one eager apply_model pair of the full-size model with a NaN-poisoned arena and per-op finiteness checks."""
import os, sys
os.environ["MD_ARENA_POISON"] = "1"; os.environ["MD_DEBUG_FINITE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from magicdance_amd import synthetic
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = bench.build_model(dev, size)
inp = synthetic.synth_inputs((size, size), frames=1, seed=0, device=dev)
c = {"c_concat": [inp["pose"]], "c_crossattn": [inp["ctx"]], "image_control": [inp["ref"]], "wonoise": True, "overlap_sampling": False}
t = torch.full((1,), 981, dtype=torch.long, device=dev)
for rep in range(2):
    e = model.apply_model(inp["x_T"], t, c, inp["ref"]); print("cond ok", float(e.abs().max()), flush=True)
    e = model.apply_model(inp["x_T"], t, c, None, uc=True); print("uncond ok", float(e.abs().max()), flush=True)
print("no uninitialised read detected")
