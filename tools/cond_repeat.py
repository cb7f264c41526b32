"""Debug: repeat the cond / uncond apply_model of the full-size model and report bitwise repeatability."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from magicdance_amd import synthetic
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
model = bench.build_model(dev, 64)
inp = synthetic.synth_inputs((64, 64), frames=1, seed=0, device=dev)
c = {"c_concat": [inp["pose"]], "c_crossattn": [inp["ctx"]], "image_control": [inp["ref"]], "wonoise": True, "overlap_sampling": False}
t = torch.full((1,), 981, dtype=torch.long, device=dev)
outs = []
for rep in range(8):
    e = model.apply_model(inp["x_T"], t, c, inp["ref"]).clone()
    u = model.apply_model(inp["x_T"], t, c, None, uc=True).clone()
    outs.append((e, u))
for i in range(1, 8):
    pass
print(os.environ.get("TAG", ""), "cond mismatches", sum(not torch.equal(outs[0][0], o[0]) for o in outs[1:]), "/7",
      "uncond mismatches", sum(not torch.equal(outs[0][1], o[1]) for o in outs[1:]), "/7", flush=True)
