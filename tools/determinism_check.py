"""Debug: which network pass is run-to-run non-deterministic (full-size model, eager)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from magicdance_amd import synthetic
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
model = bench.build_model(dev, 64)
inp = synthetic.synth_inputs((64, 64), frames=1, seed=0, device=dev)
t = torch.full((1,), 981, dtype=torch.long, device=dev)
def banks():
    b = []
    model.appearance_control_model(x=inp["ref"], hint=None, timesteps=t, context=inp["ctx"], attention_bank=b, attention_mode="write")
    return [e[0] for e in b]
def pose():
    return model.pose_control_model(x=inp["x_T"], hint=inp["pose"], timesteps=t, context=inp["ctx"])
b1, b2, b3 = banks(), banks(), banks()
for i, (x, y, z) in enumerate(zip(b1, b2, b3)):
    print("bank", i, tuple(x.shape), "equal12", torch.equal(x, y), "equal13", torch.equal(x, z), "maxdiff", float((x.float() - y.float()).abs().max()), flush=True)
p1, p2 = pose(), pose()
for i, (x, y) in enumerate(zip(p1, p2)):
    print("pose", i, tuple(x.shape), "equal", torch.equal(x, y), "maxdiff", float((x - y).abs().max()), flush=True)
unet = model.model.diffusion_model
def eps(uc):
    return unet(x=inp["x_T"], timesteps=t, context=inp["ctx"], control=None if uc else [[e] for e in b1], pose_control=None if uc else [p.clone() for p in p1],
                only_mid_control=False, attention_mode="read", uc=uc)
for uc in (True, False):
    e1, e2 = eps(uc), eps(uc)
    print("unet uc" if uc else "unet read", "equal", torch.equal(e1, e2), "maxdiff", float((e1 - e2).abs().max()), flush=True)
