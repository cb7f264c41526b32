"""run S2 diagnostic: which use of conv(gn_next=) changes the 4-step full-width latent (engine._GN_NEXT bit 0 / bit 1), is the sampler
run-to-run deterministic at all, and does the difference survive with the library-side fusion off (MD_GN_REDUCE=0: same caller
plumbing, every GroupNorm as its own launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from tests import helpers as H   # noqa: E402
from magicdance_amd import engine   # noqa: E402

dev = torch.device("cuda:0")
g = H.load_golden("c1_b1_s50")
inp = H.case_inputs(g)
mv = lambda d: {k: ([t.to(dev) for t in v] if isinstance(v, list) else v) for k, v in d.items()}  # noqa: E731
model = H.build_hip_model(320, 8, seed=0, device=dev, image_size=64)
base = None
for tag, val in (("off", 0), ("off again", 0), ("bit0 conv1->gn2", 1), ("bit1 next layer/block", 2), ("both", 3), ("off third", 0)):
    engine._GN_NEXT = val
    model._fused = None
    z, _ = model.sample_log(cond=mv(inp["c"]), batch_size=1, ddim=True, ddim_steps=4, eta=0.0, unconditional_guidance_scale=7,
                            unconditional_conditioning=mv(inp["uc"]), inpaint=None, x_T=inp["x_T"].to(dev))
    torch.cuda.synchronize()
    z = z.clone()
    if base is None:
        base = z
    d = (z - base).abs()
    print(f"MD_GN_REDUCE={os.environ.get('MD_GN_REDUCE', '1')} _GN_NEXT={val} ({tag}): equal to the first run {bool(torch.equal(z, base))}, "
          f"differing {int((d > 0).sum())} of {d.numel()}, max |diff| {float(d.max()):.3e} at max|z| {float(base.abs().max()):.2f}", flush=True)
