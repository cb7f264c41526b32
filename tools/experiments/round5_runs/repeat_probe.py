"""run S3 diagnostic: what makes two sampler calls on one model differ (run S2: every call after `model._fused = None` + fresh device
copies of the inputs gave another latent)?  Same tensors / fresh tensors x runner kept / runner dropped, consecutive results compared."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from tests import helpers as H   # noqa: E402

dev = torch.device("cuda:0")
g = H.load_golden("c1_b1_s50")
inp = H.case_inputs(g)
mv = lambda d: {k: ([t.to(dev) for t in v] if isinstance(v, list) else v) for k, v in d.items()}  # noqa: E731
model = H.build_hip_model(320, 8, seed=0, device=dev, image_size=64)
fixed = (mv(inp["c"]), mv(inp["uc"]), inp["x_T"].to(dev))
prev = None
for tag, fresh, drop in (("first", False, False), ("same tensors, runner kept", False, False), ("same tensors, runner kept", False, False),
                         ("same tensors, runner dropped", False, True), ("same tensors, runner dropped", False, True),
                         ("fresh tensors, runner kept", True, False), ("fresh tensors, runner kept", True, False),
                         ("fresh tensors, runner dropped", True, True), ("fresh tensors, runner dropped", True, True)):
    if drop:
        model._fused = None
    c, uc, x_T = (mv(inp["c"]), mv(inp["uc"]), inp["x_T"].to(dev)) if fresh else fixed
    z, _ = model.sample_log(cond=c, batch_size=1, ddim=True, ddim_steps=4, eta=0.0, unconditional_guidance_scale=7,
                            unconditional_conditioning=uc, inpaint=None, x_T=x_T)
    torch.cuda.synchronize()
    z = z.clone()
    st = model._fused
    info = f"merge_pose={getattr(st, 'merge_pose', None)} table_mode={getattr(st, 'table_mode', None)} graph={getattr(st, 'use_graph', None)}"
    if prev is not None:
        d = (z - prev).abs()
        print(f"{tag}: equal to the previous call {bool(torch.equal(z, prev))} (differing {int((d > 0).sum())}, max |diff| {float(d.max()):.3e}) {info}", flush=True)
    else:
        print(f"{tag}: max|z| {float(z.abs().max()):.2f} {info}", flush=True)
    prev = z
