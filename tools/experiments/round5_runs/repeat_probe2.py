"""run S4 diagnostic: the full-width sampler is not repeatable call to call (run S3).  One process per switch setting: three calls on one
model / one runner / the same device tensors; reports whether call 2 and 3 reproduce call 1.  argv: steps [nograph]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
from tests import helpers as H   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nograph = len(sys.argv) > 2
dev = torch.device("cuda:0")
g = H.load_golden("c1_b1_s50")
inp = H.case_inputs(g)
mv = lambda d: {k: ([t.to(dev) for t in v] if isinstance(v, list) else v) for k, v in d.items()}  # noqa: E731
model = H.build_hip_model(320, 8, seed=0, device=dev, image_size=64)
c, uc, x_T = mv(inp["c"]), mv(inp["uc"]), inp["x_T"].to(dev)
if nograph:
    from magicdance_amd import ddim
    orig = ddim.FusedStepRunner.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.use_graph = False
    ddim.FusedStepRunner.__init__ = init
zs = []
for _ in range(3):
    z, _ = model.sample_log(cond=c, batch_size=1, ddim=True, ddim_steps=steps, eta=0.0, unconditional_guidance_scale=7,
                            unconditional_conditioning=uc, inpaint=None, x_T=x_T)
    torch.cuda.synchronize()
    zs.append(z.clone())
sw = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MD_"))
d = [float((zs[i] - zs[0]).abs().max()) for i in (1, 2)]
print(f"PROBE steps={steps} graph={not nograph} [{sw}]: calls 2, 3 equal call 1: {torch.equal(zs[1], zs[0])}, {torch.equal(zs[2], zs[0])}; max |diff| {d[0]:.3e}, {d[1]:.3e} "
      f"at max|z| {float(zs[0].abs().max()):.2f}", flush=True)
