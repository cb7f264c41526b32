"""CPU tier: the drop-in seam of INTEGRATION.md section A, exercised from the REFERENCE's side.

The reference builds its model with ``instantiate_from_config`` (model_lib/ControlNet/ldm/util.py:72-87), which imports whatever
``target:`` the YAML names, and samples with ``DDIMSampler_ReferenceOnly`` (ldm/models/diffusion/ddim.py:346-729), which needs
nothing of the model but ``apply_model`` / ``q_sample`` / the schedule buffers.  Here the *unmodified* reference loader reads this
repo's YAML (same keys, six ``target:`` strings changed), so it instantiates magicdance_amd's classes, and the *unmodified*
reference sampler drives them; the result must be the golden the reference produced with its own classes (tests/golden/small_b1,
oracle/make_golden.py).  Kernels are emulated on the CPU (tests/hip_emulator.py: test infrastructure) -- the GPU tier checks them;
this checks the seam.  /root/reference exists only in the build container: skipped elsewhere."""
import numpy as np
import pytest
import torch
import yaml

from oracle import ref_shim
from tests import helpers as H
from tests import hip_emulator

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="the reference tree exists only in the build container")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


def _repo_yaml(g):
    import magicdance_amd as M
    cfg = yaml.safe_load(open(M.DEFAULT_CONFIG))["model"]      # plain dicts, as oracle/ref_shim.py feeds the reference's own YAML
    mc, nh = int(g["geo_model_channels"]), int(g["geo_num_heads"])
    for blk in ("appearance_control_stage_config", "pose_control_stage_config", "unet_config"):
        cfg["params"][blk]["params"].update(model_channels=mc, num_heads=nh)
    cfg["params"]["first_stage_config"] = "__is_first_stage__"   # the reference loader's own escape hatches (ldm/util.py:73-77)
    cfg["params"]["cond_stage_config"] = "__is_unconditional__"
    cfg["params"]["image_size"] = int(g["side"])
    return cfg, mc, nh


def test_reference_loader_and_sampler_drive_the_repo_classes(monkeypatch):
    hip_emulator.install(monkeypatch)
    ref = ref_shim.load_reference()
    g = H.load_golden("small_b1")
    cfg, mc, nh = _repo_yaml(g)
    assert cfg["target"] == "magicdance_amd.cldm.ControlLDMReferenceOnlyPose"
    model = ref.instantiate_from_config(cfg)                   # the REFERENCE's loader: ldm/util.py:72-87
    assert type(model).__module__ == "magicdance_amd.cldm" and type(model).__name__ == "ControlLDMReferenceOnlyPose"
    for name, want in (("appearance_control_model", "ControlNetReferenceOnly"), ("pose_control_model", "ControlNet")):
        assert type(getattr(model, name)).__name__ == want and type(getattr(model, name)).__module__.startswith("magicdance_amd")
    assert type(model.model.diffusion_model).__name__ == "ControlledUnetModelAttnPose"
    sd = H.synth_weights(mc, nh, seed=int(g["seed"]), device="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if k.startswith(("model.", "appearance", "pose"))]
    model = model.eval()
    inp = H.case_inputs(g)
    steps, side = int(g["steps"]), int(g["side"])
    sampler = ref.ddim.DDIMSampler_ReferenceOnly(model)        # the REFERENCE's sampler: ldm/models/diffusion/ddim.py:346-729
    traj = []
    z, inter = sampler.sample(S=steps, batch_size=1, shape=(4, side, side), conditioning=inp["c"], verbose=False, eta=0.0,
                              unconditional_guidance_scale=7, unconditional_conditioning=inp["uc"], x_T=inp["x_T"],
                              img_callback=lambda p0, i: traj.append(p0.clone()))
    assert _rel(z.numpy(), g["z"]) <= 2e-2                                       # the reference's own result with its own classes
    assert _rel(torch.stack(traj).numpy(), g["pred_x0_traj"]) <= 2e-2
    from magicdance_amd import ddim
    orig_init = ddim.FusedStepRunner.__init__

    def init(self, model_):   # (graph capture needs the GPU: the fused route as a plain launch sequence)
        orig_init(self, model_)
        self.use_graph = False
    monkeypatch.setattr(ddim.FusedStepRunner, "__init__", init)
    # ... and the repo's sampler on the same model object agrees with the reference's sampler on it (same apply_model calls)
    z2, _ = model.sample_log(cond=inp["c"], batch_size=1, ddim=True, ddim_steps=steps, eta=0.0, unconditional_guidance_scale=7,
                             unconditional_conditioning=inp["uc"], inpaint=None, x_T=inp["x_T"])
    assert _rel(z2.numpy(), z.numpy()) <= 1e-2


def test_reference_loader_builds_the_stage1_class():
    """models/cldm_v15_reference_only.yaml (stage 1: appearance control only) through the reference's loader"""
    import magicdance_amd as M
    ref = ref_shim.load_reference()
    cfg = yaml.safe_load(open(M.DEFAULT_CONFIG.replace("_pose.yaml", ".yaml")))["model"]
    for blk in ("control_stage_config", "unet_config"):
        cfg["params"][blk]["params"].update(model_channels=32, num_heads=2)
    cfg["params"]["first_stage_config"] = "__is_first_stage__"
    cfg["params"]["cond_stage_config"] = "__is_unconditional__"
    with torch.device("meta"):
        model = ref.instantiate_from_config(cfg)
    assert type(model).__module__ == "magicdance_amd.cldm" and type(model).__name__ == "ControlLDMReferenceOnly"
