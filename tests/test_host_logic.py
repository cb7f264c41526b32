"""CPU tier: the product's HOST logic (packing, layouts, bank plumbing, sampler control flow, config surface) driven
through a torch-CPU emulation of the C ABI (tests/hip_emulator.py) and checked against the reference goldens.
The kernels themselves are checked on the GPU tier (tests/test_gpu_*.py)."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests import hip_emulator


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


@pytest.mark.parametrize("name", ["small_b1", "small_b2"])
def test_engine_host_logic_matches_reference_golden(monkeypatch, name):
    hip_emulator.install(monkeypatch)
    g = H.load_golden(name)
    mc, nh = int(g["geo_model_channels"]), int(g["geo_num_heads"])
    model = H.build_hip_model(mc, nh, seed=int(g["seed"]), device="cpu", image_size=int(g["side"]))
    inp = H.case_inputs(g)
    frames = int(g["frames"])
    t = torch.full((frames,), int(g["t_probe"]), dtype=torch.long)
    bank = []
    model.appearance_control_model(x=inp["ref"], hint=None, timesteps=t, context=inp["ctx"], attention_bank=bank,
                                   attention_mode="write", uc=False)
    assert len(bank) == 16
    for i, bk in enumerate(bank):
        assert _rel(H.head_slice(bk[0]), g[f"bank{i}_head"]) <= 2e-2, f"bank{i}"
    pr = model.pose_control_model(x=inp["x_T"], hint=inp["pose"], timesteps=t, context=inp["ctx"])
    assert len(pr) == 13
    for i, p in enumerate(pr):
        assert float(np.abs(H.head_slice(p) - g[f"pose{i}_head"]).max()) <= 2e-2 * g[f"pose{i}_sum"][2], f"pose{i}"
    e_c = model.apply_model(inp["x_T"], t, inp["c"], inp["ref"]).numpy()
    e_u = model.apply_model(inp["x_T"], t, inp["c"], None, uc=True).numpy()
    assert _rel(e_c, g["eps_c"]) <= 1e-2 and _rel(e_u, g["eps_u"]) <= 1e-2
    # sample_log: fused route (as a plain launch sequence; graphs need the GPU) and the generic route
    from magicdance_amd import ddim
    monkeypatch.setattr(ddim.FusedStepRunner, "use_graph", False, raising=False)
    orig_init = ddim.FusedStepRunner.__init__

    def init(self, model):
        orig_init(self, model)
        self.use_graph = False
    monkeypatch.setattr(ddim.FusedStepRunner, "__init__", init)
    traj = []
    z, inter = model.sample_log(cond=inp["c"], batch_size=frames, ddim=True, ddim_steps=int(g["steps"]), eta=0.0,
                                unconditional_guidance_scale=7, unconditional_conditioning=inp["uc"], inpaint=None,
                                x_T=inp["x_T"], img_callback=lambda p0, i: traj.append(p0.clone()))
    assert model._fused is not None, "the entry-point configuration must take the fused route"
    assert _rel(z.numpy(), g["z"]) <= 2e-2
    assert _rel(torch.stack(traj).numpy(), g["pred_x0_traj"]) <= 2e-2
    smp = ddim.DDIMSampler_ReferenceOnly(model)
    smp.make_schedule(int(g["steps"]), ddim_eta=0.0)
    z2, _ = smp.ddim_sampling(inp["c"], tuple(inp["x_T"].shape), x_T=inp["x_T"], unconditional_guidance_scale=7,
                              unconditional_conditioning=inp["uc"], force_generic=True)
    assert _rel(z2.numpy(), z.numpy()) <= 1e-2


def test_product_refuses_cpu_and_missing_extension(monkeypatch):
    """No CPU fallback: a CPU-resident model must raise, and so must a missing libmagicdance_hip.so."""
    import magicdance_amd as M
    from magicdance_amd import _lib, nets
    net = nets.ControlNet(**H.net_kwargs(64, 2), hint_channels=3)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        net(torch.zeros(1, 4, 8, 8), torch.zeros(1, 3, 64, 64), torch.zeros(1), torch.zeros(1, 77, 768))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmagicdance_hip.so")
    with pytest.raises(_lib.MagicDanceHipError, match="no CPU/PyTorch fallback"):
        _lib.load()
    assert M.__version__
