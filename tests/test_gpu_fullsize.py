"""GPU parity at the BASELINE.json configurations (full SD-1.5 geometry, latent 64x64 = 512x512) against goldens produced by
the UNMODIFIED reference on CPU fp32 (oracle/make_golden.py TRAJ_CASES: c0_b1_s20 = configs[0], c1_b1_s50 = configs[1],
c2_b8_s2 = configs[2] geometry).  Three kinds of check per case:
  * one apply_model pair (eps_cond / eps_uncond) at the probe timestep;
  * the per-step guided eps with the REFERENCE's x_t fed in at every step (error of one evaluation, no recurrence);
  * the free-running DDIM trajectory from x_T (error including its growth through the recurrence).
Errors are relative to max|reference tensor|; the per-step curves are written to gpurun_out/parity_fullsize.log (copied to
profiles/).  fp16 operands bound one eps evaluation at ~1.2e-3 (weights ~0.85e-3, activations ~0.9e-3 in
quadrature); the asserts sit at <= 2x the measured values."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

# relative tolerances = 2x the values measured on MI355X (profiles/round2_parity_fullsize.txt: eps pair <= 1.27e-3, bank / pose
# seams <= 4.6e-4, guided eps per step <= 2.47e-3, final latent 7.97e-4 / 8.60e-4 / 1.44e-3)
TOL_EPS = 2.6e-3        # one apply_model
TOL_SEAM = 1e-3         # bank / pose tensors (norm and head slice)
TOL_GUIDED = 5e-3       # guided eps e_u + 7 (e_c - e_u) on the reference's x_t: CFG 7 combines two evaluations
TOL_TRAJ = {"c0_b1_s20": 1.8e-3, "c1_b1_s50": 1.6e-3, "c2_b8_s2": 3e-3}   # final latent of the free-running loop

_LOG = []


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    m = H.build_hip_model(320, 8, seed=0, device=dev, image_size=64)
    yield m
    os.makedirs(os.path.join(H.ROOT, "gpurun_out"), exist_ok=True)
    tag = "" if os.environ.get("MD_RES_LO", "1") != "0" else "_single_term_stream"
    with open(os.path.join(H.ROOT, "gpurun_out", f"parity_fullsize{tag}.log"), "a") as f:
        f.write(f"# MD_RES_LO={os.environ.get('MD_RES_LO', '1')} (1 = two-term fp16 residual stream)\n" + "\n".join(_LOG) + "\n")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-9, np.abs(b).max()))


def _schedule(steps):
    """DDIM coefficients of the checker (oracle/restatement.py: ddim.py:359-388, util.py:45-73)."""
    from oracle import restatement as R
    ts = R.make_ddim_timesteps(steps)
    _, a, ap = R.make_ddim_sampling_parameters(R.alphas_cumprod(), ts, 0.0)
    return np.flip(ts).copy(), np.flip(np.asarray(a, np.float64)).copy(), np.flip(np.asarray(ap, np.float64)).copy()


def _guided_eps_from_traj(x_i, x_next, a_t, a_prev):
    """Invert the eta = 0 update (ddim.py:622-644) for the guided eps the reference used between two stored x_t."""
    den = np.sqrt(1.0 - a_prev) - np.sqrt(a_prev) * np.sqrt(1.0 - a_t) / np.sqrt(a_t)
    return (x_next.astype(np.float64) - np.sqrt(a_prev / a_t) * x_i.astype(np.float64)) / den


def _case(g, dev):
    inp = H.case_inputs(dict(g, x_T=np.repeat(g["x_T"], int(g["frames"]), 0), ref=np.repeat(g["ref"], int(g["frames"]), 0)))
    mv = lambda d: {k: ([t.to(dev) for t in v] if isinstance(v, list) else v) for k, v in d.items()}  # noqa: E731
    return inp, mv(inp["c"]), mv(inp["uc"])


@pytest.mark.parametrize("name", ["c1_b1_s50", "c0_b1_s20", "c2_b8_s2"])
def test_baseline_config_matches_reference(dev, model, name):
    g = H.load_golden(name)
    frames, steps = int(g["frames"]), int(g["steps"])
    inp, c, uc = _case(g, dev)
    x_T, ref = inp["x_T"].to(dev), inp["ref"].to(dev)
    t = torch.full((frames,), int(g["t_probe"]), dtype=torch.long, device=dev)
    # (1) one apply_model pair
    e_c = model.apply_model(x_T, t, c, ref).cpu().numpy()
    e_u = model.apply_model(x_T, t, c, None, uc=True).cpu().numpy()
    rc, ru = _rel(e_c, g["eps_c"]), _rel(e_u, g["eps_u"])
    _LOG.append(f"{name}: eps_cond rel {rc:.3e}  eps_uncond rel {ru:.3e}  (t = {int(g['t_probe'])}, B = {frames})")
    assert rc <= TOL_EPS and ru <= TOL_EPS, (rc, ru)
    if frames == 1:   # bank / pose seams of the same probe (head slices + norms, as stored)
        bank = []
        model.appearance_control_model(x=ref, hint=None, timesteps=t, context=inp["ctx"].to(dev), attention_bank=bank,
                                       attention_mode="write", uc=False)
        worst = 0.0
        for i, bk in enumerate(bank):
            s, gs = H.summarize(bk[0]), g[f"bank{i}_sum"]
            worst = max(worst, abs(s[3] - gs[3]) / gs[3], _rel(H.head_slice(bk[0]), g[f"bank{i}_head"]) / 4)
        pr = model.pose_control_model(x=x_T, hint=inp["pose"].to(dev), timesteps=t, context=inp["ctx"].to(dev))
        for i, p in enumerate(pr):
            gs = g[f"pose{i}_sum"]
            worst = max(worst, abs(H.summarize(p)[3] - gs[3]) / gs[3], float(np.abs(H.head_slice(p) - g[f"pose{i}_head"]).max()) / gs[2] / 4)
        _LOG.append(f"{name}: worst bank / pose seam (norm and head slice) rel {worst:.3e}")
        assert worst <= TOL_SEAM
    # (2) per-step guided eps on the reference's own x_t (single evaluation, no recurrence)
    ts, a, ap = _schedule(steps)
    xt = g["x_traj"]                                           # [steps + 1, frames, 4, 64, 64]
    assert xt.shape[0] == steps + 1 and np.array_equal(xt[0], np.repeat(g["x_T"], frames, 0))
    curve = []
    for i in range(steps):
        want = _guided_eps_from_traj(xt[i], xt[i + 1], a[i], ap[i])
        xi = torch.from_numpy(xt[i]).to(dev)
        ti = torch.full((frames,), int(ts[i]), dtype=torch.long, device=dev)
        ec = model.apply_model(xi, ti, c, ref)
        eu = model.apply_model(xi, ti, c, None, uc=True)
        got = (eu + 7.0 * (ec - eu)).cpu().numpy()
        curve.append(_rel(got, want))
    _LOG.append(f"{name}: guided eps on the reference x_t, per step: " + " ".join(f"{v:.2e}" for v in curve))
    _LOG.append(f"{name}: guided eps per step: max {max(curve):.3e} mean {np.mean(curve):.3e}")
    assert max(curve) <= TOL_GUIDED, max(curve)
    # (3) free-running trajectory through sample_log (fused HIP-graph route)
    z, inter = model.sample_log(cond=c, batch_size=frames, ddim=True, ddim_steps=steps, eta=0.0, unconditional_guidance_scale=7,
                                unconditional_conditioning=uc, inpaint=None, x_T=x_T, log_every_t=1)
    assert model._fused is not None
    xs = torch.stack(inter["x_inter"]).cpu().numpy()
    assert xs.shape == xt.shape
    growth = [_rel(xs[i], xt[i]) for i in range(1, steps + 1)]
    _LOG.append(f"{name}: free-running x_t vs reference, per step: " + " ".join(f"{v:.2e}" for v in growth))
    rz = _rel(z.cpu().numpy(), g["z"])
    _LOG.append(f"{name}: final latent after {steps} steps rel {rz:.3e} (max-abs err {np.abs(z.cpu().numpy() - g['z']).max():.3e}, "
                f"max|z| {np.abs(g['z']).max():.3e})")
    assert rz <= TOL_TRAJ[name], rz
