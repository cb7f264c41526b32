"""Shared test plumbing: seeded weights/inputs for a case, oracle runs, golden loading."""
import os

import numpy as np
import torch

from magicdance_amd import nets, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIXES = dict(unet="model.diffusion_model.", app="appearance_control_model.", pose="pose_control_model.")


def net_kwargs(model_channels=320, num_heads=8):
    return dict(image_size=32, in_channels=4, model_channels=model_channels, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=num_heads, use_spatial_transformer=True,
                transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)


def build_nets(model_channels=320, num_heads=8, device="meta", stage1=False):
    kw = net_kwargs(model_channels, num_heads)
    with torch.device(device):
        unet = nets.ControlledUnetModelAttnPose(out_channels=4, **kw)
        app = nets.ControlNetReferenceOnly(out_channels=4, hint_channels=3, **kw)
        if stage1:
            return dict(unet=unet, app1=app)
        pose = nets.ControlNet(hint_channels=3, **kw)
    return dict(unet=unet, app=app, pose=pose)


def synth_weights(model_channels=320, num_heads=8, seed=0, device="cpu", stage1=False):
    """stage1: the appearance net's keys live under ``control_model.`` (models/cldm_v15_reference_only.yaml)."""
    mods = build_nets(model_channels, num_heads, "meta", stage1)
    sd = {}
    for k, m in mods.items():
        sd.update(synthetic.synth_state_dict(m, "control_model." if k == "app1" else PREFIXES[k], seed=seed, device=device))
    return sd


def load_golden(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return {k: z[k] for k in z.files}


def case_inputs(g):
    """Rebuild the exact inputs of a golden case from its recorded seeds (and check the stored copies)."""
    side, frames = int(g["side"]), int(g["frames"])
    inp = synthetic.synth_inputs((side, side), frames=frames, seed=int(g["seed"]))
    rep = lambda x: x.repeat(frames, *([1] * (x.dim() - 1)))
    ref, ctx, x_T, pose = rep(inp["ref"]), rep(inp["ctx"]), rep(inp["x_T"]), inp["pose"]
    assert np.array_equal(x_T.numpy(), g["x_T"]) and np.array_equal(ref.numpy(), g["ref"])
    c = {"c_concat": [pose], "c_crossattn": [ctx], "image_control": [ref], "wonoise": True, "overlap_sampling": False}
    uc = {"c_concat": [pose], "c_crossattn": [ctx], "wonoise": True, "overlap_sampling": False}
    # 'balance' variant (oracle/make_golden.py VARIANT_CASES): the unconditional dict carries the reference too, with
    # another seeded text context so that the guidance term is not a no-op
    ctx_u = synthetic.synth_inputs((side, side), frames=1, seed=7)["ctx"]
    uc_balance = {"c_concat": [pose], "c_crossattn": [ctx_u], "image_control": [ref], "wonoise": True, "overlap_sampling": False}
    return dict(ref=ref, ctx=ctx, x_T=x_T, pose=pose, c=c, uc=uc, uc_balance=uc_balance)


def overlap_case_inputs(g):
    """Inputs of the overlap_sampling fixture (oracle/make_golden.py 'overlap'): 16 pose frames, one reference / text repeated per
    frame, per-frame x_T as stored."""
    side, frames = int(g["side"]), int(g["frames"])
    inp = synthetic.synth_inputs((side, side), frames=frames, seed=int(g["seed"]))
    rep = lambda x: x.repeat(frames, *([1] * (x.dim() - 1)))
    ref, ctx, pose = rep(inp["ref"]), rep(inp["ctx"]), inp["pose"]
    x_T = torch.from_numpy(g["x_T"])
    assert np.array_equal(inp["ref"].numpy(), g["ref"]) and x_T.shape[0] == frames
    c = {"c_concat": [pose], "c_crossattn": [ctx], "image_control": [ref], "wonoise": True, "overlap_sampling": True}
    uc = {"c_concat": [pose], "c_crossattn": [ctx], "wonoise": True, "overlap_sampling": True}
    return dict(ref=ref, ctx=ctx, x_T=x_T, pose=pose, c=c, uc=uc)


def summarize(t):
    flat = t.detach().float().reshape(-1)
    return np.array([flat.mean().item(), flat.std().item(), flat.abs().max().item(), flat.norm().item()], np.float64)


def head_slice(t, n=4):
    t = t.detach().float()
    return (t[:, :n] if t.dim() == 3 else t[:, :, :1, :n]).contiguous().cpu().numpy()


TINY_CLIP = dict(num_hidden_layers=2, num_attention_heads=12, intermediate_size=64)   # ViT-L/14 text geometry (12 heads of 64), 2 thin layers


def build_hip_model(model_channels=320, num_heads=8, seed=0, device="cuda", image_size=64, stage1=False, tiny_clip=False):
    """The product model (magicdance_amd.cldm.ControlLDMReferenceOnlyPose, or the stage-1 ControlLDMReferenceOnly) built
    from the shipped YAML with the golden case's geometry, loaded with the seeded synthetic weights."""
    import magicdance_amd as M
    path = M.DEFAULT_CONFIG.replace("_pose.yaml", ".yaml") if stage1 else M.DEFAULT_CONFIG
    cfg = M.cldm.load_config(path)["model"]
    for blk in (("control_stage_config", "unet_config") if stage1 else
                ("appearance_control_stage_config", "pose_control_stage_config", "unet_config")):
        cfg["params"][blk]["params"].update(model_channels=model_channels, num_heads=num_heads)
    cfg["params"]["first_stage_config"] = "__is_first_stage__"
    cfg["params"]["cond_stage_config"] = "__is_unconditional__"
    cfg["params"]["image_size"] = image_size
    with torch.device("meta"):
        model = M.instantiate_from_config(cfg)
    model = model.to_empty(device=device)
    if tiny_clip:   # the YAML's text-encoder target, built from its embedded config (no network), real (seeded HF-init) weights
        from magicdance_amd import clip
        torch.manual_seed(seed + 11)
        model.cond_stage_model = clip.FrozenCLIPEmbedder(device=str(device), text_config=TINY_CLIP)
    model.register_schedule(timesteps=1000, linear_start=cfg["params"]["linear_start"], linear_end=cfg["params"]["linear_end"])
    model.logvar = torch.zeros(1000)
    sd = synth_weights(model_channels, num_heads, seed=seed, device="cpu", stage1=stage1)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(not k.startswith(("model.", "appearance", "pose", "control_model")) for k in missing), (missing[:5], unexpected[:5])
    return model.to(device).eval()


VAE_PREFIX = "first_stage_model."


def vae_ddconfig(ch=128):
    return dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=ch, ch_mult=[1, 2, 4, 4],
                num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def build_vae(ch=128, device="meta"):
    from magicdance_amd import autoencoder
    with torch.device(device):
        return autoencoder.AutoencoderKL(ddconfig=vae_ddconfig(ch), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)


def synth_vae_weights(ch=128, seed=0, device="cpu"):
    """Seeded first-stage weights keyed by the reference's state-dict names (``first_stage_model.*``)."""
    return synthetic.synth_state_dict(build_vae(ch, "meta"), VAE_PREFIX, seed=seed, device=device)


def build_hip_vae(ch=128, seed=0, device="cuda"):
    vae = build_vae(ch, "meta").to_empty(device="cpu")
    sd = synth_vae_weights(ch, seed)
    vae.load_state_dict({k[len(VAE_PREFIX):]: v for k, v in sd.items()}, strict=True)
    return vae.to(device).eval()
