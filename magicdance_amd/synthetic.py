"""Seeded synthetic weights and inputs: there is no network for checkpoints, and the
reference's default init zeroes every output projection (zero_module), which would make parity
vacuous.  Values are a deterministic function of (seed, key name, shape) only, so the reference model
(in the build container), the CPU oracle and the HIP engine can all be given bit-identical fp32
weights without shipping a checkpoint.
"""
import zlib

import torch

_NORM_TAGS = ("in_layers.0.", "out_layers.0.", ".norm.", ".norm1.", ".norm2.", ".norm3.", "out.0.", ".norm_out.")
# layers the reference zero-initialises (openaimodel.py:249-251,749; attention.py:357; cldm.py:614,733):
_ZERO_INIT_TAGS = ("out_layers.3.", ".proj_out.", "zero_convs.", "middle_block_out.", "out.2.",
                   "input_hint_block.14.")


def _gen(seed, key, device):
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, seed=0, device="cpu", zero_init_gain=0.5):
    g = _gen(seed, key, device)
    is_norm = any(t in key for t in _NORM_TAGS) and len(shape) == 1
    if is_norm and key.endswith("weight"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
    if len(shape) == 1:  # biases
        return 0.05 * torch.randn(shape, generator=g, device=device)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    bound = fan_in ** -0.5
    gain = zero_init_gain if any(t in key for t in _ZERO_INIT_TAGS) else 1.0
    return (torch.rand(shape, generator=g, device=device) * 2 - 1) * (bound * gain * 3 ** 0.5)


def synth_state_dict(module, prefix="", seed=0, device="cpu"):
    """Synthetic fp32 state dict for ``module`` (keys get ``prefix``; generation keyed by the full name)."""
    out = {}
    for k, v in module.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        out[prefix + k] = synth_tensor(prefix + k, tuple(v.shape), seed, device).to(torch.float32)
    return out


def synth_inputs(latent_hw=(64, 64), frames=1, seed=0, ctx_tokens=77, ctx_dim=768, device="cpu"):
    """ref latent (stand-in for VAE(ref)*0.18215), ctx (stand-in for CLIP("")), pose maps in [0,1] and
    ONE x_T shared by all frames (test_any_image_pose.py:201-202)."""
    h, w = latent_hw
    g = lambda s: torch.Generator(device="cpu").manual_seed(seed * 1000 + s)
    ref = torch.randn(1, 4, h, w, generator=g(2))
    ctx = torch.randn(1, ctx_tokens, ctx_dim, generator=g(3))
    pose = torch.rand(frames, 3, 8 * h, 8 * w, generator=g(4))
    x_T = torch.randn(1, 4, h, w, generator=g(5))
    return dict(ref=ref.to(device), ctx=ctx.to(device), pose=pose.to(device), x_T=x_T.to(device))


def synth_vae_inputs(side, batch=1, seed=0, device="cpu"):
    """First-stage inputs: a latent z [B,4,side,side] (unit normal = latent / scale_factor scale) for decode and an
    image [B,3,8*side,8*side] in [-1,1] for encode."""
    g = lambda s: torch.Generator(device="cpu").manual_seed(seed * 1000 + s)
    z = torch.randn(batch, 4, side, side, generator=g(11))
    img = torch.rand(batch, 3, 8 * side, 8 * side, generator=g(12)) * 2 - 1
    return z.to(device), img.to(device)
