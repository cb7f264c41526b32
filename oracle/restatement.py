"""TEST INFRASTRUCTURE ONLY.  CPU (torch fp32) restatement of the reference's diffusion-sampling hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file, and only as
the *checker* / reported baseline.  The product path (magicdance_amd) never imports it.

Pinning: the reference ships no tests, golden vectors or known-answer fixtures for this path
 -- "parity unpinned by the reference's own tests".  This restatement is therefore
pinned against *outputs of the reference itself*: oracle/make_golden.py imports the unmodified reference
modules (oracle/ref_shim.py) in the build container, runs them on seeded inputs and commits the vectors
under tests/golden/;  tests/test_oracle_golden.py checks this file against those vectors, and (when
/root/reference is present) tests/test_oracle_vs_reference.py checks it against the live reference.

Everything here is a plain functional walk over a flat ``state_dict`` with the reference's key names.
Each function cites the reference lines it follows (paths relative to
/root/reference/model_lib/ControlNet/).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- schedules
def make_beta_schedule_linear(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """ldm/modules/diffusionmodules/util.py:20-24 ("linear")."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def alphas_cumprod(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """ldm/models/diffusion/ddpm.py:138-157 (register_schedule): float64 cumprod then float32."""
    betas = make_beta_schedule_linear(n_timestep, linear_start, linear_end)
    ac = np.cumprod(1.0 - betas, axis=0)
    return torch.tensor(ac, dtype=torch.float32)


def make_ddim_timesteps(num_ddim, num_ddpm=1000):
    """util.py:45-59 ('uniform')."""
    c = num_ddpm // num_ddim
    return np.asarray(list(range(0, num_ddpm, c))) + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """util.py:62-73.  ``alphacums`` is the fp32 CPU tensor, as passed at ddim.py:377."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def timestep_embedding(timesteps, dim, max_period=10000):
    """util.py:189-209."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ----------------------------------------------------------------------------- blocks
class Cfg:
    """Geometry of one network (the YAML ``params`` of cldm_v15_reference_only_pose.yaml:22-71)."""

    def __init__(self, model_channels=320, channel_mult=(1, 2, 4, 4), num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), num_heads=8, in_channels=4, out_channels=4, **_):
        self.mc, self.mult, self.nrb = model_channels, tuple(channel_mult), num_res_blocks
        self.attn_res, self.heads = tuple(attention_resolutions), num_heads
        self.in_channels, self.out_channels = in_channels, out_channels


def _p(sd, key):
    return sd[key].float()


def time_embed(sd, pre, cfg, t):
    """cldm.py:66-67 / 471-472 / 737-738: sinusoid -> Linear -> SiLU -> Linear."""
    e = timestep_embedding(t, cfg.mc)
    e = F.linear(e, _p(sd, pre + "time_embed.0.weight"), _p(sd, pre + "time_embed.0.bias"))
    return F.linear(F.silu(e), _p(sd, pre + "time_embed.2.weight"), _p(sd, pre + "time_embed.2.bias"))


def resblock(sd, pre, x, emb):
    """openaimodel.py:275-295 (no up/down, no scale-shift): GN32(eps1e-5)+SiLU+conv, +Linear(SiLU(emb)),
    GN32+SiLU+conv, + skip (identity or 1x1)."""
    h = F.group_norm(x, 32, _p(sd, pre + "in_layers.0.weight"), _p(sd, pre + "in_layers.0.bias"), eps=1e-5)
    h = F.conv2d(F.silu(h), _p(sd, pre + "in_layers.2.weight"), _p(sd, pre + "in_layers.2.bias"), padding=1)
    e = F.linear(F.silu(emb), _p(sd, pre + "emb_layers.1.weight"), _p(sd, pre + "emb_layers.1.bias"))
    h = h + e[:, :, None, None]
    h = F.group_norm(h, 32, _p(sd, pre + "out_layers.0.weight"), _p(sd, pre + "out_layers.0.bias"), eps=1e-5)
    h = F.conv2d(F.silu(h), _p(sd, pre + "out_layers.3.weight"), _p(sd, pre + "out_layers.3.bias"), padding=1)
    if pre + "skip_connection.weight" in sd:
        x = F.conv2d(x, _p(sd, pre + "skip_connection.weight"), _p(sd, pre + "skip_connection.bias"))
    return x + h


def attention(sd, pre, x, context, heads):
    """attention.py:168-199 (vanilla CrossAttention._forward): q/k/v Linear (no bias), per-head
    softmax(q k^T * d^-1/2) v, to_out Linear+bias."""
    q = F.linear(x, _p(sd, pre + "to_q.weight"))
    k = F.linear(context, _p(sd, pre + "to_k.weight"))
    v = F.linear(context, _p(sd, pre + "to_v.weight"))
    b, n, c = q.shape
    d = c // heads
    sp = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (d ** -0.5)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, c)
    return F.linear(out, _p(sd, pre + "to_out.0.weight"), _p(sd, pre + "to_out.0.bias"))


def transformer_block(sd, pre, x, context, heads, mode, banks, attn_index, uc):
    """attention.py:278-320 (BasicTransformerBlock.forward) incl. the bank write / read."""
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), _p(sd, pre + n + ".weight"), _p(sd, pre + n + ".bias"))
    xn = ln(x, "norm1")
    if uc or mode is None:
        x = attention(sd, pre + "attn1.", xn, xn, heads) + x                       # :280-281
    elif mode == "write":
        banks.append([xn])                                                          # :287-292
        x = attention(sd, pre + "attn1.", xn, xn, heads) + x                       # :296-298
    elif mode == "read":
        bank = banks[attn_index]                                                    # :303
        ctx = torch.cat([xn] + bank, dim=1) if len(bank) > 0 else xn              # :305-311
        x = attention(sd, pre + "attn1.", xn, ctx, heads) + x
    else:
        raise NotImplementedError
    x = attention(sd, pre + "attn2.", ln(x, "norm2"), context, heads) + x          # :318
    h = F.linear(ln(x, "norm3"), _p(sd, pre + "ff.net.0.proj.weight"), _p(sd, pre + "ff.net.0.proj.bias"))
    a, gate = h.chunk(2, dim=-1)                                                    # GEGLU :55-57
    h = F.linear(a * F.gelu(gate), _p(sd, pre + "ff.net.2.weight"), _p(sd, pre + "ff.net.2.bias"))
    return h + x                                                                    # :319


def spatial_transformer(sd, pre, x, context, heads, mode=None, banks=None, attn_index=None, uc=False):
    """attention.py:366-385: GN(eps 1e-6) -> 1x1 -> tokens -> blocks -> image -> 1x1 -> + x_in."""
    b, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, _p(sd, pre + "norm.weight"), _p(sd, pre + "norm.bias"), eps=1e-6)
    x = F.conv2d(x, _p(sd, pre + "proj_in.weight"), _p(sd, pre + "proj_in.bias"))
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    i = 0
    while pre + f"transformer_blocks.{i}.norm1.weight" in sd:
        x = transformer_block(sd, pre + f"transformer_blocks.{i}.", x, context, heads, mode, banks, attn_index, uc)
        i += 1
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = F.conv2d(x, _p(sd, pre + "proj_out.weight"), _p(sd, pre + "proj_out.bias"))
    return x + x_in


def _block(sd, pre, h, emb, ctx, heads, mode, banks, attn_index, uc):
    """openaimodel.py:79-108 (TimestepEmbedSequential.forward): dispatch on the layer kind, thread the
    read index.  Layer kinds are recognised from the keys present under ``pre``."""
    j = 0
    while True:
        lp = f"{pre}{j}."
        if lp + "in_layers.0.weight" in sd:
            h = resblock(sd, lp, h, emb)
        elif lp + "proj_in.weight" in sd:
            if uc:
                h = spatial_transformer(sd, lp, h, ctx, heads, uc=True)
            else:
                h = spatial_transformer(sd, lp, h, ctx, heads, mode, banks, attn_index)
                if mode == "read":
                    attn_index += 1
        elif lp + "op.weight" in sd:                                                # Downsample :178-180
            h = F.conv2d(h, _p(sd, lp + "op.weight"), _p(sd, lp + "op.bias"), stride=2, padding=1)
        elif lp + "conv.weight" in sd:                                              # Upsample :129-139
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, _p(sd, lp + "conv.weight"), _p(sd, lp + "conv.bias"), padding=1)
        elif lp + "weight" in sd and sd[lp + "weight"].dim() == 4:                  # bare conv (stem)
            h = F.conv2d(h, _p(sd, lp + "weight"), _p(sd, lp + "bias"), padding=1)
        else:
            break
        j += 1
    return h, attn_index


def _n_blocks(sd, pre):
    i = 0
    while any(k.startswith(f"{pre}{i}.") for k in sd):
        i += 1
    return i


def appearance_forward(sd, pre, cfg, x, t, ctx):
    """cldm.py:469-497 (ControlNetReferenceOnly.forward, 'write'): returns the bank (16 x [norm1(x)])."""
    emb = time_embed(sd, pre, cfg, t)
    banks, hs, h = [], [], x.float()
    for i in range(_n_blocks(sd, pre + "input_blocks.")):
        h, _ = _block(sd, f"{pre}input_blocks.{i}.", h, emb, ctx, cfg.heads, "write", banks, None, False)
        hs.append(h)
    h, _ = _block(sd, pre + "middle_block.", h, emb, ctx, cfg.heads, "write", banks, None, False)
    for i in range(_n_blocks(sd, pre + "output_blocks.")):
        h = torch.cat([h, hs.pop()], dim=1)
        h, _ = _block(sd, f"{pre}output_blocks.{i}.", h, emb, ctx, cfg.heads, "write", banks, None, False)
    return banks


def hint_encoder(sd, pre, hint):
    """cldm.py:599-615: 8 convs (strides 1,1,2,1,2,1,2,1) with SiLU between, none after the last."""
    h = hint.float()
    strides = [1, 1, 2, 1, 2, 1, 2, 1]
    for i, s in enumerate(strides):
        h = F.conv2d(h, _p(sd, f"{pre}input_hint_block.{2 * i}.weight"), _p(sd, f"{pre}input_hint_block.{2 * i}.bias"),
                     stride=s, padding=1)
        if i != len(strides) - 1:
            h = F.silu(h)
    return h


def pose_forward(sd, pre, cfg, x, hint, t, ctx):
    """cldm.py:736-757 (ControlNet.forward): 13 zero-conv outputs."""
    emb = time_embed(sd, pre, cfg, t)
    guided = hint_encoder(sd, pre, hint)
    outs, h = [], x.float()
    for i in range(_n_blocks(sd, pre + "input_blocks.")):
        h, _ = _block(sd, f"{pre}input_blocks.{i}.", h, emb, ctx, cfg.heads, None, None, None, False)
        if guided is not None:
            h = h + guided
            guided = None
        outs.append(F.conv2d(h, _p(sd, f"{pre}zero_convs.{i}.0.weight"), _p(sd, f"{pre}zero_convs.{i}.0.bias")))
    h, _ = _block(sd, pre + "middle_block.", h, emb, ctx, cfg.heads, None, None, None, False)
    outs.append(F.conv2d(h, _p(sd, pre + "middle_block_out.0.weight"), _p(sd, pre + "middle_block_out.0.bias")))
    return outs


def unet_forward(sd, pre, cfg, x, t, ctx, banks=None, pose=None, uc=False, only_mid_control=False):
    """cldm.py:59-112 (ControlledUnetModelAttnPose.forward): uc branch :70-84, read branch :86-107."""
    emb = time_embed(sd, pre, cfg, t)
    hs, h, idx = [], x.float(), 0
    mode = None if uc else "read"
    pose = None if pose is None else list(pose)
    for i in range(_n_blocks(sd, pre + "input_blocks.")):
        h, idx = _block(sd, f"{pre}input_blocks.{i}.", h, emb, ctx, cfg.heads, mode, banks, idx, uc)
        hs.append(h)
    h, idx = _block(sd, pre + "middle_block.", h, emb, ctx, cfg.heads, mode, banks, idx, uc)
    if not uc and pose is not None:
        h = h + pose.pop()                                                          # :93-95
    for i in range(_n_blocks(sd, pre + "output_blocks.")):
        if uc or only_mid_control or banks is None or pose is None:
            skip = hs.pop()                                                         # :78-84, :98-100, :105-106
        else:
            skip = hs.pop() + pose.pop()                                            # :102-104
        h = torch.cat([h, skip], dim=1)
        if not uc and (only_mid_control or banks is None):
            h, _ = _block(sd, f"{pre}output_blocks.{i}.", h, emb, ctx, cfg.heads, None, None, None, False)
        else:
            h, idx = _block(sd, f"{pre}output_blocks.{i}.", h, emb, ctx, cfg.heads, mode, banks, idx, uc)
    h = F.group_norm(h, 32, _p(sd, pre + "out.0.weight"), _p(sd, pre + "out.0.bias"), eps=1e-5)
    return F.conv2d(F.silu(h), _p(sd, pre + "out.2.weight"), _p(sd, pre + "out.2.bias"), padding=1)


UNET, APP, POSE = "model.diffusion_model.", "appearance_control_model.", "pose_control_model."


STAGE1_APP = "control_model."


def apply_model(sd, cfg, x_noisy, t, cond, reference_image_noisy, uc=False, only_mid_control=False, stage1=False):
    """cldm.py:1099-1117 (ControlLDMReferenceOnlyPose.apply_model); ``stage1``: cldm.py:1066-1077
    (ControlLDMReferenceOnly.apply_model -- appearance net under ``control_model.``, no pose ControlNet, c_concat unused)."""
    ctx = torch.cat(cond["c_crossattn"], 1).float()
    if stage1:
        banks = appearance_forward(sd, STAGE1_APP, cfg, reference_image_noisy, t, ctx) if reference_image_noisy is not None else []
        return unet_forward(sd, UNET, cfg, x_noisy, t, ctx, banks, None, uc, only_mid_control)
    ctx_void = torch.cat(cond["c_crossattn_void"], 1).float() if cond.get("c_crossattn_void") is not None else ctx
    banks = []
    if reference_image_noisy is not None:
        banks = appearance_forward(sd, APP, cfg, reference_image_noisy, t, ctx_void)
    pose = None
    if cond.get("c_concat") is not None:
        pose = pose_forward(sd, POSE, cfg, x_noisy, torch.cat(cond["c_concat"], 1), t, ctx_void)
    return unet_forward(sd, UNET, cfg, x_noisy, t, ctx, banks, pose, uc, only_mid_control)


def q_sample(ac, x0, t, noise):
    """ddpm.py:356-359."""
    a = ac[t].sqrt()[:, None, None, None]
    b = (1 - ac[t]).sqrt()[:, None, None, None]
    return a * x0 + b * noise


def overlap_windows(num_frames, offset, window=16, stride=12):
    """ddim.py:576-577: frame-index windows of the temporal overlap sampling, ``indices = arange(start, start + 16) % F`` for
    ``start in range(offset, offset + F - 16 + 1 + 12, 12)``."""
    return [torch.arange(s0, s0 + window) % num_frames for s0 in range(offset, offset + num_frames - window + 1 + stride, stride)]


def ddim_sample(sd, cfg, cond, uncond, x_T, steps=50, eta=0.0, scale=7.0, record=None, stage1=False, q_noises=None,
                offsets=None):
    """ddim.py:391-516 + 519-645 ('controlnet is more important' branch :595-605, or the 'balance' 2B-batched branch
    :540-567 when the unconditional dict carries ``image_control`` too; eps parameterisation,
    eta = 0 so sigma_t = 0 and the noise term vanishes; the reference still draws it, :641).
    ``record(i, dict)`` receives eps_t / eps_uc / x_prev per step for seam-level parity."""
    ac = alphas_cumprod()
    ts = make_ddim_timesteps(steps)
    sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac, ts, eta)
    sqrt_1m = np.sqrt(1.0 - alphas)
    img = x_T.float()
    b = img.shape[0]
    ref0 = torch.cat(cond["image_control"], 1).float()
    for i, step in enumerate(np.flip(ts)):
        index = len(ts) - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if cond["wonoise"]:
            ref = ref0
        else:   # :529-535; ``q_noises[i]`` = the randn_like draw of step i (fixtures record it), else drawn here
            ref = q_sample(ac, ref0, t, torch.randn_like(ref0) if q_noises is None else q_noises[i])
        if uncond is None or scale == 1.0:
            e_t = apply_model(sd, cfg, img, t, cond, ref, stage1=stage1)                    # :537-538
            e_c = e_u = e_t
        elif uncond.get("image_control") is not None:                                       # balance :540-567
            c_in = {k: ([torch.cat([uncond[k][j], cond[k][j]]) for j in range(len(cond[k]))] if isinstance(cond[k], list)
                        else cond[k]) for k in cond}
            e_u, e_c = apply_model(sd, cfg, torch.cat([img] * 2), torch.cat([t] * 2), c_in, torch.cat([ref] * 2),
                                   stage1=stage1).chunk(2)
            e_t = e_u + scale * (e_c - e_u)
        elif cond.get("overlap_sampling"):                                                  # :569-594
            # windows of 16 frames, stride 12, from a random offset (python ``random``, :575 -- passed in as offsets[i]);
            # per-window CFG, accumulated and divided by the per-frame visit count
            nf = cond["c_concat"][0].shape[0]
            import random
            off = random.randint(0, nf - 1) if offsets is None else int(offsets[i])
            pred_all, counts = torch.zeros_like(img), torch.zeros(nf)
            for idx in overlap_windows(nf, off):
                c_w = dict(cond)
                c_w["c_concat"] = [cond["c_concat"][0][idx]]
                m_t = apply_model(sd, cfg, img[idx], t, c_w, ref, stage1=stage1)
                m_u = apply_model(sd, cfg, img[idx], t, c_w, None, uc=True, stage1=stage1)
                pred_all[idx] += m_u + scale * (m_t - m_u)
                counts[idx] += 1
            e_t = pred_all / counts.reshape(-1, 1, 1, 1)
            e_c = e_u = e_t
        else:
            e_c = apply_model(sd, cfg, img, t, cond, ref, stage1=stage1)                    # :603
            e_u = apply_model(sd, cfg, img, t, cond, None, uc=True, stage1=stage1)          # :604
            e_t = e_u + scale * (e_c - e_u)                                                 # :605
        a_t = torch.full((b, 1, 1, 1), float(alphas[index]))
        a_prev = torch.full((b, 1, 1, 1), float(alphas_prev[index]))
        sigma_t = torch.full((b, 1, 1, 1), float(sigmas[index]))
        s1m = torch.full((b, 1, 1, 1), float(sqrt_1m[index]))
        pred_x0 = (img - s1m * e_t) / a_t.sqrt()                                            # :624
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t                                 # :640
        img = a_prev.sqrt() * pred_x0 + dir_xt                                              # :644 (sigma = 0)
        if record is not None:
            record(i, dict(eps_c=e_c, eps_u=e_u, x_prev=img, pred_x0=pred_x0, t=int(step)))
    return img
