"""Parameter containers + topology for the three networks on the hot path.

These classes mirror the *constructor kwargs* and the *state-dict key layout* of the reference
networks so that a reference checkpoint (``model_state-*.th``) loads with ``strict=True`` and the
reference YAML instantiates them by swapping only the ``target:`` strings:

  * ``ControlledUnetModelAttnPose``  <- model_lib/ControlNet/cldm/cldm.py:59-112 (on
    ldm/modules/diffusionmodules/openaimodel.py:462-756 ``UNetModel.__init__``)
  * ``ControlNetReferenceOnly``      <- cldm.py:164-497 (appearance net, "write")
  * ``ControlNet``                   <- cldm.py:500-757 (pose ControlNet, 13 zero-convs)

They hold fp32 master parameters only.  No arithmetic lives here: ``forward`` of every network
hands over to the HIP engine (``magicdance_amd.engine``), which fails loudly when the gfx950
extension is missing.  The leaf containers are stock ``torch.nn`` parameter holders placed at the
same ``nn.Sequential`` indices as the reference so the key names come out identical.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn


# ----------------------------------------------------------------------------- leaf containers
class ResBlock(nn.Module):
    """Keys: in_layers.{0,2}, emb_layers.1, out_layers.{0,3}, skip_connection (openaimodel.py:183-261)."""

    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(),
                                       nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(0.0),
                                        nn.Conv2d(out_channels, out_channels, 3, padding=1))
        self.skip_connection = (nn.Identity() if out_channels == channels
                                else nn.Conv2d(channels, out_channels, 1))


class _Attn(nn.Module):
    """to_q/to_k/to_v (no bias) + to_out.0 (attention.py:146-163)."""

    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class _FeedForward(nn.Module):
    """ff.net.0.proj (GEGLU), ff.net.2 (attention.py:50-77)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(_GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))


class BasicTransformerBlock(nn.Module):
    """attn1, ff, attn2, norm1..3 (attention.py:253-273)."""

    def __init__(self, dim, n_heads, d_head, context_dim):
        super().__init__()
        self.attn1 = _Attn(dim, dim, n_heads, d_head)
        self.ff = _FeedForward(dim)
        self.attn2 = _Attn(dim, context_dim, n_heads, d_head)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)


class SpatialTransformer(nn.Module):
    """norm (GroupNorm eps 1e-6), proj_in/proj_out 1x1, transformer_blocks (attention.py:323-364)."""

    def __init__(self, in_channels, n_heads, d_head, depth, context_dim):
        super().__init__()
        inner = n_heads * d_head
        self.in_channels, self.n_heads, self.d_head = in_channels, n_heads, d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)


class Downsample(nn.Module):
    """op = conv3x3 stride 2 (openaimodel.py:154-180)."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.op = nn.Conv2d(channels, channels, 3, stride=2, padding=1)


class Upsample(nn.Module):
    """nearest x2 then conv3x3 (openaimodel.py:111-139)."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


# ----------------------------------------------------------------------------- topology
@dataclass
class NetConfig:
    in_channels: int = 4
    model_channels: int = 320
    out_channels: int = 4
    hint_channels: int = 3
    num_res_blocks: int = 2
    attention_resolutions: tuple = (4, 2, 1)
    channel_mult: tuple = (1, 2, 4, 4)
    num_heads: int = 8
    num_head_channels: int = -1
    transformer_depth: int = 1
    context_dim: int = 768

    @property
    def time_embed_dim(self):
        return self.model_channels * 4


_UNSUPPORTED = dict(dims=2, use_scale_shift_norm=False, resblock_updown=False, conv_resample=True,
                    use_new_attention_order=False, num_classes=None, n_embed=None,
                    disable_self_attentions=None, num_attention_blocks=None,
                    disable_middle_self_attn=False, use_linear_in_transformer=False, dropout=0)


def _net_config(kw):
    """Accept the reference ctor kwargs (openaimodel.py:462-492 / cldm.py:165-195, 501-530); refuse the
    variants the pose config never reaches instead of silently mis-computing them."""
    kw = dict(kw)
    for name, ok in _UNSUPPORTED.items():
        if name in kw and kw[name] != ok and not (name == "dropout" and kw[name] in (0, 0.0)):
            raise NotImplementedError(f"{name}={kw[name]!r} is outside the pose-config hot path")
    if not kw.get("use_spatial_transformer", False):
        raise NotImplementedError("only the SpatialTransformer attention path is on the hot path")
    if kw.get("use_fp16", False):
        raise NotImplementedError("use_fp16=True (convert_to_fp16) is not part of the pose config")
    ctx = kw.get("context_dim")
    if isinstance(ctx, (list, tuple)):
        ctx = list(ctx)
        assert len(set(ctx)) == 1
        ctx = ctx[0]
    nrb = kw["num_res_blocks"]
    if not isinstance(nrb, int):
        assert len(set(nrb)) == 1
        nrb = nrb[0]
    nh, nhc = kw.get("num_heads", -1), kw.get("num_head_channels", -1)
    if kw.get("num_heads_upsample", -1) not in (-1, nh):
        raise NotImplementedError("num_heads_upsample != num_heads")
    assert nh != -1 or nhc != -1, "Either num_heads or num_head_channels has to be set"
    return NetConfig(in_channels=kw["in_channels"], model_channels=kw["model_channels"],
                     out_channels=kw.get("out_channels", 4), hint_channels=kw.get("hint_channels", 3),
                     num_res_blocks=nrb, attention_resolutions=tuple(kw["attention_resolutions"]),
                     channel_mult=tuple(kw.get("channel_mult", (1, 2, 4, 8))), num_heads=nh,
                     num_head_channels=nhc, transformer_depth=kw.get("transformer_depth", 1),
                     context_dim=ctx)


def _heads(cfg, ch):
    if cfg.num_head_channels == -1:
        return cfg.num_heads, ch // cfg.num_heads
    return ch // cfg.num_head_channels, cfg.num_head_channels


class _Seq(nn.Sequential):
    """Stand-in for TimestepEmbedSequential (openaimodel.py:72-108): container only."""


def _make_encoder(cfg):
    """input_blocks + middle_block shared by all three nets (openaimodel.py:552-660)."""
    mc, ted = cfg.model_channels, cfg.time_embed_dim
    blocks = nn.ModuleList([_Seq(nn.Conv2d(cfg.in_channels, mc, 3, padding=1))])
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [ResBlock(ch, ted, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                nh, dh = _heads(cfg, ch)
                layers.append(SpatialTransformer(ch, nh, dh, cfg.transformer_depth, cfg.context_dim))
            blocks.append(_Seq(*layers))
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            blocks.append(_Seq(Downsample(ch)))
            chans.append(ch)
            ds *= 2
    nh, dh = _heads(cfg, ch)
    middle = _Seq(ResBlock(ch, ted, ch),
                  SpatialTransformer(ch, nh, dh, cfg.transformer_depth, cfg.context_dim),
                  ResBlock(ch, ted, ch))
    return blocks, middle, chans, ch, ds


def _make_decoder(cfg, chans, ch, ds):
    """output_blocks (openaimodel.py:662-744)."""
    mc, ted = cfg.model_channels, cfg.time_embed_dim
    chans = list(chans)
    blocks = nn.ModuleList()
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [ResBlock(ch + ich, ted, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                nh, dh = _heads(cfg, ch)
                layers.append(SpatialTransformer(ch, nh, dh, cfg.transformer_depth, cfg.context_dim))
            if level and i == cfg.num_res_blocks:
                layers.append(Upsample(ch))
                ds //= 2
            blocks.append(_Seq(*layers))
    return blocks, ch


def _time_embed(cfg):
    return nn.Sequential(nn.Linear(cfg.model_channels, cfg.time_embed_dim), nn.SiLU(),
                         nn.Linear(cfg.time_embed_dim, cfg.time_embed_dim))


def _hint_block(cfg):
    """input_hint_block (cldm.py:265-281, 599-615): 8 convs, SiLU between, indices 0,2,..,14."""
    spec = [(cfg.hint_channels, 16, 1), (16, 16, 1), (16, 32, 2), (32, 32, 1), (32, 96, 2), (96, 96, 1),
            (96, 256, 2), (256, cfg.model_channels, 1)]
    layers = []
    for i, (ci, co, s) in enumerate(spec):
        layers.append(nn.Conv2d(ci, co, 3, padding=1, stride=s))
        if i != len(spec) - 1:
            layers.append(nn.SiLU())
    return _Seq(*layers)


class _HotPathNet(nn.Module):
    """Common base: remembers the config, owns the lazily-built HIP engine program cache."""

    def __init__(self):
        super().__init__()
        self._md_engine = None

    def md_engine(self):
        from . import engine
        if self._md_engine is None:
            self._md_engine = engine.NetEngine(self)
        return self._md_engine

    def invalidate_engine(self):
        self._md_engine = None

    def _apply(self, fn, *a, **k):  # .to()/.cuda()/.half() move the master copy; repack lazily
        self._md_engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._md_engine = None
        return super().load_state_dict(*a, **k)


class ControlledUnetModelAttnPose(_HotPathNet):
    """The denoising UNet with bank 'read' + pose residuals (cldm.py:59-112).  kind = 'unet'."""
    kind = "unet"

    def __init__(self, image_size=None, **kw):
        super().__init__()
        self.cfg = cfg = _net_config(kw)
        self.image_size, self.in_channels, self.out_channels = image_size, cfg.in_channels, cfg.out_channels
        self.model_channels, self.dtype = cfg.model_channels, torch.float32
        self.time_embed = _time_embed(cfg)
        self.input_blocks, self.middle_block, chans, ch, ds = _make_encoder(cfg)
        self.output_blocks, ch = _make_decoder(cfg, chans, ch, ds)
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(),
                                 nn.Conv2d(cfg.model_channels, cfg.out_channels, 3, padding=1))

    def forward(self, x, timesteps=None, context=None, control=None, pose_control=None,
                only_mid_control=False, attention_mode=None, uc=False, **kwargs):
        """Same signature as cldm.py:60.  ``control`` = attention bank (list of [tensor]) or None,
        ``pose_control`` = list of 13 residuals (consumed by pop(), as in the reference)."""
        return self.md_engine().unet_forward(x, timesteps, context, control, pose_control,
                                             only_mid_control, attention_mode, uc)


class ControlledUnetModelAttn(ControlledUnetModelAttnPose):
    """Stage-1 denoising UNet: bank 'read' only, no pose residuals (cldm.py:115-161).  Same parameters / keys; a
    ``pose_control`` argument is accepted and ignored like the reference's."""

    def forward(self, x, timesteps=None, context=None, control=None, pose_control=None,
                only_mid_control=False, attention_mode=None, uc=False, **kwargs):
        return self.md_engine().unet_forward(x, timesteps, context, control, None, only_mid_control, attention_mode, uc)


class ControlNetReferenceOnly(_HotPathNet):
    """Appearance Control Model: full UNet topology, fills the attention bank (cldm.py:164-497)."""
    kind = "appearance"

    def __init__(self, image_size=None, **kw):
        super().__init__()
        self.cfg = cfg = _net_config(kw)
        self.image_size, self.in_channels, self.out_channels = image_size, cfg.in_channels, cfg.out_channels
        self.model_channels, self.dtype = cfg.model_channels, torch.float32
        self.time_embed = _time_embed(cfg)
        self.input_blocks, self.middle_block, chans, ch, ds = _make_encoder(cfg)
        # present in the checkpoint, never executed (cldm.py:473 is commented out in the reference)
        self.input_hint_block = _hint_block(cfg)
        self.output_blocks, ch = _make_decoder(cfg, chans, ch, ds)

    def forward(self, x, hint, timesteps, context, attention_bank=None, attention_mode=None, uc=False, **kwargs):
        """cldm.py:469-497: appends 16 x [norm1(x)] to ``attention_bank``; returns []."""
        return self.md_engine().appearance_forward(x, timesteps, context, attention_bank, attention_mode, uc)


class ControlNet(_HotPathNet):
    """OpenPose ControlNet: hint encoder + encoder half + 13 zero-convs (cldm.py:500-757)."""
    kind = "pose"

    def __init__(self, image_size=None, **kw):
        super().__init__()
        self.cfg = cfg = _net_config(kw)
        self.image_size, self.in_channels = image_size, cfg.in_channels
        self.model_channels, self.dtype = cfg.model_channels, torch.float32
        self.time_embed = _time_embed(cfg)
        self.input_blocks, self.middle_block, chans, ch, ds = _make_encoder(cfg)
        self.zero_convs = nn.ModuleList([_Seq(nn.Conv2d(c, c, 1)) for c in chans])
        self.input_hint_block = _hint_block(cfg)
        self.middle_block_out = _Seq(nn.Conv2d(ch, ch, 1))

    def forward(self, x, hint, timesteps, context, **kwargs):
        """cldm.py:736-757: returns the 13 residual tensors (NCHW, caller-owned)."""
        return self.md_engine().pose_forward(x, hint, timesteps, context)


def bank_shapes(cfg: NetConfig, latent_hw):
    """Token/channel shapes of the 16 bank entries in execution order."""
    h, w = latent_hw
    out = []
    ch, ds = cfg.model_channels, 1
    enc = []
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            ch = mult * cfg.model_channels
            if ds in cfg.attention_resolutions:
                enc.append(((h // ds) * (w // ds), ch))
        if level != len(cfg.channel_mult) - 1:
            ds *= 2
    out += enc
    out.append(((h // ds) * (w // ds), ch))
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ch = mult * cfg.model_channels
            if ds in cfg.attention_resolutions:
                out.append(((h // ds) * (w // ds), ch))
            if level and i == cfg.num_res_blocks:
                ds //= 2
    return out
