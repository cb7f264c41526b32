"""First-stage model (SD-1.5 KL autoencoder) on the MI355X kernels -- SURVEY 8(f) rank 1, the stage either side of the
sampling loop: ``encode`` once per reference image, ``decode`` once per generated frame.

``AutoencoderKL`` mirrors the constructor kwargs and the state-dict key layout of the reference
(model_lib/ControlNet/ldm/models/autoencoder.py:14-91 on ldm/modules/diffusionmodules/model.py:452-652), so the
``first_stage_model.*`` keys of a ``model_state-*.th`` checkpoint load strict and the YAML only swaps the ``target:``.
It holds fp32 master parameters; all arithmetic is the HIP engine below (NHWC fp16 activations, fp32 accumulate):

  ResnetBlock (model.py:90-149)  GroupNorm(32, eps 1e-6)+SiLU fused -> md_igemm 3x3 -> GN+SiLU -> md_igemm 3x3 with
                                 the skip (identity or 1x1 nin_shortcut) fused as the residual epilogue
  AttnBlock   (model.py:152-203) single head, d = C = 512 over h*w tokens: q and fused k|v 1x1 projections (V stored
                                 transposed), scores = md_igemm(Q, K) in fp32, md_softmax_rows, P.V = md_igemm(P, V^T),
                                 proj_out with the residual fused -- d = 512 does not fit the flash kernel's register
                                 budget and the score matrix (64 MB fp32 per sample at 64x64) is nothing in 288 GB
  Upsample    (model.py:50-65)   nearest x2 folded into the conv gather (ups=1)
  Downsample  (model.py:68-87)   F.pad (0,1,0,1) + conv3x3 stride 2 pad 0 = md_igemm asym_pad
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import MD_ACT_NONE
from .engine import Act, F16, F32, _f, _h, _require_gpu, get_arena, is_tiled, pack_conv, _WS


# ----------------------------------------------------------------------------- parameter containers
def _norm(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = _norm(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = _norm(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1)


class AttnBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.in_channels = c
        self.norm = _norm(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))


class _Resample(nn.Module):
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=0 if stride == 2 else 1)


def _mid(c):
    m = nn.Module()
    m.block_1, m.attn_1, m.block_2 = ResnetBlock(c, c), AttnBlock(c), ResnetBlock(c, c)
    return m


class Encoder(nn.Module):
    """Keys: conv_in, down.{l}.block.{i}, down.{l}.downsample.conv, mid.{block_1,attn_1,block_2}, norm_out, conv_out."""

    def __init__(self, *, ch, out_ch=3, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, **ignored):
        super().__init__()
        if len(attn_resolutions) or not resamp_with_conv or dropout != 0.0:
            raise NotImplementedError("the SD-1.5 first stage has no per-level attention, conv resampling, no dropout")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for lvl in range(self.num_resolutions):
            block_in, block_out = ch * in_mult[lvl], ch * ch_mult[lvl]
            d = nn.Module()
            d.block = nn.ModuleList()
            d.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                d.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if lvl != self.num_resolutions - 1:
                d.downsample = _Resample(block_in, 2)
            self.down.append(d)
        self.mid = _mid(block_in)
        self.norm_out = _norm(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)


class Decoder(nn.Module):
    """Keys: conv_in, mid.*, up.{l}.block.{i}, up.{l}.upsample.conv (l counted from the full-resolution end), norm_out,
    conv_out."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels=3, resolution=256, z_channels, give_pre_end=False, tanh_out=False,
                 **ignored):
        super().__init__()
        if len(attn_resolutions) or not resamp_with_conv or give_pre_end or tanh_out:
            raise NotImplementedError("not an SD-1.5 first-stage decoder configuration")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = _mid(block_in)
        self.up = nn.ModuleList()
        for lvl in reversed(range(self.num_resolutions)):
            block_out = ch * ch_mult[lvl]
            u = nn.Module()
            u.block = nn.ModuleList()
            u.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                u.block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            if lvl != 0:
                u.upsample = _Resample(block_in, 1)
            self.up.insert(0, u)
        self.norm_out = _norm(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)


class DiagonalGaussianDistribution:
    """ldm/modules/distributions/distributions.py:24-62 (sample / mode only; the noise is drawn on the host exactly like
    the reference, ``torch.randn(shape).to(device)``, so a seeded run consumes the same RNG stream)."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std, self.var = torch.exp(0.5 * self.logvar), torch.exp(self.logvar)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, ema_decay=None, learn_logvar=False):
        super().__init__()
        assert ddconfig["double_z"]
        if ema_decay is not None or ckpt_path is not None:
            raise NotImplementedError("EMA / standalone VAE checkpoints are training-side features")
        self.encoder, self.decoder = Encoder(**ddconfig), Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim, self.image_key = embed_dim, image_key
        self._engine = None

    def md_engine(self):
        if self._engine is None:
            self._engine = VaeEngine(self)
        return self._engine

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def encode(self, x):
        """autoencoder.py:82-86: x [B,3,H,W] in [-1,1] -> posterior over [B,4,H/8,W/8]."""
        return DiagonalGaussianDistribution(self.md_engine().encode_moments(x))

    @torch.no_grad()
    def decode(self, z):
        """autoencoder.py:88-91: z [B,4,h,w] (already divided by scale_factor) -> image [B,3,8h,8w] fp32."""
        return self.md_engine().decode(z)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        return self.decode(posterior.sample() if sample_posterior else posterior.mode()), posterior


# ----------------------------------------------------------------------------- HIP engine
class VaeEngine:
    MAX_BATCH = 8   # per pass: keeps every tensor inside the kernels' 32-bit byte offsets at 512x512 (0.5 GB / tensor)

    def __init__(self, vae):
        p = next(vae.parameters())
        _require_gpu(p.device)
        self.device = d = p.device
        self.arena = get_arena(d)

        def res(m):
            r = dict(cin=m.in_channels, cout=m.out_channels, gn1=(_f(m.norm1.weight, d), _f(m.norm1.bias, d)),
                     conv1_w=pack_conv(m.conv1.weight, d), conv1_b=_f(m.conv1.bias, d),
                     gn2=(_f(m.norm2.weight, d), _f(m.norm2.bias, d)),
                     conv2_w=pack_conv(m.conv2.weight, d), conv2_b=_f(m.conv2.bias, d))
            if hasattr(m, "nin_shortcut"):
                r["skip_w"], r["skip_b"] = pack_conv(m.nin_shortcut.weight, d), _f(m.nin_shortcut.bias, d)
            return r

        def attn(m):
            c = m.in_channels
            w = torch.cat([m.q.weight, m.k.weight, m.v.weight], 0).reshape(3 * c, c)
            b = torch.cat([m.q.bias, m.k.bias, m.v.bias], 0)
            return dict(c=c, gn=(_f(m.norm.weight, d), _f(m.norm.bias, d)), qkv_w=_h(w, d), qkv_b=_f(b, d),
                        o_w=pack_conv(m.proj_out.weight, d), o_b=_f(m.proj_out.bias, d))

        def mid(m):
            return dict(b1=res(m.block_1), attn=attn(m.attn_1), b2=res(m.block_2))

        def conv(m, cin_pad=None, cout_pad=None):
            w, b = m.weight.detach(), m.bias.detach()
            if cout_pad is not None and cout_pad > w.shape[0]:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_pad - w.shape[0]))
                b = torch.nn.functional.pad(b, (0, cout_pad - b.shape[0]))
            return dict(w=pack_conv(w, d, cin_pad=cin_pad), b=_f(b, d), cout=w.shape[0], k=w.shape[2])

        dec, enc = vae.decoder, vae.encoder
        self.dec = dict(post_quant=conv(vae.post_quant_conv, cin_pad=8, cout_pad=8), conv_in=conv(dec.conv_in, cin_pad=8),
                        mid=mid(dec.mid),
                        up=[dict(blocks=[res(b) for b in u.block], ups=conv(u.upsample.conv) if hasattr(u, "upsample") else None)
                            for u in dec.up],
                        gn_out=(_f(dec.norm_out.weight, d), _f(dec.norm_out.bias, d)),
                        conv_out=conv(dec.conv_out, cout_pad=4), out_ch=dec.conv_out.out_channels)
        self.enc = dict(conv_in=conv(enc.conv_in, cin_pad=8),
                        down=[dict(blocks=[res(b) for b in dn.block], down=conv(dn.downsample.conv) if hasattr(dn, "downsample") else None)
                              for dn in enc.down],
                        mid=mid(enc.mid), gn_out=(_f(enc.norm_out.weight, d), _f(enc.norm_out.bias, d)),
                        conv_out=conv(enc.conv_out), quant=conv(vae.quant_conv), zc2=vae.quant_conv.out_channels)

    # ------------------------------------------------------------------ helpers (same conventions as NetEngine)
    def _ws(self):
        key = (id(self), "vae")
        buf = _WS.get(key)
        if buf is None:
            buf = _WS[key] = torch.empty(96 << 20, dtype=torch.uint8, device=self.device)
            buf[:16384].zero_()
        return buf

    def _gn_ws(self):
        key = (id(self), "vae-gn")
        buf = _WS.get(key)
        if buf is None:
            buf = _WS[key] = torch.empty(1 << 20, dtype=torch.uint8, device=self.device)
        return buf

    def conv(self, x, cv, *, stride=1, ups=0, asym=False, res=None, out_f32=False, stats=True):
        """``stats``: let the epilogue write the GroupNorm partial statistics of the output (Act.part) where the GroupNorm that may
        consume it would otherwise run its own statistics pass (engine.NetEngine.want_part)."""
        from .engine import NetEngine
        n, k = cv["cout"], cv["k"]
        if ups:
            ho, wo = 2 * x.h, 2 * x.w
        elif stride == 2:
            ho, wo = x.h // 2, x.w // 2
        else:
            ho, wo = x.h, x.w
        out = self.arena.alloc((x.b, ho * wo, n), F32 if out_f32 else F16)
        part = None
        if stats and not out_f32 and NetEngine.want_part(x.b, ho * wo, n):
            part = self.arena.alloc((x.b * ho * wo // 64, 2, n), F32)
        ops.igemm(x.t, cv["w"], n, batch=x.b, hin=x.h, win=x.w, hout=ho, wout=wo, c0=x.c, ksize=k, stride=stride, ups=ups,
                  bias=cv["b"], res=None if res is None else res.t, ld_res=0 if res is None else res.c, act=MD_ACT_NONE,
                  out=out, ld_out=n, out_f32=out_f32, ws=self._ws(), asym_pad=asym, gn_part=part, w_tiled=is_tiled(cv["w"]))
        return Act(out, x.b, ho, wo, n, None, part)

    def gn(self, x, gb, silu=True):
        out = self.arena.alloc((x.b, x.hw, x.c), F16)
        ops.groupnorm(x.t, gb[0], gb[1], out, self._gn_ws(), batch=x.b, hw=x.hw, c0=x.c, groups=32, eps=1e-6, silu=silu,
                      part0=x.part)
        return Act(out, x.b, x.h, x.w, x.c)

    def resblock(self, r, x):
        h = self.gn(x, r["gn1"])
        h = self.conv(h, dict(w=r["conv1_w"], b=r["conv1_b"], cout=r["cout"], k=3))
        h = self.gn(h, r["gn2"])
        skip = self.conv(x, dict(w=r["skip_w"], b=r["skip_b"], cout=r["cout"], k=1), stats=False) if "skip_w" in r else x
        return self.conv(h, dict(w=r["conv2_w"], b=r["conv2_b"], cout=r["cout"], k=3), res=skip)

    def attnblock(self, a, x):
        """AttnBlock.forward (model.py:179-203): softmax(q k^T c^-1/2) v over the h*w tokens, one head of width c."""
        b, n, c = x.b, x.hw, a["c"]
        hn = self.gn(x, a["gn"], silu=False)
        # q and k stay contiguous [n, c]: they are the A / weight operands of the score contraction below
        q = self.arena.alloc((b, n, c), F16)
        k = self.arena.alloc((b, n, c), F16)
        vt = self.arena.alloc((b, c, n), F16)
        ops.igemm(hn.t, a["qkv_w"], c, batch=b, hin=1, win=n, hout=1, wout=n, c0=c, bias=a["qkv_b"], out=q, ld_out=c,
                  ws=self._ws())
        ops.igemm(hn.t, a["qkv_w"][c:], 2 * c, batch=b, hin=1, win=n, hout=1, wout=n, c0=c, bias=a["qkv_b"][c:], out=k,
                  ld_out=c, out_t=vt, n_tr_begin=c, ld_t=n, ws=self._ws())
        att = self.arena.alloc((b, n, c), F16)
        sc = self.arena.alloc((n, n), F32)
        pr = self.arena.alloc((n, n), F16)
        for i in range(b):  # K / V^T act as the per-sample "weight" operand of the two contractions
            ops.igemm(q[i], k[i], n, batch=1, hin=1, win=n, hout=1, wout=n, c0=c, out=sc, ld_out=n, out_f32=True,
                      ws=self._ws())
            ops.softmax_rows(sc, n, pr, n, n, n, float(c) ** -0.5)
            ops.igemm(pr, vt[i], c, batch=1, hin=1, win=n, hout=1, wout=n, c0=n, out=att[i], ld_out=c, ws=self._ws())
        return self.conv(Act(att, b, x.h, x.w, c), dict(w=a["o_w"], b=a["o_b"], cout=c, k=1), res=x)

    # ------------------------------------------------------------------ the two networks
    def _to_nhwc(self, x, cpad):
        b, c, hh, ww = x.shape
        x = x.detach().to(device=self.device, dtype=F32).contiguous()
        t = self.arena.alloc((b, hh * ww, cpad), F16)
        ops.nchw_to_nhwc_f16(x, t, b, c, hh * ww, cpad)
        return Act(t, b, hh, ww, cpad)

    def _to_nchw(self, a, c):
        out = torch.empty((a.b, c, a.h, a.w), dtype=F32, device=self.device)
        ops.nhwc_to_nchw_f32(a.t, out, a.b, c, a.hw, a.c)
        return out

    def decode(self, z):
        if z.shape[0] > self.MAX_BATCH:
            return torch.cat([self.decode(z[i:i + self.MAX_BATCH]) for i in range(0, z.shape[0], self.MAX_BATCH)], 0)
        d = self.dec
        self.arena.reset()
        h = self._to_nhwc(z, 8)
        h = self.conv(h, d["post_quant"])
        h = self.conv(h, d["conv_in"])
        h = self.resblock(d["mid"]["b1"], h)
        h = self.attnblock(d["mid"]["attn"], h)
        h = self.resblock(d["mid"]["b2"], h)
        for lvl in reversed(range(len(d["up"]))):
            for r in d["up"][lvl]["blocks"]:
                h = self.resblock(r, h)
            if d["up"][lvl]["ups"] is not None:
                h = self.conv(h, d["up"][lvl]["ups"], ups=1)
        h = self.gn(h, d["gn_out"])
        h = self.conv(h, d["conv_out"], out_f32=True)
        return self._to_nchw(h, d["out_ch"])

    def encode_moments(self, x):
        if x.shape[0] > self.MAX_BATCH:
            return torch.cat([self.encode_moments(x[i:i + self.MAX_BATCH]) for i in range(0, x.shape[0], self.MAX_BATCH)], 0)
        e = self.enc
        self.arena.reset()
        h = self._to_nhwc(x, 8)
        h = self.conv(h, e["conv_in"])
        for lvl in e["down"]:
            for r in lvl["blocks"]:
                h = self.resblock(r, h)
            if lvl["down"] is not None:
                h = self.conv(h, lvl["down"], stride=2, asym=True)
        h = self.resblock(e["mid"]["b1"], h)
        h = self.attnblock(e["mid"]["attn"], h)
        h = self.resblock(e["mid"]["b2"], h)
        h = self.gn(h, e["gn_out"])
        h = self.conv(h, e["conv_out"])
        h = self.conv(h, e["quant"], out_f32=True)
        return self._to_nchw(h, e["zc2"])
