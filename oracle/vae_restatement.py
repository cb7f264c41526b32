"""TEST INFRASTRUCTURE ONLY.  CPU (torch fp32) restatement of the reference's first-stage model (SD-1.5 KL autoencoder),
the stage either side of the sampling loop.  Same rules as oracle/restatement.py: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, as the checker.

Pinning: the reference holds no golden vectors for this path ("parity unpinned by the reference's own tests"); the
restatement is pinned against outputs of the unmodified reference modules run in the build container
(oracle/make_golden.py vae_* cases -> tests/golden/vae_*.npz, checked by tests/test_oracle_vae_golden.py).

Functional walk over a flat state dict with the reference's key names; paths relative to
/root/reference/model_lib/ControlNet/.
"""
import torch
import torch.nn.functional as F

SD15_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                     num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def _gn_swish(sd, pre, x, swish=True):
    """Normalize = GroupNorm(32, eps=1e-6) (ldm/modules/diffusionmodules/model.py:46-47); nonlinearity = x*sigmoid(x) (:41-43)."""
    h = F.group_norm(x, 32, sd[pre + "weight"], sd[pre + "bias"], eps=1e-6)
    return h * torch.sigmoid(h) if swish else h


def _conv(sd, pre, x, stride=1, padding=1):
    return F.conv2d(x, sd[pre + "weight"], sd[pre + "bias"], stride=stride, padding=padding)


def resnet_block(sd, pre, x):
    """ResnetBlock.forward with temb=None (model.py:129-149)."""
    h = _conv(sd, pre + "conv1.", _gn_swish(sd, pre + "norm1.", x))
    h = _conv(sd, pre + "conv2.", _gn_swish(sd, pre + "norm2.", h))
    if pre + "nin_shortcut.weight" in sd:
        x = _conv(sd, pre + "nin_shortcut.", x, padding=0)
    return x + h


def attn_block(sd, pre, x):
    """AttnBlock.forward (model.py:179-203): single-head attention over the h*w positions, width c."""
    h = _gn_swish(sd, pre + "norm.", x, swish=False)
    q, k, v = (_conv(sd, pre + n + ".", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    h = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, pre + "proj_out.", h, padding=0)


def _mid(sd, pre, h):
    h = resnet_block(sd, pre + "block_1.", h)
    h = attn_block(sd, pre + "attn_1.", h)
    return resnet_block(sd, pre + "block_2.", h)


def decoder(sd, pre, z, ddconfig):
    """Decoder.forward (model.py:619-652)."""
    nres, nblk = len(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
    h = _conv(sd, pre + "conv_in.", z)
    h = _mid(sd, pre + "mid.", h)
    for lvl in reversed(range(nres)):
        for i in range(nblk + 1):
            h = resnet_block(sd, f"{pre}up.{lvl}.block.{i}.", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")        # Upsample.forward (:61-65)
            h = _conv(sd, f"{pre}up.{lvl}.upsample.conv.", h)
    return _conv(sd, pre + "conv_out.", _gn_swish(sd, pre + "norm_out.", h))


def encoder(sd, pre, x, ddconfig):
    """Encoder.forward (model.py:518-543)."""
    nres, nblk = len(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
    h = _conv(sd, pre + "conv_in.", x)
    for lvl in range(nres):
        for i in range(nblk):
            h = resnet_block(sd, f"{pre}down.{lvl}.block.{i}.", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)          # Downsample.forward (:80-84)
            h = _conv(sd, f"{pre}down.{lvl}.downsample.conv.", h, stride=2, padding=0)
    h = _mid(sd, pre + "mid.", h)
    return _conv(sd, pre + "conv_out.", _gn_swish(sd, pre + "norm_out.", h))


def vae_decode(sd, pre, z, ddconfig=SD15_DDCONFIG):
    """AutoencoderKL.decode (ldm/models/autoencoder.py:88-91); ``z`` already divided by scale_factor
    (decode_first_stage, ldm/models/diffusion/ddpm.py:2107-2108)."""
    return decoder(sd, pre + "decoder.", _conv(sd, pre + "post_quant_conv.", z, padding=0), ddconfig)


def vae_encode_moments(sd, pre, x, ddconfig=SD15_DDCONFIG):
    """AutoencoderKL.encode up to the posterior parameters (autoencoder.py:82-86): [B, 2*z, h, w] = mean | logvar."""
    return _conv(sd, pre + "quant_conv.", encoder(sd, pre + "encoder.", x, ddconfig), padding=0)


def posterior_sample(moments, noise):
    """DiagonalGaussianDistribution (ldm/modules/distributions/distributions.py:24-37) with the noise supplied."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise
