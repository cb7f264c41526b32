"""Drop-in for the reference's ControlLDM surface on the sampling hot path.

Mirrors (same names, ctor kwargs, call signatures, cond-dict keys, state-dict key prefixes):
  ControlLDMReferenceOnlyPose          model_lib/ControlNet/cldm/cldm.py:1087-1121
  LatentDiffusionReferenceOnly bits    model_lib/ControlNet/ldm/models/diffusion/ddpm.py:138-192 (register_schedule),
                                       :356-359 (q_sample), :1935 (get_first_stage_encoding), :2402-2413 (sample_log)
  DiffusionWrapper                     ddpm.py:1313-1352 (only the 'crossattn' route the pose config uses)
  create_model / instantiate_from_config   cldm/model.py:24-28, ldm/util.py:72-87
The YAML (configs/cldm_v15_reference_only_pose.yaml) differs from the reference's only in six ``target:`` strings (model, three
networks, first stage, text encoder).  The first stage (VAE) is this package's own ``autoencoder.AutoencoderKL`` on the
same kernels; the text encoder is ``clip.FrozenCLIPEmbedder``: the transformers module as parameter container and
tokenizer, its arithmetic on the GPU through ``clip.ClipTextEngine`` (this library's kernels).
"""
import importlib
import os

import numpy as np
import torch
import torch.nn as nn

from . import engine as E
from . import ops

F32 = torch.float32


# ----------------------------------------------------------------------------------------------- config plumbing
def get_obj_from_str(string):
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config):
    """ldm/util.py:72-79: {'target': 'pkg.mod.Class', 'params': {...}} -> object."""
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**dict(config.get("params", dict()) or dict()))


def load_config(path):
    """OmegaConf.load replacement: plain dicts (omegaconf is not in the image; the YAML uses no interpolation)."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


def create_model(config_path):
    """cldm/model.py:24-28."""
    config = load_config(config_path)
    model = instantiate_from_config(config["model"]).cpu()
    print(f"Loaded model config from [{config_path}]")
    return model


def get_state_dict(d):
    return d.get("state_dict", d)


def load_state_dict(ckpt_path, location="cpu"):
    """cldm/model.py:12-21 (``.th`` / ``.ckpt`` pickles and ``.safetensors``)."""
    _, ext = os.path.splitext(ckpt_path)
    if ext.lower() == ".safetensors":
        import safetensors.torch
        sd = safetensors.torch.load_file(ckpt_path, device=location)
    else:
        sd = get_state_dict(torch.load(ckpt_path, map_location=torch.device(location)))
    sd = get_state_dict(sd)
    print(f"Loaded state_dict from [{ckpt_path}]")
    return sd


DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "cldm_v15_reference_only_pose.yaml")


# ----------------------------------------------------------------------------------------------- schedule
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """ldm/modules/diffusionmodules/util.py:20-42 (the 'linear' schedule the SD-1.5 configs use)."""
    if schedule != "linear":
        raise NotImplementedError(f"beta schedule '{schedule}' is not used by the pose config")
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64, device="cpu") ** 2).numpy()


def adapt_clip_keys(state_dict, model_keys, prefix="cond_stage_model.transformer."):
    """Checkpoints written with transformers 4.x name the text tower ``<prefix>text_model.*`` (and carry a ``position_ids``
    buffer); transformers 5 drops that level.  Rename towards whatever the instantiated module uses."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith(prefix) and k not in model_keys:
            rest = k[len(prefix):]
            alt = prefix + (rest[len("text_model."):] if rest.startswith("text_model.") else "text_model." + rest)
            if alt in model_keys:
                k = alt
            elif rest.endswith("position_ids"):
                continue
        out[k] = v
    return out


class _Unavailable(nn.Module):
    """Placeholder for VAE / CLIP when their implementation is not importable in this image."""

    def __init__(self, what, err):
        super().__init__()
        self.what, self.err = what, err

    def __getattr__(self, name):
        if name in ("what", "err") or name.startswith("_"):
            return super().__getattr__(name)
        raise RuntimeError(f"{self.what} is outside the MI355X hot path of this build and could not be instantiated "
                           f"({self.err}); pass latents / context tensors directly")


class DiffusionWrapper(nn.Module):
    """ddpm.py:1313-1352 -- holds the UNet under the ``diffusion_model`` key."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert conditioning_key in (None, "crossattn"), "the pose config routes text through cross-attention only"


class ControlLDMReferenceOnlyPose(nn.Module):
    has_pose = True

    def __init__(self, control_key=None, only_mid_control=False, appearance_control_stage_config=None,
                 pose_control_stage_config=None, unet_config=None, first_stage_config=None, cond_stage_config=None,
                 timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3,
                 image_size=64, channels=4, scale_factor=1.0, conditioning_key="crossattn", parameterization="eps",
                 v_posterior=0.0, logvar_init=0.0, use_ema=False, first_stage_key="image", cond_stage_key="image",
                 cond_stage_trainable=False, num_timesteps_cond=1, log_every_t=100, monitor=None, **unused):
        super().__init__()
        if parameterization != "eps":
            raise NotImplementedError("only the eps parameterisation is on the hot path (ddim.py:617-624)")
        if use_ema:
            raise NotImplementedError("EMA weights are a training feature")
        self.parameterization = parameterization
        self.control_key, self.only_mid_control, self.control_enabled = control_key, only_mid_control, True
        self.image_size, self.channels, self.scale_factor = image_size, channels, scale_factor
        self.first_stage_key, self.cond_stage_key = first_stage_key, cond_stage_key
        self.cond_stage_trainable, self.num_timesteps_cond, self.log_every_t = cond_stage_trainable, num_timesteps_cond, log_every_t
        self.conditioning_key, self.v_posterior, self.use_ema = conditioning_key, v_posterior, use_ema
        self.sd_locked = True
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self._build_control(appearance_control_stage_config, pose_control_stage_config)
        self.first_stage_model = self._optional(first_stage_config, "first_stage_model (VAE)")
        self.cond_stage_model = self._optional(cond_stage_config, "cond_stage_model (CLIP text encoder)")
        self.register_schedule(beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start,
                               linear_end=linear_end, cosine_s=cosine_s)
        self.register_buffer("logvar", torch.full(fill_value=logvar_init, size=(self.num_timesteps,)))
        self._fused = None

    def _build_control(self, appearance_cfg, pose_cfg):
        self.appearance_control_model = instantiate_from_config(appearance_cfg)
        self.pose_control_model = instantiate_from_config(pose_cfg)

    @staticmethod
    def _optional(config, what):
        if config is None or isinstance(config, str):
            return None
        try:
            m = instantiate_from_config(config)
        except Exception as e:  # noqa: BLE001 -- absent third-party packages (omegaconf, clip, xformers ...)
            return _Unavailable(what, f"{type(e).__name__}: {e}")
        if m is not None:
            m = m.eval()
            for p in m.parameters():
                p.requires_grad = False
        return m

    # ------------------------------------------------------------------ schedule buffers (ddpm.py:138-192)
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(beta_schedule, timesteps, linear_start,
                                                                               linear_end, cosine_s)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        tt = lambda a: torch.tensor(a, dtype=torch.float32)  # noqa: E731
        self.register_buffer("betas", tt(betas))
        self.register_buffer("alphas_cumprod", tt(ac))
        self.register_buffer("alphas_cumprod_prev", tt(ac_prev))
        self.register_buffer("sqrt_alphas_cumprod", tt(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", tt(np.sqrt(1.0 - ac)))
        self.register_buffer("log_one_minus_alphas_cumprod", tt(np.log(1.0 - ac)))
        self.register_buffer("sqrt_recip_alphas_cumprod", tt(np.sqrt(1.0 / ac)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", tt(np.sqrt(1.0 / ac - 1)))
        pv = (1 - self.v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + self.v_posterior * betas
        self.register_buffer("posterior_variance", tt(pv))
        self.register_buffer("posterior_log_variance_clipped", tt(np.log(np.maximum(pv, 1e-20))))
        self.register_buffer("posterior_mean_coef1", tt(betas * np.sqrt(ac_prev) / (1.0 - ac)))
        self.register_buffer("posterior_mean_coef2", tt((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)))

    @property
    def device(self):
        return self.betas.device

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Checkpoints (``model_state-*.th``) carry VAE / CLIP weights; when those sub-modules are not built here
        their keys are dropped instead of failing a strict load (test_any_image_pose.py:371 loads strict)."""
        drop = tuple(p for p, m in (("first_stage_model.", self.first_stage_model), ("cond_stage_model.", self.cond_stage_model))
                     if m is None or isinstance(m, _Unavailable))
        if drop:
            state_dict = {k: v for k, v in state_dict.items() if not k.startswith(drop)}
        if not any(p.startswith("cond_stage_model") for p in drop):
            state_dict = adapt_clip_keys(state_dict, set(self.state_dict().keys()))
        if hasattr(self.cond_stage_model, "note_loaded_keys"):
            self.cond_stage_model.note_loaded_keys(state_dict)
        self._fused = None
        # nn.Module.load_state_dict recurses through _load_from_state_dict, not through the children's load_state_dict
        # overrides: drop every packed-fp16 engine here, or a second checkpoint would render with the first one's weights
        for m in self.modules():
            if hasattr(m, "invalidate_engine"):
                m.invalidate_engine()
            elif hasattr(m, "_engine") and hasattr(m, "md_engine"):
                m._engine = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # ------------------------------------------------------------------ VAE / CLIP glue (the VAE is magicdance_amd.autoencoder)
    def get_learned_conditioning(self, c):
        if self.cond_stage_model is None:
            raise RuntimeError("no cond_stage_model: pass the [B,77,768] context tensor directly")
        return self.cond_stage_model.encode(c) if hasattr(self.cond_stage_model, "encode") else self.cond_stage_model(c)

    def get_unconditional_conditioning(self, N):
        return self.get_learned_conditioning([""] * N)                            # cldm.py:1119-1121

    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    def decode_first_stage(self, z):
        return self.first_stage_model.decode(1.0 / self.scale_factor * z)          # ddpm.py:2100-2128

    def get_first_stage_encoding(self, encoder_posterior):
        z = encoder_posterior.sample() if hasattr(encoder_posterior, "sample") else encoder_posterior
        return self.scale_factor * z                                               # ddpm.py:1935-1943

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:356-359."""
        noise = torch.randn_like(x_start) if noise is None else noise
        shp = (-1,) + (1,) * (x_start.dim() - 1)
        return (self.sqrt_alphas_cumprod[t].reshape(shp) * x_start +
                self.sqrt_one_minus_alphas_cumprod[t].reshape(shp) * noise)

    # ------------------------------------------------------------------ the hot path
    def engines(self):
        return (self.appearance_control_model.md_engine(), self.pose_control_model.md_engine() if self.has_pose else None,
                self.model.diffusion_model.md_engine())

    @torch.no_grad()
    def apply_model(self, x_noisy, t, cond, reference_image_noisy=None, uc=False, *args, **kwargs):
        """cldm.py:1099-1117: appearance net ('write') if a reference latent is given, pose ControlNet if c_concat,
        then the UNet ('read', or the plain branch when uc).  The reference also runs the pose net on the uc pass
        and discards its result (cldm.py:1112-1114 vs :70-84); that dead pass is skipped here.  Returns NCHW fp32."""
        eps = self.apply_model_nhwc(x_noisy, t, cond, reference_image_noisy, uc)
        b, hw, oc = eps.shape
        out = torch.empty((b, oc) + tuple(x_noisy.shape[2:]), dtype=F32, device=self.device)
        ops.nhwc_to_nchw_f32(eps, out, b, oc, hw, oc)
        return out

    @torch.no_grad()
    def apply_model_nhwc(self, x_noisy, t, cond, reference_image_noisy=None, uc=False):
        """apply_model with the kernels' native output layout: eps as NHWC fp32 [B, H*W, 4] (arena memory, valid
        until the next apply_model call)."""
        assert isinstance(cond, dict)
        app, pose_e, unet = self.engines()
        # (a one-element list -- every caller on this path -- is passed through AS the caller's tensor: the engines' context / hint
        #  caches recognise it by identity + version, without the device -> host content comparison a fresh torch.cat copy needs)
        cat1 = lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, 1)  # noqa: E731
        cond_txt = cat1(cond["c_crossattn"])
        if self.control_enabled and cond.get("c_crossattn_void") is not None:
            cond_txt_void = cat1(cond["c_crossattn_void"])
        else:
            cond_txt_void = cond_txt
        x = x_noisy.detach().to(device=self.device, dtype=F32).contiguous()
        b, _, hh, ww = x.shape
        arena = unet.arena
        arena.reset()
        t_dev = unet._t_dev(t, b)
        banks = None
        if reference_image_noisy is not None:
            ref = reference_image_noisy.detach().to(device=self.device, dtype=F32).contiguous()
            banks = app.appearance(ref, app._t_dev(t, ref.shape[0]), app.context_kv(cond_txt_void))
        pose = None
        if self.has_pose and self.control_enabled and cond.get("c_concat") is not None and not uc:
            hint = cat1(cond["c_concat"])
            pose = pose_e.pose(x, pose_e.hint_features(hint), t_dev, pose_e.context_kv(cond_txt_void))
        return unet.unet(x, t_dev, unet.context_kv(cond_txt), banks=None if uc else banks, pose=pose,
                         nread=0 if uc else b, only_mid_control=self.only_mid_control)

    @torch.no_grad()
    def sample_log(self, cond, batch_size, ddim, ddim_steps, **kwargs):
        """ddpm.py:2402-2413."""
        if not ddim:
            raise NotImplementedError("ancestral DDPM sampling is not used by the inference entry points")
        from .ddim import DDIMSampler_ReferenceOnly
        sampler = DDIMSampler_ReferenceOnly(self)
        shape = (self.channels, self.image_size, self.image_size)
        return sampler.sample(ddim_steps, batch_size, shape, cond, verbose=False, **kwargs)


class ControlLDMReferenceOnly(ControlLDMReferenceOnlyPose):
    """Stage-1 model (appearance control only): cldm/cldm.py:1055-1081 with models/cldm_v15_reference_only.yaml.  The
    appearance net sits under ``control_model.*`` (ctor kwarg ``control_stage_config``), there is no pose ControlNet and
    ``apply_model`` never looks at ``c_concat``.  Sampling takes the generic per-step route (same kernels; the fused
    graph route is specific to the pose model)."""
    has_pose = False

    def __init__(self, control_key=None, only_mid_control=False, control_stage_config=None, **kw):
        self._stage1_cfg = control_stage_config
        super().__init__(control_key=control_key, only_mid_control=only_mid_control, **kw)

    def _build_control(self, appearance_cfg, pose_cfg):
        self.control_model = instantiate_from_config(self._stage1_cfg)

    @property
    def appearance_control_model(self):
        return self.control_model
