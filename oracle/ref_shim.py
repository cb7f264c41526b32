"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *unmodified* reference hot-path code from /root/reference on CPU so that
(a) the torch-fp32 restatement in oracle/restatement.py can be validated against it and
(b) golden input/output vectors can be generated (oracle/make_golden.py).

/root/reference exists only in the build container; nothing under tests/ -m gpu, smoke() or
bench.py may call this module.  Recipe follows SURVEY.md section 8(c): stub the packages absent from
the image, import attention.py before faking xformers so the vanilla CrossAttention
(model_lib/ControlNet/ldm/modules/attention.py:146-199) is selected, drop the sampler's forced
.to("cuda") (ldm/models/diffusion/ddim.py:353-357).
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("MAGICDANCE_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model_lib", "ControlNet"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_LOADED = {}


def load_reference():
    """Returns a namespace with the reference modules (cldm, ddim, attention, openaimodel, util)."""
    if _LOADED:
        return types.SimpleNamespace(**_LOADED)
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)

    ident = lambda f: f
    if "pytorch_lightning" not in sys.modules:
        _mod("pytorch_lightning", LightningModule=nn.Module)
        _mod("pytorch_lightning.utilities")
        _mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=ident)
        _mod("pytorch_lightning.utilities.distributed", rank_zero_only=ident)
        _mod("pytorch_lightning.callbacks", Callback=object)

    class ListConfig(list):
        pass

    if "omegaconf" not in sys.modules:
        _mod("omegaconf", ListConfig=ListConfig)
        _mod("omegaconf.listconfig", ListConfig=ListConfig)
    if "torchvision" not in sys.modules:
        _mod("torchvision")
        _mod("torchvision.utils", make_grid=lambda *a, **k: None, save_image=lambda *a, **k: None)

    class _D:
        def __init__(self, *a, **k):
            pass

    if "diffusers" not in sys.modules:
        _mod("diffusers")
        _mod("diffusers.configuration_utils", ConfigMixin=_D, register_to_config=ident)
        _mod("diffusers.modeling_utils", ModelMixin=nn.Module)
        _mod("diffusers.utils", BaseOutput=_D)
        _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
        _mod("diffusers.models")
        _mod("diffusers.models.attention", CrossAttention=nn.Module, FeedForward=nn.Module)
    if "clip" not in sys.modules:
        _mod("clip")

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import model_lib.ControlNet.ldm.modules.attention as att
    assert att.XFORMERS_IS_AVAILBLE is False
    if "xformers" not in sys.modules:
        _mod("xformers")
        _mod("xformers.ops")
    import model_lib.ControlNet.ldm.modules.diffusionmodules.model as vae
    vae.XFORMERS_IS_AVAILBLE = False
    from model_lib.ControlNet.cldm import cldm
    from model_lib.ControlNet.ldm.util import instantiate_from_config
    from model_lib.ControlNet.ldm.models.diffusion import ddim
    import model_lib.ControlNet.ldm.modules.diffusionmodules.openaimodel as oam
    import model_lib.ControlNet.ldm.modules.diffusionmodules.util as util

    ddim.DDIMSampler_ReferenceOnly.register_buffer = lambda self, n, a: setattr(self, n, a)
    from model_lib.ControlNet.ldm.models import autoencoder
    _LOADED.update(cldm=cldm, ddim=ddim, attention=att, openaimodel=oam, util=util, autoencoder=autoencoder, vae_modules=vae,
                   instantiate_from_config=instantiate_from_config)
    return types.SimpleNamespace(**_LOADED)


def reference_yaml_config(overrides=None, stage1=False):
    """The reference's own YAML (CN/models/cldm_v15_reference_only_pose.yaml) as plain dicts, with
    the three network param blocks optionally overridden (small test geometries) and CLIP/VAE
    replaced by the reference's own '__is_unconditional__' / identity escape hatches."""
    import yaml
    path = os.path.join(REFERENCE_ROOT, "model_lib/ControlNet/models/cldm_v15_reference_only%s.yaml" % ("" if stage1 else "_pose"))
    cfg = yaml.safe_load(open(path))["model"]
    cfg["params"]["cond_stage_config"] = "__is_unconditional__"
    if overrides:
        for blk in (("control_stage_config", "unet_config") if stage1 else
                    ("appearance_control_stage_config", "pose_control_stage_config", "unet_config")):
            cfg["params"][blk]["params"].update(overrides)
    return cfg


def build_reference_model(overrides=None, image_size=None, with_vae=False, stage1=False):
    ref = load_reference()
    cfg = reference_yaml_config(overrides, stage1=stage1)
    if not with_vae:
        # shrink the VAE (unused on the hot path) so construction is quick
        dd = cfg["params"]["first_stage_config"]["params"]["ddconfig"]
        dd.update(ch=32, ch_mult=[1], num_res_blocks=1)
    model = ref.instantiate_from_config(cfg).eval()
    type(model).device = property(lambda s: torch.device("cpu"))
    if image_size is not None:
        model.image_size = image_size
    return model


def build_reference_vae(ddconfig):
    """The unmodified reference AutoencoderKL (ldm/models/autoencoder.py) with the vanilla AttnBlock (xformers absent,
    ldm/modules/diffusionmodules/model.py:280-293)."""
    ref = load_reference()
    return ref.autoencoder.AutoencoderKL(ddconfig=dict(ddconfig), lossconfig={"target": "torch.nn.Identity"},
                                         embed_dim=ddconfig["z_channels"]).eval()
