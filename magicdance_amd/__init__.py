"""magicdance_amd: MI355X-native (gfx950) implementation of MagicDance's diffusion-sampling hot path.

Public surface = the reference's for this path: ``create_model`` / ``load_state_dict`` (cldm/model.py),
``ControlLDMReferenceOnlyPose`` with ``apply_model`` / ``sample_log`` (cldm/cldm.py, ddpm.py),
``DDIMSampler_ReferenceOnly`` (ddim.py) and the three network classes named by the YAML ``target:`` strings.
All arithmetic runs in libmagicdance_hip.so (include/magicdance_hip.h); importing this package does not load it,
using any op without it raises.
"""
from .cldm import (ControlLDMReferenceOnlyPose, ControlLDMReferenceOnly, create_model, load_state_dict, instantiate_from_config,  # noqa: F401
                   DEFAULT_CONFIG)
from .nets import ControlledUnetModelAttnPose, ControlledUnetModelAttn, ControlNetReferenceOnly, ControlNet  # noqa: F401

__version__ = "0.1.0"
