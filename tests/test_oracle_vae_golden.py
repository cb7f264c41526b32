"""CPU: pins oracle/vae_restatement.py against vectors produced by the unmodified reference AutoencoderKL
(oracle/make_golden.py vae_* cases).  fp32 vs fp32 on the same CPU kernels -> rounding-level tolerance."""
import numpy as np
import pytest
import torch

from magicdance_amd import synthetic
from oracle import vae_restatement as V
from tests import helpers as H

CASES = ["vae_small", "vae_full16", pytest.param("vae_full64", marks=pytest.mark.slow)]


@pytest.mark.parametrize("name", CASES)
def test_vae_restatement_matches_reference_golden(name):
    g = H.load_golden(name)
    dd = dict(V.SD15_DDCONFIG, ch=int(g["ch"]))
    sd = H.synth_vae_weights(int(g["ch"]), seed=int(g["seed"]))
    z, img = synthetic.synth_vae_inputs(int(g["side"]), int(g["batch"]), seed=int(g["seed"]))
    assert np.allclose(H.summarize(z), g["z_sum"]) and np.allclose(H.summarize(img), g["img_sum"])
    with torch.no_grad():
        dec = V.vae_decode(sd, H.VAE_PREFIX, z, dd)
        mom = V.vae_encode_moments(sd, H.VAE_PREFIX, img, dd)
    scale = float(g["dec_sum"][2])
    if "dec" in g:
        np.testing.assert_allclose(dec.numpy(), g["dec"], atol=2e-5 * max(1.0, scale), rtol=1e-4)
    else:
        np.testing.assert_allclose(dec[:, :, ::4, ::4].numpy(), g["dec_sub"], atol=2e-5 * max(1.0, scale), rtol=1e-4)
    np.testing.assert_allclose(H.summarize(dec), g["dec_sum"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(mom.numpy(), g["mom"], atol=2e-5 * max(1.0, float(g["mom_sum"][2])), rtol=1e-4)


def test_vae_container_has_the_reference_state_dict_layout():
    """magicdance_amd.autoencoder.AutoencoderKL must expose exactly the reference's keys / shapes (strict checkpoint load)."""
    g = H.load_golden("vae_full16")
    mine = "\n".join(f"{k}:{tuple(v.shape)}" for k, v in H.build_vae(128, "meta").state_dict().items())
    assert sorted(mine.split("\n")) == sorted(str(g["state_keys"]).split("\n"))
