"""GPU tier: the entry-point script end to end (reference CLI flags -> PNG inputs -> HIP VAE encode -> sampling -> HIP VAE
decode -> JPG tree), seeded synthetic weights, 4 DDIM steps, tiny frames count."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_any_image_pose_entry_point_writes_the_reference_output_tree(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from PIL import Image
    rs = np.random.RandomState(0)
    Image.fromarray((rs.rand(600, 500, 3) * 255).astype(np.uint8)).save(tmp_path / "ref.png")
    os.makedirs(tmp_path / "poses")
    for i in range(2):
        Image.fromarray((rs.rand(512, 512, 3) * 255).astype(np.uint8)).save(tmp_path / "poses" / f"{i:04d}.png")
    ctx = torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(3))
    torch.save(ctx, tmp_path / "ctx.pt")
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(H.ROOT, "test_any_image_pose.py"), "--model_config",
           os.path.join(H.ROOT, "magicdance_amd/configs/cldm_v15_reference_only_pose.yaml"), "--num_train_steps", "1",
           "--img_bin_limit", "all", "--train_batch_size", "1", "--use_fp16", "--control_mode", "controlnet_important",
           "--control_type", "body+hand+face", "--train_dataset", "tiktok_video_arnold", "--v4", "--with_text", "--wonoise",
           "--local_image_dir", str(out), "--local_log_dir", str(tmp_path / "log"), "--local_pose_path", str(tmp_path / "poses"),
           "--local_cond_image_path", str(tmp_path / "ref.png"), "--synthetic_weights", "--ddim_steps", "4",
           "--context_embedding", str(tmp_path / "ctx.pt"), "--frames_per_batch", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=H.ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for sub, ext in (("gen_images", "jpg"), ("pose_maps", "jpg"), ("latents", "pt")):
        for i in range(2):
            assert os.path.exists(out / "0" / sub / f"{i:03d}.{ext}"), (sub, i, r.stdout[-1000:])
    img = np.asarray(Image.open(out / "0" / "gen_images" / "000.jpg"))
    assert img.shape == (512, 512, 3) and img.std() > 0
    z = torch.load(out / "0" / "latents" / "000.pt")
    assert tuple(z.shape) == (1, 4, 64, 64) and bool(torch.isfinite(z).all())
