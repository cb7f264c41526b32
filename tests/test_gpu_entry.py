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
    assert os.path.exists(out / "0" / "condition.jpg") and not os.path.exists(out / "0" / "gt_images")
    img = np.asarray(Image.open(out / "0" / "gen_images" / "000.jpg"))
    assert img.shape == (512, 512, 3) and img.std() > 0
    z = torch.load(out / "0" / "latents" / "000.pt")
    assert tuple(z.shape) == (1, 4, 64, 64) and bool(torch.isfinite(z).all())


def test_tiktok_entry_point_dataset_flow(tmp_path):
    """test_tiktok.py without --local_* overrides: subjects come from the validation loader (magicdance_amd/tiktok.py), frame 0 is
    the reference image, the other frames are rendered; gt_images holds the VAE round trip of the ground-truth frames
    (test_tiktok.py:273-279), condition.jpg the reference image."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from PIL import Image
    rs = np.random.RandomState(1)
    for s, n in (("s0", 3), ("s1", 2)):
        os.makedirs(tmp_path / "frames" / s)
        os.makedirs(tmp_path / "poses" / s)
        for i in range(n):
            Image.fromarray((rs.rand(540, 512, 3) * 255).astype(np.uint8)).save(tmp_path / "frames" / s / f"{i:04d}.png")
            Image.fromarray((rs.rand(512, 512, 3) * 255).astype(np.uint8)).save(tmp_path / "poses" / s / f"{i:04d}.png")
    torch.save(torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(3)), tmp_path / "ctx.pt")
    out = tmp_path / "out"
    cmd = [sys.executable, os.path.join(H.ROOT, "test_tiktok.py"), "--model_config",
           os.path.join(H.ROOT, "magicdance_amd/configs/cldm_v15_reference_only_pose.yaml"), "--num_train_steps", "2",
           "--img_bin_limit", "all", "--train_batch_size", "1", "--use_fp16", "--control_mode", "controlnet_important",
           "--control_type", "body+hand+face", "--train_dataset", "tiktok_video_arnold", "--v4", "--with_text", "--wonoise",
           "--local_image_dir", str(out), "--local_log_dir", str(tmp_path / "log"), "--synthetic_weights", "--ddim_steps", "4",
           "--context_embedding", str(tmp_path / "ctx.pt"), "--frames_per_batch", "2",
           "--tiktok_data_path", str(tmp_path / "frames"), "--tiktok_pose_path", str(tmp_path / "poses")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=H.ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for itr, nfr in ((0, 2), (1, 1)):
        assert os.path.exists(out / str(itr) / "condition.jpg")
        for sub in ("gen_images", "gt_images", "pose_maps"):
            assert sorted(os.listdir(out / str(itr) / sub)) == [f"{i:03d}.jpg" for i in range(nfr)], (itr, sub)
    gt = np.asarray(Image.open(out / "0" / "gt_images" / "000.jpg")).astype(np.float32)
    assert gt.shape == (512, 512, 3) and gt.std() > 0
    cond = np.asarray(Image.open(out / "0" / "condition.jpg")).astype(np.float32)
    src = np.asarray(Image.open(tmp_path / "frames" / "s0" / "0000.png").crop((0, 14, 512, 526)).resize((512, 512), Image.BILINEAR)).astype(np.float32)
    blk = lambda a: a.reshape(16, 32, 16, 32, 3).mean((1, 3))  # noqa: E731 -- JPEG of a noise image is lossy per pixel, not per block
    assert np.abs(blk(cond) - blk(src)).max() < 6.0              # = the centre-cropped, resized reference frame


def test_image_to_u8_kernel():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from magicdance_amd import ops
    dev = torch.device("cuda:0")
    x = (torch.randn(2, 3, 40, 24, generator=torch.Generator().manual_seed(0)) * 0.8).to(dev)
    out = torch.empty((2, 40, 24, 3), dtype=torch.uint8, device=dev)
    ops.image_to_u8(x, out, 2, 3, 40 * 24, 0.5, 0.5)
    want = ((x.clamp(-1, 1) + 1) * 0.5 * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert int((out.int() - want.int()).abs().max()) <= 1 and float((out != want).float().mean()) < 1e-3
