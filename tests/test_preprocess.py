"""CPU: the entry points' host-side data path -- image preprocessing against the torchvision restatement
(oracle/preprocess_restatement.py), the TikTok validation loader (magicdance_amd/tiktok.py vs dataset/tiktok_video_arnold_copy.py:
217-280), contiguous frame sharding."""
import os
import random

import numpy as np
import pytest
import torch

from magicdance_amd import entry, tiktok
from oracle import preprocess_restatement as P


def _png(path, h, w, seed, mode="RGB", flat=None):
    from PIL import Image
    rs = np.random.RandomState(seed)
    arr = (rs.rand(h, w, 4 if mode == "RGBA" else (1 if mode == "L" else 3)) * 255).astype(np.uint8)
    if flat is not None:
        arr[...] = flat
    Image.fromarray(arr.squeeze() if mode == "L" else arr, mode).save(path)


@pytest.mark.parametrize("h,w,mode", [(600, 500, "RGB"), (512, 512, "RGB"), (300, 777, "RGB"), (640, 360, "RGBA"), (129, 200, "L")])
def test_load_square_matches_torchvision_restatement(tmp_path, h, w, mode):
    p = str(tmp_path / "x.png")
    _png(p, h, w, seed=h + w, mode=mode)
    for normalize in (True, False):
        for seed in (0, 1, 2):   # the random draws of get_params never change the outcome with scale = ratio = (1, 1)
            want = P.preprocess(p, normalize, 512, rng=random.Random(seed))
            got = entry.load_square(p, normalize, 512)
            assert got.shape == (3, 512, 512) and torch.equal(got, want)
    assert torch.equal(entry.load_square(p, False, 768), P.preprocess(p, False, 768))


@pytest.mark.skipif(not os.path.isdir("/root/reference/example_data"), reason="reference example data lives in the build container only")
def test_load_square_on_the_reference_example_data():
    root = "/root/reference/example_data"
    n = 0
    for dp, _, files in os.walk(root):
        for f in sorted(files)[:3]:
            if f.lower().endswith((".png", ".jpg", ".jpeg")):
                p = os.path.join(dp, f)
                assert torch.equal(entry.load_square(p, "image" in dp, 512), P.preprocess(p, "image" in dp, 512)), p
                n += 1
    assert n >= 4


def _dataset_tree(root):
    data, pose = root / "frames", root / "poses"
    for s, n in (("b_subject", 4), ("a_subject", 3), ("c_mono", 3)):
        os.makedirs(data / s)
        os.makedirs(pose / s)
        for i in range(n):
            _png(str(data / s / f"{i:04d}.png"), 96, 80, seed=hash((s, i)) % 1000, flat=(7 if (s == "c_mono" and i == 0) or (s == "b_subject" and i == 2) else None))
            _png(str(pose / s / f"{i:04d}.png"), 96, 80, seed=1000 + i)
    return str(data), str(pose)


def test_tiktok_val_loader(tmp_path):
    data, pose = _dataset_tree(tmp_path)
    ds = tiktok.tiktok_video_arnold_val(data, pose, rank=0, world_size=1, img_bin_limit="all", image_size=64)
    items = list(ds)
    # sorted subject order; the subject whose reference frame is monochromatic is skipped (tiktok_video_arnold_copy.py:224-229)
    assert [it["subject"] for it in items] == ["a_subject", "b_subject"]
    a, b = items
    assert a["condition_image"].shape == (3, 64, 64) and -1.0 <= float(a["condition_image"].min()) and float(a["condition_image"].max()) <= 1.0
    assert 0.0 <= float(a["src_pose_map"].min()) and float(a["src_pose_map"].max()) <= 1.0
    assert len(a["image_list"]) == len(a["pose_map_list"]) == 2          # frames 1, 2 of 3
    # b_subject: frame 2 is flat -> dropped together with its pose map (:259-265): frames 1 and 3 remain
    assert len(b["image_list"]) == len(b["pose_map_list"]) == 2
    want_pose3 = entry.load_square(os.path.join(pose, "b_subject", "0003.png"), False, 64)
    assert torch.equal(b["pose_map_list"][1], want_pose3)
    # img_bin_limit counts images of the folder including the reference frame (:250-253)
    lim = list(tiktok.tiktok_video_arnold_val(data, pose, img_bin_limit=2, image_size=64))
    assert [len(it["image_list"]) for it in lim] == [1, 1]
    # contiguous frame blocks per rank, sizes differ by at most one, every frame exactly once
    for world in (1, 2, 3, 8):
        blocks = [tiktok.TikTokValDataset(data, pose, rank=r, world_size=world, image_size=64).shard_frames(7) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == 7 and all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        assert max(b1 - b0 for b0, b1 in blocks) - min(b1 - b0 for b0, b1 in blocks) <= 1


def test_monochromatic_rule():
    from PIL import Image
    assert tiktok.is_monochromatic_image(Image.fromarray(np.full((8, 8, 3), 200, np.uint8)))
    assert not tiktok.is_monochromatic_image(Image.fromarray((np.random.RandomState(0).rand(8, 8, 3) * 255).astype(np.uint8)))
