"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the image preprocessing of the reference entry points
(test_any_image_pose.py:46-81 center_crop_to_512 / center_crop_pose_to_512; test_tiktok.py:441-459 test_image_transform /
test_pose_transform), which is torchvision code -- a third-party dependency absent from /root/reference and from this image
(environment.yml pins torchvision==0.14.1): ``T.RandomResizedCrop(size, scale=(1.0, 1.0), ratio=(1., 1.), BILINEAR)`` +
``T.ToTensor()`` (+ ``T.Normalize(0.5, 0.5)``).  Published algorithm restated here step by step:

  RandomResizedCrop.get_params (torchvision/transforms/transforms.py, v0.14): area = H*W; up to 10 attempts:
      target_area = area * U(scale); aspect = exp(U(log ratio)); w = round(sqrt(target_area * aspect)); h = round(sqrt(target_area / aspect));
      if 0 < w <= W and 0 < h <= H: i = randint(0, H - h + 1), j = randint(0, W - w + 1); return (i, j, h, w)
    fallback: in_ratio = W / H; if in_ratio < min(ratio): w = W, h = round(w / min(ratio)); elif in_ratio > max(ratio): h = H,
      w = round(h * max(ratio)); else w = W, h = H;  i = (H - h) // 2, j = (W - w) // 2
  F.resized_crop on a PIL image: img.crop((j, i, j + w, i + h)).resize((size, size), PIL.Image.BILINEAR)
  ToTensor: uint8 HWC -> float CHW / 255;  Normalize(mean, std): (x - mean) / std
With scale = ratio = (1, 1) the attempt succeeds only for a square image (w = h = round(sqrt(H W)) = side, i = j = 0); otherwise
the fallback is the centred square crop.  Parity is "unpinned" by reference tests (there are none); the product's
magicdance_amd.entry.load_square is checked against this restatement bit for bit (tests/test_preprocess.py).
"""
import math
import random

import numpy as np
import torch


def get_params(height, width, scale=(1.0, 1.0), ratio=(1.0, 1.0), rng=None):
    rng = rng or random.Random(0)
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect_ratio = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = rng.randint(0, height - h)        # torch.randint(0, height - h + 1): upper bound exclusive
            j = rng.randint(0, width - w)
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def preprocess(path, normalize, size=512, rng=None):
    from PIL import Image
    img = Image.open(path)
    if img.mode != "RGB":
        img = img.convert("RGB")                               # T.Lambda(lambda img: img.convert('RGB') if img.mode != 'RGB' else img)
    width, height = img.size
    i, j, h, w = get_params(height, width, rng=rng)
    img = img.crop((j, i, j + w, i + h)).resize((size, size), Image.BILINEAR)
    t = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).float().div(255)
    if normalize:
        t = (t - 0.5) / 0.5
    return t
