"""Validation-set loader of the reference's ``test_tiktok.py`` path: the inference branch of
``ImageTextControlDataset.__iter__`` (dataset/tiktok_video_arnold_copy.py:217-280) and the dataset factory
``tiktok_video_arnold_val`` (:293-296), without its torchvision / langdetect / annotator imports.

Layout on disk (reference defaults ./TikTok-v4/disco_test_set and ./TikTok-v4/pose_map_disco_test_set): one folder per
subject in ``data_path`` holding the frames, a folder of the same name in ``pose_path`` holding the pose maps; both listings
are sorted and paired by position.  One item per subject:
  condition_image [3,512,512] in [-1,1]   frame 0 (the reference image)
  src_pose_map    [3,512,512] in [0,1]    pose map 0
  image_list      frames 1.. (ground truth, [-1,1]) and  pose_map_list  pose maps 1..  -- a frame that is monochromatic
                  (ImageStat variance sum < 0.3) or nearly constant (std < 0.02) is dropped together with its pose map
  limited to ``img_bin_limit`` images per subject (``'all'`` = no limit), as the reference does.
Differences, on purpose: subjects are visited in SORTED order (the reference shuffles them with the global ``random`` state,
:143) so that all ranks of a sharded run see the same subject at the same iteration, and ``shard_frames`` gives every rank a
contiguous block of a subject's frames (the reference passes rank / world_size into the dataset but never uses them: every
rank renders every frame, :128-131).  Images are decoded on the host by PIL; preprocessing = entry._load_square_512, the exact
PIL path of the reference's torchvision RandomResizedCrop(scale=(1,1), ratio=(1,1), BILINEAR) + ToTensor (+ Normalize).
"""
import os

import torch

MONOCHROMATIC_MAX_VARIANCE = 0.3    # dataset/tiktok_video_arnold_copy.py:19


def is_monochromatic_image(pil_img):
    """tiktok_video_arnold_copy.py:51-53"""
    from PIL import ImageStat
    return sum(ImageStat.Stat(pil_img.convert("RGB")).var) < MONOCHROMATIC_MAX_VARIANCE


class TikTokValDataset:
    def __init__(self, data_path, pose_path, rank=0, world_size=1, img_bin_limit="all", image_size=512):
        assert len(data_path) > 0 and len(pose_path) > 0, "Data / pose path must not be empty."
        assert 0 <= rank < world_size, "Rank must be >= 0 and < world_size."
        self.data_path, self.pose_path, self.rank, self.world_size = data_path, pose_path, rank, world_size
        self.img_bin_limit, self.image_size = img_bin_limit, image_size
        self.all_subjects = sorted(os.listdir(self.data_path))

    def __len__(self):
        return len(self.all_subjects)

    def _load(self, path, normalize):
        from .entry import load_square
        return load_square(path, normalize, self.image_size, return_pil=True)

    def __iter__(self):
        for subject in self.all_subjects:
            folder, pose_folder = os.path.join(self.data_path, subject), os.path.join(self.pose_path, subject)
            images, poses = sorted(os.listdir(folder)), sorted(os.listdir(pose_folder))
            cond, cond_pil = self._load(os.path.join(folder, images[0]), True)
            if is_monochromatic_image(cond_pil) or float(cond.std()) < 0.02:
                continue                                                              # :224-229
            res = {"subject": subject, "condition_image": cond,
                   "src_pose_map": self._load(os.path.join(pose_folder, poses[0]), False)[0]}
            n = len(images) if self.img_bin_limit == "all" else min(int(self.img_bin_limit), len(images))   # :250-253
            image_list, pose_map_list = [], []
            for i in range(n - 1):
                img, pil = self._load(os.path.join(folder, images[i + 1]), True)
                if is_monochromatic_image(pil) or float(img.std()) < 0.02:
                    continue                                                          # :259-265
                image_list.append(img)
                pose_map_list.append(self._load(os.path.join(pose_folder, poses[i + 1]), False)[0])
            res["image_list"], res["pose_map_list"] = image_list, pose_map_list
            yield res

    def shard_frames(self, n_frames):
        """contiguous block [f0, f1) of a subject's frames rendered by this rank (sizes differ by at most one)"""
        from .parallel import FrameShardedSampler
        return FrameShardedSampler.frame_block(n_frames, self.rank, self.world_size)


def tiktok_video_arnold_val(data_path="./TikTok-v4/disco_test_set", pose_path="./TikTok-v4/pose_map_disco_test_set", **kwargs):
    """tiktok_video_arnold_copy.py:293-296"""
    return TikTokValDataset(data_path, pose_path, **kwargs)


class AsyncImageWriter:
    """Decoded frames -> JPG files without stalling the sampling stream: the uint8 NHWC conversion runs on the GPU
    (md_image_to_u8), the device -> pinned-host copy is asynchronous on a side stream, and the JPEG encode + file write happen on a
    worker thread once the copy's event has completed."""

    def __init__(self, device):
        import queue
        import threading
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.q = queue.Queue()
        self.err = None
        self.th = threading.Thread(target=self._work, daemon=True)
        self.th.start()

    def _work(self):
        from PIL import Image
        while True:
            item = self.q.get()
            if item is None:
                return
            host, ev, paths = item
            try:
                ev.synchronize()
                arr = host.numpy()
                for i, p in enumerate(paths):
                    Image.fromarray(arr[i]).save(p)
            except Exception as e:  # noqa: BLE001 -- reported by close()
                self.err = e

    def save(self, images, paths, value_range=(-1.0, 1.0)):
        """images: NCHW fp32 [B,3,H,W] on the device with values in ``value_range``; paths: B file names."""
        from . import ops
        b, c, h, w = images.shape
        assert c == 3 and len(paths) == b
        cur = torch.cuda.current_stream()
        u8 = torch.empty((b, h, w, 3), dtype=torch.uint8, device=images.device)
        lo, hi = value_range
        ops.image_to_u8(images.contiguous(), u8, b, 3, h * w, 1.0 / (hi - lo), -lo / (hi - lo))
        self.stream.wait_stream(cur)
        host = torch.empty((b, h, w, 3), dtype=torch.uint8, pin_memory=True)
        with torch.cuda.stream(self.stream):
            host.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        u8.record_stream(self.stream)
        self.q.put((host, ev, list(paths)))

    def close(self):
        self.q.put(None)
        self.th.join()
        if self.err is not None:
            raise self.err
