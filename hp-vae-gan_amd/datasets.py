"""Data front-end on the device (the step before the hot path).

Mirrors the reference's dataset surface - `SingleVideoDataset` (datasets/video.py:13-93) and the image datasets
(datasets/image.py:13-61): `generate_frames(scale_idx)` at every stage change, `__len__`, `__getitem__` returning the
stage clip (and, above stage 0, the stage-0 clip) - but keeps the clip's full-resolution frames resident in HBM as uint8
and produces the normalised C,T,H,W tensor of a stage with ONE kernel (`hpvg_frames_resize_norm_u8_f32`): resize with
cv2's INTER_LINEAR geometry, temporal window, /255, optional hflip, normalize(0.5, 0.5), permute.

Not replicated: mp4 decoding (cv2.VideoCapture, datasets/generate_frames.py:15-41) - no decoder exists in this image.
Frames come from an in-memory uint8 array, a .npy file or a directory of image files (PIL).  Parity against cv2's
fixed-point resizer is unpinned (cv2 absent): single pixels may differ by one uint8 level."""
import os
import random

import numpy as np
import torch

from . import lib as hplib
from .utils import images as hp_images

__all__ = ["SingleVideoDataset", "SingleImageDataset", "load_frames"]


def load_frames(path):
    """[N, H, W, 3] uint8 RGB from a .npy file or a directory of image files (sorted by name)."""
    if os.path.isdir(path):
        from PIL import Image
        names = sorted(n for n in os.listdir(path) if n.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
        if not names:
            raise FileNotFoundError("no image files in %s" % path)
        return np.stack([np.asarray(Image.open(os.path.join(path, n)).convert("RGB")) for n in names])
    if path.endswith(".npy"):
        return np.load(path, allow_pickle=False)
    if path.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")):
        from PIL import Image
        return np.asarray(Image.open(path).convert("RGB"))[None]
    raise NotImplementedError("hp-vae-gan_amd.datasets: no video decoder in this build (cv2 absent); pass a directory of "
                              "frames, a .npy array [N,H,W,3] uint8 or opt.frames")


def _stage_size(opt, scale_idx):
    base = hp_images.get_scales_by_index(scale_idx, opt.scale_factor, opt.stop_scale, opt.img_size)
    return [int(base * opt.ar), base]  # datasets/video.py:85-86


class _DeviceFrames:
    def __init__(self, frames, device):
        frames = np.ascontiguousarray(frames)
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[-1] != 3:
            raise ValueError("frames must be uint8 [N, H, W, 3] RGB")
        self.N, self.H, self.W = frames.shape[:3]
        self.dev = torch.from_numpy(frames).to(device)

    def clip(self, first, step, count, h, w, hflip, quantize=True):
        out = torch.empty(3, count, h, w, dtype=torch.float32, device=self.dev.device)
        hplib.call("hpvg_frames_resize_norm_u8_f32", hplib.ptr(self.dev), hplib.ptr(out), self.N, self.H, self.W, first, step,
                   count, h, w, 1 if hflip else 0, 1 if quantize else 0, hplib.stream())
        return out


class SingleVideoDataset(torch.utils.data.Dataset):
    """datasets/video.py:13-93.  opt.frames (uint8 [N,H,W,3]) or opt.video_path (frame directory / .npy);
    opt.start_frame / opt.max_frames trim the clip as video_to_frames does (generate_frames.py:20-25)."""

    def __init__(self, opt, transforms=None, device=None):
        frames = getattr(opt, "frames", None)
        if frames is None:
            if not os.path.exists(opt.video_path):
                raise FileNotFoundError("invalid path: %s" % opt.video_path)
            frames = load_frames(opt.video_path)
        start = getattr(opt, "start_frame", 0)
        assert len(frames) > start >= 0, "Start-Frame out of range"
        frames = frames[start:start + opt.max_frames] if getattr(opt, "max_frames", None) else frames[start:]
        self.device = torch.device(device if device is not None else getattr(opt, "device", "cuda"))
        self.store = _DeviceFrames(frames, self.device)
        if not hasattr(opt, "org_fps") or opt.org_fps is None:
            opt.org_fps = 24.0  # cv2.CAP_PROP_FPS of the source; a frame directory has none
        opt.ar = self.store.H / self.store.W  # H2W
        opt.fps_lcm = int(np.lcm.reduce(opt.sampling_rates))
        self.opt = opt
        self.transforms = transforms
        self.size0 = _stage_size(opt, 0)
        self.size = self.size0

    def __len__(self):
        return (self.store.N - self.opt.fps_lcm) * self.opt.data_rep

    def generate_frames(self, scale_idx):
        """Stage change (train_video.py:364-366): only the target size changes; the resize happens per __getitem__."""
        self.size = _stage_size(self.opt, scale_idx)
        self.opt.scaled_size = self.size

    def _clip(self, idx, every, size, hflip):
        count = len(range(idx, idx + self.opt.fps_lcm + 1, every))
        return self.store.clip(idx, every, count, size[0], size[1], hflip)

    def __getitem__(self, idx):
        idx = idx % (self.store.N - self.opt.fps_lcm)
        hflip = random.random() < 0.5 if self.opt.hflip else False
        frames = self._clip(idx, self.opt.sampling_rates[self.opt.fps_index], self.size, hflip)
        if self.opt.scale_idx > 0:
            return [frames, self._clip(idx, self.opt.sampling_rates[0], self.size0, hflip)]
        return frames


class SingleImageDataset(torch.utils.data.Dataset):
    """datasets/image.py:13-61 with one image (SingleImageDataset / MultipleImageDataset hold a list)."""

    def __init__(self, opt, transforms=None, device=None):
        frames = getattr(opt, "frames", None)
        if frames is None:
            frames = load_frames(opt.image_path)
        self.device = torch.device(device if device is not None else getattr(opt, "device", "cuda"))
        self.store = _DeviceFrames(frames, self.device)
        opt.ar = self.store.H / self.store.W
        self.opt = opt
        self.transforms = transforms

    def __len__(self):
        return self.store.N * getattr(self.opt, "data_rep", 1)

    def _image(self, idx, scale_idx, hflip):
        size = _stage_size(self.opt, scale_idx)
        self.opt.scaled_size = size
        return self.store.clip(idx % self.store.N, 1, 1, size[0], size[1], hflip)[:, 0]

    def __getitem__(self, idx):
        hflip = random.random() < 0.5 if self.opt.hflip else False
        img = self._image(idx, self.opt.scale_idx, hflip)
        if self.opt.scale_idx > 0:
            return [img, self._image(idx, 0, hflip)]
        return img
