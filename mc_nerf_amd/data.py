"""Device-resident image sets.

The reference materialises 50 float copies of every image on the host (`Data_set.expand_data_length`,
data/data_read.py:286-297), ships one 7.7 MB float image per step through a 12-worker DataLoader and gathers
the batch's ground truth with an advanced index (model/mc_nerf.py:379, 80).  Here the images stay in HBM as
uint8 (110 x 800 x 800 x 4 B = 282 MB) and a kernel gathers + converts only the selected pixels.

The on-disk format is the reference's: Blender `transforms_{split}.json` + RGBA PNGs
(data/data_read.py:84-139); poses / intrinsics are converted exactly like `blender_pose_transform` (:246-257)
and `blender_fov_to_intrinsic` (:141-152).  AprilTag detection (cv2 / apriltag) is out of scope.
"""
from __future__ import annotations

import json
import math
import os
from typing import List, Optional

import numpy as np
import torch

from . import ops


class DeviceImageSet:
    """uint8 images [C, H*W, channels] on the device; `gather(cam, pix)` -> fp32 [n,3]."""

    def __init__(self, images_u8: torch.Tensor, H: int, W: int):
        assert images_u8.dtype == torch.uint8 and images_u8.dim() == 3 and images_u8.shape[1] == H * W
        assert images_u8.shape[2] in (3, 4)
        self.images = images_u8.contiguous()
        self.H, self.W = H, W

    def __len__(self):
        return self.images.shape[0]

    @property
    def device(self):
        return self.images.device

    def gather(self, cam: int, pix: torch.Tensor) -> torch.Tensor:
        return ops.gather_gt(self.images[cam], pix.contiguous())

    def full_image(self, cam: int) -> torch.Tensor:
        """[H*W,3] fp32 of one camera (validation / metrics)."""
        return self.gather(cam, torch.arange(self.H * self.W, device=self.device))

    @staticmethod
    def synthetic(C: int, H: int, W: int, device, channels: int = 4, seed: int = 0, distinct: int = 4) -> "DeviceImageSet":
        """Random images (only `distinct` different ones are generated; throughput does not depend on content)."""
        g = torch.Generator().manual_seed(seed)
        base = torch.randint(0, 256, (distinct, H * W, channels), dtype=torch.uint8, generator=g)
        idx = torch.arange(C) % distinct
        return DeviceImageSet(base[idx].to(device), H, W)


def blender_pose_to_reference(c2w: np.ndarray) -> np.ndarray:
    """Blender camera-to-world [4,4] -> the reference's world->cam [3,4] (flip y,z; invert)."""
    R = c2w[:3, :3] @ np.diag([1.0, -1.0, -1.0])
    t = c2w[:3, 3:]
    Rinv = R.T
    return np.concatenate([Rinv, -Rinv @ t], axis=1)


def load_blender_split(root: str, split: str, device, with_images: bool = True):
    """Reads transforms_{split}.json (+ PNGs) -> dict(pose [C,3,4], K [C,3,3], images DeviceImageSet | None, H, W)."""
    with open(os.path.join(root, f"transforms_{split}.json")) as f:
        meta = json.load(f)
    poses, fovs, imgs = [], [], []
    H = W = None
    for fr in meta["frames"]:
        poses.append(blender_pose_to_reference(np.asarray(fr["transform_matrix"], dtype=np.float64)))
        fovs.append(float(fr.get("camera_angle_x", meta.get("camera_angle_x"))))
        if with_images:
            from PIL import Image
            im = np.asarray(Image.open(os.path.join(root, fr["file_path"] + ".png")))
            if im.ndim == 2:
                im = np.stack([im] * 3, -1)
            H, W = im.shape[:2]
            imgs.append(torch.from_numpy(np.ascontiguousarray(im)).reshape(H * W, -1))
    if H is None:
        H, W = int(meta.get("h", 800)), int(meta.get("w", 800))
    K = np.stack([[[(W / 2) / math.tan(f / 2), 0, W / 2], [0, (H / 2) / math.tan(f / 2), H / 2], [0, 0, 1]] for f in fovs])
    out = dict(pose=torch.tensor(np.stack(poses), dtype=torch.float32), K=torch.tensor(K, dtype=torch.float32), H=H, W=W,
               images=DeviceImageSet(torch.stack(imgs).to(device), H, W) if with_images else None)
    return out
