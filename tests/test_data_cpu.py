"""Blender-format loader: on-disk format of the reference kept, conventions converted like
data/data_read.py:141-152 (fov -> K) and :246-257 (blender c2w -> reference w2c)."""
import json
import math
import os

import numpy as np
import torch


def test_load_blender_split(tmp_path):
    from PIL import Image
    from mc_nerf_amd.data import load_blender_split, DeviceImageSet
    rs = np.random.RandomState(0)
    H, W = 6, 8
    frames = []
    os.makedirs(tmp_path / "train")
    c2ws = []
    for i in range(3):
        a, b = rs.uniform(0, 2 * math.pi, 2)
        Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        c2w = np.eye(4)
        c2w[:3, :3] = Rz @ Rx
        c2w[:3, 3] = rs.uniform(-3, 3, 3)
        c2ws.append(c2w)
        Image.fromarray(rs.randint(0, 256, (H, W, 4), dtype=np.uint8), "RGBA").save(tmp_path / "train" / f"r_{i}.png")
        frames.append({"file_path": f"./train/r_{i}", "camera_angle_x": math.radians(40 + 10 * i),
                       "transform_matrix": c2w.tolist()})
    json.dump({"frames": frames}, open(tmp_path / "transforms_train.json", "w"))
    d = load_blender_split(str(tmp_path), "train", "cpu")
    assert d["H"] == H and d["W"] == W and len(d["images"]) == 3 and d["images"].images.shape == (3, H * W, 4)
    for i, c2w in enumerate(c2ws):
        # reference convention: world->cam of the camera whose y,z axes are flipped
        flip = np.eye(4); flip[1, 1] = flip[2, 2] = -1
        w2c = np.linalg.inv(c2w @ flip)[:3]
        assert np.abs(d["pose"][i].numpy() - w2c).max() < 1e-5
        f = math.radians(40 + 10 * i)
        assert abs(float(d["K"][i, 0, 0]) - (W / 2) / math.tan(f / 2)) < 1e-3
        assert abs(float(d["K"][i, 1, 1]) - (H / 2) / math.tan(f / 2)) < 1e-3
        assert float(d["K"][i, 0, 2]) == W / 2 and float(d["K"][i, 1, 2]) == H / 2
        img = np.asarray(Image.open(tmp_path / "train" / f"r_{i}.png")).reshape(H * W, 4)
        assert np.array_equal(d["images"].images[i].numpy(), img)
    s = DeviceImageSet.synthetic(10, 4, 4, "cpu")
    assert len(s) == 10 and torch.equal(s.images[0], s.images[4]) and not torch.equal(s.images[0], s.images[1])
