"""Synthetic Ball-shaped multi-camera rigs and the `sys_param` dict for benchmarks / smoke tests.

No dataset ships with the reference (README.md:24 points to Google Drive) and there is no network, so
the benchmark workload is generated: camera placement follows the reference's Blender generator for the
Ball scenes (synthetic_dataset_code/Ball.py:23-25, 116-128, 146-153, 163-190: radius 3, theta in
linspace(0,360,12) x phi in linspace(-80,80,9) plus the two poles = 110 cameras, Euler XYZ
(90deg - phi, 0, theta), integer FOV in [40,80] from random.seed(seed)), converted to the reference's
world->cam convention exactly like data/data_read.py:141-152 (fov -> K) and :246-257 (flip y,z; invert).
Images are uniform noise: throughput does not depend on them.
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def _rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def ball_cameras(seed: int = 0, radius: float = 3.0, H: int = 800, W: int = 800):
    """-> pose [110,3,4] (world->cam, reference convention), K [110,3,3], fov_deg [110]."""
    thetas = list(np.linspace(0, 360, 12, endpoint=False))
    phis = list(np.linspace(-80, 80, 9))
    grid = [(th, ph) for ph in phis for th in thetas] + [(0.0, -90.0), (0.0, 90.0)]
    rnd = random.Random(seed)
    fovs = [rnd.randint(40, 80) for _ in grid]
    start = np.array([0.0, -radius, 0.0])
    poses, Ks = [], []
    for (th, ph), fov in zip(grid, fovs):
        # location: start_pos . Rot_X(phi) . Rot_Z(theta) with the generator's row-vector matrices
        rp, rt = -math.radians(ph), math.radians(th)
        rx = np.array([[1, 0, 0], [0, math.cos(rp), math.sin(rp)], [0, -math.sin(rp), math.cos(rp)]])
        rz = np.array([[math.cos(rt), math.sin(rt), 0], [-math.sin(rt), math.cos(rt), 0], [0, 0, 1]])
        loc = start @ rx @ rz
        # Blender Euler XYZ (90deg - phi, 0, theta): camera-to-world rotation Rz(theta) Rx(90deg - phi)
        R_c2w = _rot_z(rt) @ _rot_x(math.radians(90.0) - math.radians(ph))
        R_new = R_c2w @ np.diag([1.0, -1.0, -1.0])          # blender -> reference camera axes
        R_w2c = R_new.T
        t_w2c = -R_w2c @ loc.reshape(3, 1)
        poses.append(np.concatenate([R_w2c, t_w2c], axis=1))
        f = math.radians(fov)
        Ks.append(np.array([[(W / 2) / math.tan(f / 2), 0, W / 2], [0, (H / 2) / math.tan(f / 2), H / 2], [0, 0, 1]]))
    return (torch.tensor(np.stack(poses), dtype=torch.float32), torch.tensor(np.stack(Ks), dtype=torch.float32),
            np.array(fovs))


def se3_log(pose: torch.Tensor) -> torch.Tensor:
    """[C,3,4] -> [C,6] (w,u) such that the model's se3_to_SE3(wu) reproduces the pose (closed-form A,B,C)."""
    out = []
    for P in pose.double():
        R, t = P[:, :3], P[:, 3]
        cos = max(-1.0, min(1.0, (float(R.trace()) - 1.0) / 2.0))
        th = math.acos(cos)
        if th < 1e-6:
            w = torch.zeros(3, dtype=torch.float64)
        elif math.pi - th < 1e-4:
            # near pi the antisymmetric part vanishes: R ~ 2 a a^T - I, take the axis from the symmetric part
            M = (R + torch.eye(3, dtype=torch.float64)) / 2.0
            k = int(torch.argmax(torch.diagonal(M)))
            a = M[:, k] / math.sqrt(float(M[k, k]))
            w = th * a / a.norm()
        else:
            w = th / (2.0 * math.sin(th)) * torch.stack([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        wx = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
        if th < 1e-6:
            V = torch.eye(3, dtype=torch.float64)
        else:
            B, C = (1 - math.cos(th)) / th ** 2, (th - math.sin(th)) / th ** 3
            V = torch.eye(3, dtype=torch.float64) + B * wx + C * wx @ wx
        out.append(torch.cat([w, torch.linalg.solve(V, t)]))
    return torch.stack(out).float()


def calibration_points(pose: torch.Tensor, K: torch.Tensor, seed: int = 0):
    """Five world points per camera (tag centre + corners on a unit cube face) and their GT pixel
    projections: the inputs of the reprojection branch (data/data_read.py:72-75 layout)."""
    g = torch.Generator().manual_seed(seed)
    C = pose.shape[0]
    base = torch.tensor([[0, 0, 0.5], [-0.4, -0.4, 0.5], [0.4, -0.4, 0.5], [0.4, 0.4, 0.5], [-0.4, 0.4, 0.5]])
    wpts = base.unsqueeze(0).repeat(C, 1, 1) + 0.01 * torch.randn(C, 5, 3, generator=g)
    cam = torch.cat([wpts, torch.ones(C, 5, 1)], -1) @ pose.transpose(-2, -1)
    pix = cam @ K.transpose(-2, -1)
    return wpts.unsqueeze(0), (pix[..., :2] / pix[..., 2:]).unsqueeze(0)


def make_sys_param(device, *, samples=64, scale=2, batch=7000, H=800, W=800, coarse=(4, 128, [2]), fine=(8, 256, [4]),
                   mode=0, barf_mask=False, barf_start=0.3846, barf_end=0.6923, seed=0, n_val_images=1, **extra):
    """`sys_param` for a Ball_Lego-shaped synthetic scene (the keys listed in SURVEY.md 8b)."""
    pose, K, _ = ball_cameras(seed, H=H, W=W)
    C = pose.shape[0]
    Kinv = torch.linalg.inv(K)
    g = torch.Generator().manual_seed(seed + 1)
    p = dict(mode=mode, device_type=str(device), batch=batch, near=1.0, far=8.0, samples=samples, scale=scale,
             MLP_deg=2, white_back=True, root_weight="./weights", demo_render_pth="./results", boader_min=-3.5,
             boader_max=3.5, grid_nerf=384, sigma_init=30.0, sigma_default=-20.0, warmup_epoch=100,
             sample_weight_thresh=1e-3, res_h=H, res_w=W, data_name="ball_lego_synth", emb_freqs_xyz=10,
             barf_mask=barf_mask, barf_start=barf_start, barf_end=barf_end,
             coarse_MLP_depth=coarse[0], coarse_MLP_width=coarse[1], coarse_MLP_skip=list(coarse[2]),
             fine_MLP_depth=fine[0], fine_MLP_width=fine[1], fine_MLP_skip=list(fine[2]), distributed=False,
             intr_mat=[K, K, K], intr_mat_inv=[Kinv, Kinv, Kinv], gt_pose=pose, test_pose=pose, valid_pose=pose,
             valid_rgbs=torch.rand(n_val_images, H * W, 3, generator=g).expand(C, -1, -1) if n_val_images == 1
             else torch.rand(C, H * W, 3, generator=g),
             data_img_h=H, data_img_w=W, data_numb=[C, C, C], train_json_file="")
    p.update(extra)
    return p


def init_cameras_near_gt(model, noise: float = 0.0, seed: int = 0):
    """Puts the learnable camera parameters at (a perturbation of) the ground truth, i.e. the state the
    reference reaches after its camera-initialisation stage."""
    g = torch.Generator().manual_seed(seed)
    K, pose = model.intr_train, model.gt_pose.cpu()
    W, H = float(model.img_w), float(model.img_h)
    with torch.no_grad():
        model.weights_pose.copy_((se3_log(pose) + noise * torch.randn(pose.shape[0], 6, generator=g)).to(model.weights_pose.device))
        model.weights_pose_intr.copy_(model.weights_pose)
        model.weights_fx.copy_((K[:, 0, 0] / W).to(model.weights_fx.device))
        model.weights_fy.copy_((K[:, 1, 1] / W).to(model.weights_fy.device))
        model.weights_ux.copy_((K[:, 0, 2] / (W / 2)).to(model.weights_ux.device))
        model.weights_uy.copy_((K[:, 1, 2] / (H / 2)).to(model.weights_uy.device))
