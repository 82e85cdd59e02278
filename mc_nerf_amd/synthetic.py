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


def _pose_from_blender(loc, euler_xz, fov, H, W):
    """Blender camera (location, Euler XYZ = (rx, 0, rz), FOV in degrees) -> the reference's world->cam [R|t] and K
    (data/data_read.py:141-152 fov -> K, :246-257 flip y,z and invert)."""
    rx, rz = euler_xz
    R_c2w = _rot_z(rz) @ _rot_x(rx)
    R_w2c = (R_c2w @ np.diag([1.0, -1.0, -1.0])).T
    t_w2c = -R_w2c @ np.asarray(loc, dtype=np.float64).reshape(3, 1)
    f = math.radians(fov)
    K = np.array([[(W / 2) / math.tan(f / 2), 0, W / 2], [0, (H / 2) / math.tan(f / 2), H / 2], [0, 0, 1]])
    return np.concatenate([R_w2c, t_w2c], axis=1), K


def _rig_tensors(locs, eulers, fovs, H, W):
    P, Ks = zip(*[_pose_from_blender(l, e, f, H, W) for l, e, f in zip(locs, eulers, fovs)])
    return torch.tensor(np.stack(P), dtype=torch.float32), torch.tensor(np.stack(Ks), dtype=torch.float32), np.array(fovs)


def _gen_rot_x(deg):        # the generators' row-vector Rot_X (note the sign flip inside, Array.py:129-134)
    r = -math.radians(deg)
    return np.array([[1, 0, 0], [0, math.cos(r), math.sin(r)], [0, -math.sin(r), math.cos(r)]])


def _gen_rot_z(deg):        # Array.py:122-127
    r = math.radians(deg)
    return np.array([[math.cos(r), math.sin(r), 0], [-math.sin(r), math.cos(r), 0], [0, 0, 1]])


def array_cameras(seed: int = 0, H: int = 800, W: int = 800):
    """Array_* rig (BASELINE configs[2]; synthetic_dataset_code/Array.py:18-35, 176-191, 248-263): a 10 x 10 planar grid of
    3 x 3 units at distance 4, tilted 45 degrees about z after a 90-degree turn about x, every camera aimed at the origin
    by the generator's look-at (pitch from atan(z / r_xy), yaw from the xy direction); integer FOVs in [40, 80] from
    random.seed(seed).  -> pose [100,3,4], K [100,3,3], fov_deg [100]"""
    rnd = random.Random(seed)
    fovs = [rnd.randint(40, 80) for _ in range(100)]
    xs, ys = np.linspace(-1.5, 1.5, 10), np.linspace(-1.5, 1.5, 10)
    X, Y = np.meshgrid(xs, ys)
    cord = np.stack([X, Y, -4.0 * np.ones_like(X)], -1).reshape(-1, 3) @ _gen_rot_x(90) @ _gen_rot_z(45)
    eulers = []
    for loc in cord:
        r = np.linalg.norm(loc[:2])
        phi = math.atan(loc[2] / r)
        v = loc[:2] / r
        cos_t, sin_t = float(np.dot(v, [0.0, -1.0])), float(v[0] * -1.0 - v[1] * 0.0)        # (2-D cross product v x (0, -1))
        theta = 2 * math.pi - math.acos(max(-1.0, min(1.0, cos_t))) if sin_t > 0 else math.acos(max(-1.0, min(1.0, cos_t)))
        eulers.append((math.radians(90) - phi, theta))
    return _rig_tensors(cord, eulers, fovs, H, W)


def halfball_cameras(seed: int = 0, H: int = 800, W: int = 800, radius: float = 3.0):
    """HalfBall_* rig (BASELINE configs[3]; synthetic_dataset_code/HalfBall.py:14-26, 162-178): 100 cameras on the upper
    half sphere of radius 3 at integer angles theta in [0, 360], phi in [0, 90] from random.seed(seed) (FOVs first, as
    the generator draws them), Euler XYZ (90 deg - phi, 0, theta)."""
    rnd = random.Random(seed)
    fovs = [rnd.randint(40, 80) for _ in range(100)]
    start = np.array([0.0, -radius, 0.0])
    locs, eulers = [], []
    for _ in range(100):
        theta, phi = rnd.randint(0, 360), rnd.randint(0, 90)
        locs.append(start @ _gen_rot_x(phi) @ _gen_rot_z(theta))
        eulers.append((math.radians(90) - math.radians(phi), math.radians(theta)))
    return _rig_tensors(locs, eulers, fovs, H, W)


def room_cameras(seed: int = 0, H: int = 800, W: int = 800):
    """Room_* rig (BASELINE configs[4]; synthetic_dataset_code/Room.py:18-34, 171-180, 197-310): 88 inward-looking cameras
    on the walls of a 6 x 4 x 3 room -- 24 on the floor edge and 24 on the ceiling edge (every 15 degrees, where the ray
    from the centre meets the rectangle) and 5 rings of 8 (wall centres and corners) in between; pitch atan(z / r_xy)
    towards the floor centre, yaw stepping with the position."""
    rnd = random.Random(seed)
    fovs = [rnd.randint(40, 80) for _ in range(88)]
    rx, ry, rz, step, rounds = 6.0, 4.0, 3.0, 15, 7
    n_loc = 180 // step
    mid = []
    for i in range(1, rounds - 1):
        z = rz / rounds * i
        mid += [(rx / 2, 0, z), (rx / 2, ry / 2, z), (0, ry / 2, z), (-rx / 2, ry / 2, z), (-rx / 2, 0, z), (-rx / 2, -ry / 2, z),
                (0, -ry / 2, z), (rx / 2, -ry / 2, z)]
    e1, e2 = [], []
    for i in range(n_loc):
        ang = step * i
        if ang == 90:
            p1, p2 = (0.0, ry / 2), (0.0, -ry / 2)
        elif ang == 0:
            p1, p2 = (rx / 2, 0.0), (-rx / 2, 0.0)
        else:
            t = math.tan(math.radians(ang))
            x_abs = ry / (2 * t)
            sym = 1.0 if x_abs > 0 else -1.0
            y_abs = sym * t * rx / 2
            if abs(x_abs) >= rx / 2:
                p1, p2 = (rx / 2 * sym, y_abs), (-rx / 2 * sym, -y_abs)
            else:
                p1, p2 = (x_abs, ry / 2), (-x_abs, -ry / 2)
        e1.append(p1)
        e2.append(p2)
    edge = e1 + e2
    locs = [(x, y, 0.0) for x, y in edge] + mid + [(x, y, rz) for x, y in edge]
    theta_t = math.atan(ry / rx)
    ring_yaw = [math.radians(90), math.radians(90) + theta_t, math.radians(180), math.radians(270) - theta_t, math.radians(270),
                math.radians(270) + theta_t, math.radians(360), math.radians(450) - theta_t]
    eulers = []
    for k, loc in enumerate(locs):
        pitch = -math.atan(loc[2] / math.hypot(loc[0], loc[1]))
        if k < 24 or k >= 64:                      # floor / ceiling edge: yaw = 90 deg + 15 deg per position
            yaw = math.radians(90) + math.radians(step) * (k if k < 24 else k - 64)
        else:
            yaw = ring_yaw[(k - 24) % 8]
        eulers.append((math.radians(90) + pitch, yaw))
    return _rig_tensors(locs, eulers, fovs, H, W)


RIGS = {"ball": None, "array": array_cameras, "halfball": halfball_cameras, "room": room_cameras}


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
                   mode=0, barf_mask=False, barf_start=0.3846, barf_end=0.6923, seed=0, n_val_images=1, rig="ball", **extra):
    """`sys_param` for a synthetic scene on one of the reference's rigs (`rig`: ball | array | halfball | room; the
    keys listed in SURVEY.md 8b)."""
    pose, K, _ = ball_cameras(seed, H=H, W=W) if rig == "ball" else RIGS[rig](seed, H=H, W=W)
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


# ---------------------------------------------------------------------------------------------------------------------
# A procedural scene with analytic ground truth (no dataset ships with the reference and there is no network): three soft
# coloured blobs on a white background, volume-rendered by dense quadrature in plain torch.  Used by the convergence runs
# (scripts/train_procedural.py, scripts/train_joint.py) and the convergence gate of the GPU test suite.
BLOB_CENTERS = ((0.5, 0.0, 0.1), (-0.45, 0.35, -0.2), (0.0, -0.5, 0.35))
BLOB_SIGMAS = (0.32, 0.28, 0.22)
BLOB_COLORS = ((0.9, 0.15, 0.1), (0.1, 0.7, 0.2), (0.15, 0.25, 0.9))
BLOB_DENSITY = 18.0


@torch.no_grad()
def blob_scene_render(rays_d, rays_o, near=1.0, far=8.0, n_quad=384):
    """Ground-truth colour [M,3] of the blob scene along rays (white background), the reference's compositing rule
    (model/mc_nerf.py:729-736 without noise) on a dense uniform quadrature."""
    dev = rays_d.device
    cen, sg, col = (torch.tensor(v, device=dev) for v in (BLOB_CENTERS, BLOB_SIGMAS, BLOB_COLORS))
    z = torch.linspace(near, far, n_quad, device=dev)
    x = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.view(1, -1, 1)
    w = torch.exp(-((x.unsqueeze(-2) - cen) ** 2).sum(-1) / (2 * sg ** 2))
    sig, c = BLOB_DENSITY * w.sum(-1), (w.unsqueeze(-1) * col).sum(-2) / (w.sum(-1, keepdim=True) + 1e-8)
    alpha = 1 - torch.exp(-sig * (far - near) / (n_quad - 1))
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], 1), 1)[:, :-1]
    wt = alpha * T
    return (wt.unsqueeze(-1) * c).sum(1) + (1 - wt.sum(1, keepdim=True))


@torch.no_grad()
def blob_scene_images(pose, K, H, W, chunk=8192):
    """[C, H*W, 3] ground-truth images of the blob scene for world->cam poses [C,3,4] and intrinsics [C,3,3] (on their device)."""
    from . import ops
    Kinv = torch.linalg.inv(K)
    allpix = torch.arange(H * W, device=pose.device)
    out = []
    for i in range(pose.shape[0]):
        d, o = ops.raygen_fwd(pose[i].contiguous(), Kinv[i].contiguous(), allpix, W)
        out.append(torch.cat([blob_scene_render(d[j:j + chunk], o[j:j + chunk]) for j in range(0, H * W, chunk)]))
    return torch.stack(out)
