"""Joint camera + radiance-field optimisation (the GLOBAL_OPTIM stage of MC-NeRF, BARF mask on) on the procedural scene of
train_procedural.py, through MC_Model: device-resident uint8 images, fused camera kernels, HIP renderer, fused RAdam.
Cameras start from the ground truth perturbed by `noise` in se(3) (rad / scene units); reports camera errors and PSNR.
Usage (GPU box):  python scripts/train_joint.py [f32|f16x3] [steps] [noise]
"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mc_nerf_amd import ops, synthetic as S
from mc_nerf_amd.data import DeviceImageSet
from mc_nerf_amd.model import MC_Model, RAdam, MC_NeRF_Loss

precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
dev = torch.device("cuda:0")
H = W = 200
N = 8192
torch.manual_seed(0)

CENTERS = torch.tensor([[0.5, 0.0, 0.1], [-0.45, 0.35, -0.2], [0.0, -0.5, 0.35]], device=dev)
SIGMAS = torch.tensor([0.32, 0.28, 0.22], device=dev)
COLORS = torch.tensor([[0.9, 0.15, 0.1], [0.1, 0.7, 0.2], [0.15, 0.25, 0.9]], device=dev)


@torch.no_grad()
def render_gt(d, o, near=1.0, far=8.0, S_=384):
    z = torch.linspace(near, far, S_, device=dev)
    x = o.unsqueeze(1) + d.unsqueeze(1) * z.view(1, -1, 1)
    w = torch.exp(-((x.unsqueeze(-2) - CENTERS) ** 2).sum(-1) / (2 * SIGMAS ** 2))
    sig, col = 18.0 * w.sum(-1), (w.unsqueeze(-1) * COLORS).sum(-2) / (w.sum(-1, keepdim=True) + 1e-8)
    alpha = 1 - torch.exp(-sig * (far - near) / (S_ - 1))
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], 1), 1)[:, :-1]
    wt = alpha * T
    return (wt.unsqueeze(-1) * col).sum(1) + (1 - wt.sum(1, keepdim=True))


sp = S.make_sys_param(dev, samples=64, scale=2, batch=N, H=H, W=W, barf_mask=True, precision=precision)
pose_gt, K_gt = sp["gt_pose"].to(dev), sp["intr_mat"][0].to(dev)
Kinv_gt = torch.linalg.inv(K_gt)
C = pose_gt.shape[0]
allpix = torch.arange(H * W, device=dev)
imgs = []
for i in range(C):
    d, o = ops.raygen_fwd(pose_gt[i].contiguous(), Kinv_gt[i].contiguous(), allpix, W)
    imgs.append(torch.cat([render_gt(d[j:j + 8192], o[j:j + 8192]) for j in range(0, H * W, 8192)]))
imgs = torch.stack(imgs)
u8 = torch.cat([(imgs * 255).round().clamp(0, 255).to(torch.uint8), torch.full((C, H * W, 1), 255, dtype=torch.uint8, device=dev)], -1)
images = DeviceImageSet(u8, H, W)

model = MC_Model(sp).to(dev)
S.init_cameras_near_gt(model, noise=noise, seed=3)
loss_fn = MC_NeRF_Loss(sp)
cam_params = [p for n, p in model.named_parameters() if not n.startswith("nerf.")]
nerf_params = [p for n, p in model.named_parameters() if n.startswith("nerf.")]
opt = RAdam([{"params": nerf_params, "lr": 5e-4}, {"params": cam_params, "lr": 1e-3}], weight_decay=0.0)
wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
wpts, pts = wpts.to(dev), pts.to(dev)


@torch.no_grad()
def report(tag):
    K_e, pose_e, _ = model.add_weights2param(True, True, True)
    R_e, t_e, R_g, t_g = pose_e[:, :, :3], pose_e[:, :, 3], pose_gt[:, :, :3], pose_gt[:, :, 3]
    cosang = ((R_e.transpose(1, 2) @ R_g).diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2
    rot = torch.rad2deg(torch.acos(cosang.clamp(-1, 1)))
    c_e = -(R_e.transpose(1, 2) @ t_e.unsqueeze(-1)).squeeze(-1)          # camera centres
    c_g = -(R_g.transpose(1, 2) @ t_g.unsqueeze(-1)).squeeze(-1)
    foc = ((K_e[:, 0, 0] - K_gt[:, 0, 0]).abs() / K_gt[:, 0, 0]).mean()
    ps = []
    for i in range(3, C, 13):                                               # 9 views, rendered with the ESTIMATED cameras
        kinv = model.inverse_intrinsic(K_e[i:i + 1])[0]
        d, o = ops.raygen_fwd(pose_e[i].contiguous(), kinv.contiguous(), allpix, W)
        rgb = torch.cat([model.nerf.render_rays_test(d[j:j + 8192], o[j:j + 8192], model.nerf.nerf_coarse, model.nerf.nerf_fine)[0]
                         for j in range(0, H * W, 8192)])
        ps.append(-10 * math.log10(float(((rgb - imgs[i]) ** 2).mean())))
    print(f"{tag}: rotation error {float(rot.mean()):.3f} deg (max {float(rot.max()):.3f}), camera-centre error "
          f"{float((c_e - c_g).norm(dim=-1).mean()):.4f}, focal error {float(foc) * 100:.2f} %, PSNR {sum(ps) / len(ps):.2f} dB")


print(f"precision {precision}, {C} cameras {H}x{W}, se(3) noise {noise}, {N} rays/step, GLOBAL_OPTIM (BARF 0.1 -> 0.5 of the run)")
report("step     0")
t0 = time.time()
sp_b = (sp["barf_start"], sp["barf_end"])
for step in range(1, steps + 1):
    cam = int(torch.randint(C, (1,)))
    data = (images, torch.tensor([cam]), wpts, pts, wpts, pts)
    # progress ratio mapped so that the BARF window [barf_start, barf_end] covers 10 % .. 50 % of this run
    prog = step / steps
    ratio = sp_b[0] + (sp_b[1] - sp_b[0]) * min(max((prog - 0.1) / 0.4, 0.0), 1.0) if prog > 0.1 else 0.0
    loss_dict, _, _, _ = model(data, step, "GLOBAL_OPTIM_EPOCH", ratio)
    loss = loss_fn(loss_dict, "GLOBAL_OPTIM_EPOCH")
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    if step % max(1000, steps // 6) == 0 or step == steps:
        torch.cuda.synchronize()
        report(f"step {step:5d}")
print(f"{(time.time() - t0) / steps * 1e3:.1f} ms/step incl. reports")
