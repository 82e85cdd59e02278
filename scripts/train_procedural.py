"""End-to-end convergence check on a PROCEDURAL scene (no dataset exists in the container): three soft coloured blobs
on a white background, ground-truth images volume-rendered analytically with dense quadrature in plain torch, 110
Ball-rig cameras with known poses.  Trains the coarse+fine nets through the HIP path (renderer fwd/bwd + fused RAdam) and
reports PSNR of held-out views.  Usage (GPU box):  python scripts/train_procedural.py [f32|f16x3|f16x3h|f16|bf16] [steps]
"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mc_nerf_amd import ops, synthetic as S
from mc_nerf_amd.model import NeRF_Model, RAdam, MC_NeRF_Loss

precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = torch.device("cuda:0")
H = W = 200
N = 8192
torch.manual_seed(0)

CENTERS = torch.tensor([[0.5, 0.0, 0.1], [-0.45, 0.35, -0.2], [0.0, -0.5, 0.35]], device=dev)
SIGMAS = torch.tensor([0.32, 0.28, 0.22], device=dev)
COLORS = torch.tensor([[0.9, 0.15, 0.1], [0.1, 0.7, 0.2], [0.15, 0.25, 0.9]], device=dev)
DENS = 18.0


def scene(x):                                   # x [...,3] -> density [...], colour [...,3]
    d2 = ((x.unsqueeze(-2) - CENTERS) ** 2).sum(-1)
    w = torch.exp(-d2 / (2 * SIGMAS ** 2))
    return DENS * w.sum(-1), (w.unsqueeze(-1) * COLORS).sum(-2) / (w.sum(-1, keepdim=True) + 1e-8)


@torch.no_grad()
def render_gt(d, o, near=1.0, far=8.0, S_=384):
    z = torch.linspace(near, far, S_, device=dev)
    x = o.unsqueeze(1) + d.unsqueeze(1) * z.view(1, -1, 1)
    sig, col = scene(x)
    dz = (far - near) / (S_ - 1)
    alpha = 1 - torch.exp(-sig * dz)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], 1), 1)[:, :-1]
    w = alpha * T
    return (w.unsqueeze(-1) * col).sum(1) + (1 - w.sum(1, keepdim=True))


pose, K, _ = S.ball_cameras(seed=0, radius=3.0, H=H, W=W)
pose, K = pose.to(dev), K.to(dev)
Kinv = torch.linalg.inv(K)
C = pose.shape[0]
test_ids = list(range(5, C, 22))                # 5 held-out views
train_ids = [i for i in range(C) if i not in test_ids]
allpix = torch.arange(H * W, device=dev)
imgs = {}
for i in range(C):
    d, o = ops.raygen_fwd(pose[i].contiguous(), Kinv[i].contiguous(), allpix, W)
    imgs[i] = torch.cat([render_gt(d[j:j + 8192], o[j:j + 8192]) for j in range(0, H * W, 8192)])

sp = S.make_sys_param(dev, samples=64, scale=2, batch=N, H=H, W=W, precision=precision)
model = NeRF_Model(sp).to(dev)
opt = RAdam(model.parameters(), lr=5e-4, weight_decay=0.0)
loss_fn = MC_NeRF_Loss(sp)


@torch.no_grad()
def psnr_test():
    vals = []
    for i in test_ids:
        d, o = ops.raygen_fwd(pose[i].contiguous(), Kinv[i].contiguous(), allpix, W)
        rgb = torch.cat([model.render_rays_test(d[j:j + 8192], o[j:j + 8192], model.nerf_coarse, model.nerf_fine)[0]
                         for j in range(0, H * W, 8192)])
        vals.append(-10 * math.log10(float(((rgb - imgs[i]) ** 2).mean())))
    return sum(vals) / len(vals)


print(f"precision {precision}: {C} cameras {H}x{W}, {len(train_ids)} train / {len(test_ids)} test views, {N} rays/step")
print(f"step {0:5d}  test PSNR {psnr_test():6.2f} dB")
t0 = time.time()
for step in range(1, steps + 1):
    i = train_ids[int(torch.randint(len(train_ids), (1,)))]
    pix = torch.randperm(H * W, device=dev)[:N]
    d, o = ops.raygen_fwd(pose[i].contiguous(), Kinv[i].contiguous(), pix, W)
    rgb_c, rgb_f = model.render_rays_train(d, o, step, 1.0)
    loss = loss_fn.get_rgb_loss([rgb_c, rgb_f, imgs[i][pix]])
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    if step % 250 == 0 or step == steps:
        torch.cuda.synchronize()
        print(f"step {step:5d}  loss {float(loss.detach()):.5f}  test PSNR {psnr_test():6.2f} dB  ({(time.time() - t0) / step * 1e3:.1f} ms/step incl. eval)")
