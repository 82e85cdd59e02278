import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mc_nerf_amd import synthetic as S, distributed as D
from mc_nerf_amd.model import MC_Model, MC_NeRF_Loss, RAdam
from mc_nerf_amd.data import DeviceImageSet
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
torch.manual_seed(42)
sp = S.make_sys_param(dev, samples=64, scale=2, batch=N, H=800, W=800, precision=prec)
model = MC_Model(sp).to(dev); S.init_cameras_near_gt(model, noise=1e-3)
loss_fn = MC_NeRF_Loss(sp); opt = RAdam(model.parameters(), lr=5e-4, weight_decay=4e-4)
sync = D.FlatGradSync(model, 1)
wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0]); wpts, pts = wpts.to(dev), pts.to(dev)
images = DeviceImageSet.synthetic(110, 800, 800, dev)
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
bad_steps = 0
for i in range(STEPS):
    data = (images, torch.tensor([i % 110]), wpts, pts, wpts, pts)
    ld, *_ = model(data, 20, "GLOBAL_OPTIM_EPOCH", 0.6)
    loss = loss_fn(ld, "GLOBAL_OPTIM_EPOCH")
    opt.zero_grad(set_to_none=True); sync.prepare(); loss.backward(); sync.sync()
    g = sync.arena
    k = int(model.nerf.last_selection[1].item())
    fin = bool(torch.isfinite(g).all())
    bad_steps += 0 if fin else 1
    if STEPS <= 12 or i % 25 == 0 or not fin:
      print(f"step {i}: loss {float(loss.detach()):.5f} K/ray {k/N:.1f} grad finite {bool(torch.isfinite(g).all())} |g|max {float(g.abs().max()):.3e} "
          f"nan params {sum(int((~torch.isfinite(p)).sum()) for p in model.parameters())}")
    bad = [(n, float(p.grad.abs().max())) for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if bad: print("   non-finite grads in:", bad[:6])
    opt.step()
print(f"{STEPS} steps, steps with non-finite gradients: {bad_steps}, non-finite parameters at the end: "
      f"{sum(int((~torch.isfinite(p)).sum()) for p in model.parameters())}")
