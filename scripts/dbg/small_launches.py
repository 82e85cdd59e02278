"""Debug aid: which Python lines of one training step launch the small torch kernels (elementwise / copy / fill / reduce / random)?
    python scripts/dbg/small_launches.py        (GPU box)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from mc_nerf_amd import synthetic as S, distributed as D
from mc_nerf_amd.model import MC_Model, MC_NeRF_Loss, RAdam
from mc_nerf_amd.data import DeviceImageSet
dev = torch.device("cuda:0")
sp = S.make_sys_param(dev, samples=64, scale=2, batch=4096, H=800, W=800, barf_mask=False, precision="f16x3h", rig="ball", coarse=(4, 128, [2]))
model = MC_Model(sp).to(dev)
S.init_cameras_near_gt(model, noise=1e-3)
loss_fn = MC_NeRF_Loss(sp)
opt = RAdam(model.parameters(), lr=5e-4, weight_decay=4e-4)
sync = D.FlatGradSync(model, 1)
model.nerf.reserve_workspaces(4096)
wpts, pts = S.calibration_points(sp["gt_pose"], sp["intr_mat"][0])
wpts, pts = wpts.to(dev), pts.to(dev)
images = DeviceImageSet.synthetic(model.train_numb, 800, 800, dev, channels=4, seed=7)
def step(i):
    data = (images, torch.tensor([i % 100]), wpts, pts, wpts, pts)
    loss_dict, _, _, _ = model(data, 20, "GLOBAL_OPTIM_EPOCH", 0.6)
    loss = loss_fn(loss_dict, "GLOBAL_OPTIM_EPOCH")
    opt.zero_grad(set_to_none=True)
    sync.prepare()
    loss.backward()
    sync.sync()
    opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(4): step(5 + i)
    torch.cuda.synchronize()
ev = prof.events()
by = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and any(k in e.name for k in
        ("copy_", "fill_", "zero_", "add", "mul", "sub", "div", "clone", "randn", "normal_", "uniform_", "sum", "mean", "cat", "stack", "to", "abs", "neg", "index", "select", "where", "clamp", "max", "min", "empty", "zeros", "ones")):
        st = [s for s in (e.stack or []) if "/repo/" in s and "small_launches" not in s]
        where = st[0].split("/repo/")[-1] if st else "?"
        by[(e.name, where)] += 1
for (name, where), n in sorted(by.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n / 4:6.1f}/step  {name:28s} {where}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=35, max_name_column_width=60))
