"""Host-side contract of the reference-shaped classes that needs no GPU: constructor keys, parameter names /
shapes (checkpoint compatibility), flat-buffer aliasing, loss and RAdam arithmetic."""
import math

import pytest
import torch

from mc_nerf_amd import synthetic as S


def test_state_dict_keys_match_reference_checkpoint_format():
    from mc_nerf_amd.model import MC_Model
    sp = S.make_sys_param("cpu", samples=64, scale=2, batch=128, H=8, W=8)
    m = MC_Model(sp)
    keys = list(m.state_dict().keys())
    expect = ["weights_pose", "weights_pose_intr", "weights_ux", "weights_uy", "weights_fx", "weights_fy"]
    for net, depth in (("nerf_coarse", 4), ("nerf_fine", 8)):
        for i in range(1, depth + 1):
            expect += [f"nerf.{net}.xyz_encoding_{i}.0.weight", f"nerf.{net}.xyz_encoding_{i}.0.bias"]
        for head in ("sigma", "sh"):
            expect += [f"nerf.{net}.{head}.0.weight", f"nerf.{net}.{head}.0.bias",
                       f"nerf.{net}.{head}.2.weight", f"nerf.{net}.{head}.2.bias"]
    assert keys == expect and len(keys) == 46
    sd = m.state_dict()
    assert sd["nerf.nerf_fine.xyz_encoding_5.0.weight"].shape == (256, 319)       # skip layer: [x_enc, h]
    assert sd["nerf.nerf_coarse.xyz_encoding_3.0.weight"].shape == (128, 191)
    assert sd["nerf.nerf_fine.sh.2.weight"].shape == (27, 256) and sd["weights_pose"].shape == (110, 6)
    n_coarse = sum(p.numel() for p in m.nerf.nerf_coarse.parameters())
    n_fine = sum(p.numel() for p in m.nerf.nerf_fine.parameters())
    assert (n_coarse, n_fine) == (102428, 631836)
    # main.py:182-186 groups parameters by the first dotted component
    assert {k.split(".")[0] for k, _ in m.named_parameters() if "." in k} == {"nerf"}


def test_parameters_alias_the_flat_buffer_and_survive_load_and_to():
    from mc_nerf_amd.model import NeRF_Model
    from oracle import mcnerf_oracle as O
    sp = S.make_sys_param("cpu", samples=32, scale=2, batch=16, H=8, W=8, coarse=(4, 32, [2]), fine=(8, 64, [4]))
    m = NeRF_Model(sp)
    net = m.nerf_coarse
    flat = net.flat_params()
    p = net.xyz_encoding_2[0].bias
    with torch.no_grad():
        p.add_(1.0)                                   # an optimiser-style in-place update ...
    off = net._offsets[3]
    assert torch.equal(flat[off:off + p.numel()], p.detach())       # ... is visible in the flat buffer
    ref = O.init_params(O.NetCfg(4, 32, (2,)), 5)
    net.load_state_dict(ref)                          # load_state_dict copies in place: aliasing preserved
    assert net.flat_params().data_ptr() == flat.data_ptr()
    assert torch.equal(flat[off:off + p.numel()], ref["xyz_encoding_2.0.bias"])
    net.double().float()                              # .to()/dtype round trip re-creates storages: re-flattened lazily
    flat2 = net.flat_params()
    assert net._aliased() and torch.equal(flat2[off:off + p.numel()], ref["xyz_encoding_2.0.bias"])


def test_loss_matches_reference_formula():
    from mc_nerf_amd.model import MC_NeRF_Loss
    g = torch.Generator().manual_seed(0)
    a, b, gt = (torch.rand(50, 3, generator=g) for _ in range(3))
    L = MC_NeRF_Loss(dict(data_img_h=600, data_img_w=800))
    assert torch.allclose(L.get_rgb_loss([a, b, gt]), ((a - gt) ** 2).mean() + ((b - gt) ** 2).mean())
    assert torch.allclose(L.get_rgb_loss([a, None, gt]), ((a - gt) ** 2).mean())
    pd, pg = torch.rand(1, 4, 5, 2, generator=g) * 800, torch.rand(1, 4, 5, 2, generator=g) * 800
    lr = ((pd[..., 0] - pg[..., 0]) / 800).pow(2).mean() + ((pd[..., 1] - pg[..., 1]) / 600).pow(2).mean()
    assert torch.allclose(L.get_reproject_loss([pd, pg]), lr, rtol=1e-5)
    tot = L({"intr": [pd, pg], "rgb": [a, b, gt]}, "GLOBAL_OPTIM_EPOCH")          # intr term rescaled to 1
    assert torch.allclose(tot, lr / (lr + 1e-8) + L.get_rgb_loss([a, b, gt]))


def test_radam_follows_reference_update_rule():
    """Replays model/net_utils.py:62-99 in float64 python for one scalar parameter."""
    from mc_nerf_amd.model import RAdam
    p = torch.nn.Parameter(torch.tensor([0.7]))
    opt = RAdam([p], lr=0.05, betas=(0.9, 0.999), weight_decay=0.01)
    x, m, v = 0.7, 0.0, 0.0
    for step in range(1, 13):
        g = math.sin(step) + 0.3 * x
        p.grad = torch.tensor([g], dtype=torch.float32)
        opt.step()
        v = 0.999 * v + 0.001 * g * g
        m = 0.9 * m + 0.1 * g
        b2t = 0.999 ** step
        n_max = 2 / (1 - 0.999) - 1
        n_sma = n_max - 2 * step * b2t / (1 - b2t)
        x += -0.01 * 0.05 * x
        if n_sma >= 5:
            ss = math.sqrt((1 - b2t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (1 - 0.9 ** step)
            x += -ss * 0.05 * m / (math.sqrt(v) + 1e-8)
        else:
            x += -0.05 * m / (1 - 0.9 ** step)
        assert abs(float(p) - x) < 2e-5, (step, float(p), x)


def test_barf_weights_and_host_settings():
    from mc_nerf_amd.model import NeRF_Model
    from oracle import mcnerf_oracle as O
    sp = S.make_sys_param("cpu", samples=64, scale=2, batch=16, H=8, W=8, barf_mask=True)
    m = NeRF_Model(sp)
    cfg = O.RenderCfg(barf_mode=True, barf_start=sp["barf_start"], barf_end=sp["barf_end"])
    for r in (0.1, 0.45, 0.6, 0.9):
        assert torch.equal(m.emmbedding_xyz.barf_weights(r), O.barf_weights(r, cfg))
    m.emmbedding_xyz.barf_mode = False
    assert torch.equal(m.emmbedding_xyz.barf_weights(0.5), torch.ones(10))
    assert torch.equal(m.z_vals_f, torch.linspace(1.0, 8.0, 128)) and m.settings.samples_f == 128
    with pytest.raises(Exception):
        m(torch.zeros(4, 3), torch.zeros(4, 3), 0, 1.0)          # CPU tensors: fails loudly, no fallback


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """save_model writes {'model_nerf': MC_Model.state_dict()} (model/mc_nerf.py:738-752); demo mode loads it back
    through the prefix-stripping rewrite (:577-584, 815-837)."""
    from mc_nerf_amd.model import MC_Model, NeRF_Model
    sp = S.make_sys_param("cpu", samples=32, scale=2, batch=16, H=8, W=8, coarse=(4, 32, [2]), fine=(8, 64, [4]),
                          root_weight=str(tmp_path))
    torch.manual_seed(1)
    m = MC_Model(sp)
    path = m.nerf.save_model(m, epoch=3)
    ck = torch.load(path, map_location="cpu")
    assert list(ck.keys()) == ["model_nerf"] and len(ck["model_nerf"]) == 46
    sp2 = dict(sp, mode=1, demo_ckpt=path)
    torch.manual_seed(2)                               # different init: everything must come from the checkpoint
    n = NeRF_Model(sp2)
    for k, v in m.nerf.nerf_fine.state_dict().items():
        assert torch.equal(n.nerf_fine.state_dict()[k], v)
    for k, v in m.nerf.nerf_coarse.state_dict().items():
        assert torch.equal(n.nerf_coarse.state_dict()[k], v)
    # a checkpoint made from plain nn.Linear modules with the reference's names loads as well
    ref_sd = {k: v.clone() for k, v in ck["model_nerf"].items()}
    m2 = MC_Model(sp)
    m2.load_state_dict(ref_sd)
    assert torch.equal(m2.nerf.nerf_fine.flat_params()[:5], m.nerf.nerf_fine.flat_params()[:5])


def test_main_py_epoch_end_call_sequence(tmp_path):
    """The reference's train loop ends every epoch with (main.py:92-95)
        mc_nerf.nerf.save_model(model, epoch); mc_nerf.show_estimate_param(intr, pose, epoch, epoch_type);
        mc_nerf.show_RT_est_results(epoch, epoch_type, mode='epoch'); mc_nerf.nerf.valid_train(epoch, rays_valid, epoch_type)
    with exactly these arguments: the hooks must accept them (a signature drift here raises a TypeError at
    the end of the first epoch)."""
    from mc_nerf_amd.model import MC_Model
    sp = S.make_sys_param("cpu", samples=32, scale=2, batch=16, H=8, W=8, coarse=(4, 32, [2]), fine=(8, 64, [4]),
                          root_weight=str(tmp_path), demo_render_pth=str(tmp_path))
    m = MC_Model(sp)
    K, pose, _ = m.add_weights2param(True, True, True)
    m.intr_adj, m.pose_adj = K, pose
    intr_show = [m.intr_train.detach(), K.detach()]
    pose_show = [m.gt_pose.detach(), pose.detach()]
    rays_valid = [torch.zeros(64, 3), torch.zeros(64, 3), torch.zeros(1, 64, 3)]
    for epoch_type in ("CAM_PARAM_EPOCH",):
        m.nerf.save_model(m, 0)
        assert m.nerf.file_path.endswith(m.nerf.model_name) and m.nerf.model_name.endswith(".ckpt")      # reference attributes
        m.show_estimate_param(intr_show, pose_show, 0, epoch_type)
        m.show_RT_est_results(0, epoch_type, mode='epoch')
        assert m.nerf.valid_train(0, rays_valid, epoch_type) == 0          # camera-only stage: no render (mc_nerf.py:755-756)
    import inspect
    sig = inspect.signature(m.show_RT_est_results)
    assert list(sig.parameters) == ["epoch", "epoch_type", "mode", "show_info"]
    assert list(inspect.signature(m.nerf.valid_train).parameters) == ["epoch", "val_data", "epoch_type"]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` (the driver's scaling form) must start 2 ranks itself; --selftest runs the launcher
    and the rendezvous over gloo without kernels."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCNERF_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["allreduce_ok"] and out["backend"] == "gloo"
    # the driver's largest form: eight ranks of one node (rendezvous on 127.0.0.1, all-reduce across all eight)
    r8 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--selftest"], env=env,
                        capture_output=True, text=True, timeout=480)
    assert r8.returncode == 0, r8.stderr[-2000:]
    out8 = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][-1])
    assert out8["n_gpus"] == 8 and out8["allreduce_ok"] and out8["backend"] == "gloo"
    # a world size that disagrees with --gpus is an error, not a silent 1-rank run
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "--gpus 2" in (r2.stderr + r2.stdout)


def test_bench_under_the_drivers_torchrun_command():
    """The driver's N > 1 launch, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- the ranks are torchrun's processes (bench.py must NOT start its own), they read RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, and rank 0 prints the one line (`--selftest`: rendezvous + all-reduce
    over gloo, no kernels)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCNERF_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["allreduce_ok"] and out["backend"] == "gloo"
