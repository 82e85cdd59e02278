"""Data parallelism of the reference (DDP + DistributedSampler, main.py:60-62, data/data_read.py:358-360,
utils/distributed_init.py:11-34) re-designed for one MI355X node: one process per GPU, cameras sharded over
ranks, and ONE all-reduce per step over ONE flat fp32 gradient buffer (coarse net | fine net | camera
parameters, <= 2.94 MB) instead of per-tensor buckets - the message is latency-bound on xGMI, so fewer,
contiguous collectives are what matters.  `backend="nccl"` is RCCL on ROCm; the CPU tests
run the same code over gloo.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None):
    """env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*), as utils/distributed_init.py:12-31.
    Returns (rank, world, device)."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        if dev.type == "cuda":
            torch.cuda.set_device(dev)
        return 0, 1, dev
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    use_cuda = torch.cuda.is_available()
    if backend is None:
        backend = os.environ.get("MCNERF_DIST_BACKEND") or ("nccl" if use_cuda else "gloo")
    if use_cuda and os.environ.get("MCNERF_SHARE_GPU") == "1":
        # test rigs with fewer GPUs than ranks (RCCL itself needs one GPU per rank: combine with MCNERF_DIST_BACKEND=gloo)
        local %= torch.cuda.device_count()
    dev = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(dev)
    if not dist.is_initialized():
        kw = {"device_id": dev} if (use_cuda and backend == "nccl") else {}
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kw)
    return rank, world, dev


def shard_cameras(num_cams: int, epoch: int, rank: int, world: int, seed: int = 0, shuffle: bool = True) -> List[int]:
    """Camera ids this rank renders in `epoch`: a seeded permutation dealt round-robin, padded so every rank
    takes the same number of steps (the semantics of DistributedSampler(shuffle=True) + set_epoch)."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        order = torch.randperm(num_cams, generator=g).tolist()
    else:
        order = list(range(num_cams))
    total = ((num_cams + world - 1) // world) * world
    order = order + order[: total - num_cams]
    return order[rank:total:world]


class FlatGradSync:
    """One-collective gradient averaging for MC_Model.

    Before `loss.backward()` call `prepare()`: it hands the renderer a zeroed arena so that the HIP backward
    accumulates the two nets' gradients directly into one contiguous buffer.  After backward call `sync()`:
    camera-parameter gradients are appended, the arena is all-reduced once (SUM, then divided by the world
    size = DDP's averaging) and the parameters' `.grad` are pointed at / refreshed from it.
    """

    def __init__(self, model, world: int, check_flags: bool = False, force_collective: bool = False):
        """A parameter receives the reduced gradient whenever ANY rank produced one (DDP semantics).  Deciding that needs the
        reduced "some rank has a gradient" flags on the host.  `check_flags=True` reads them back every step (one device ->
        host synchronisation per step).  The default reads them back only the FIRST time a local flag pattern occurs (a
        pattern changes at stage boundaries: a handful of reads per training): if the ranks agree the pattern is trusted
        from then on and the step runs without host synchronisation, disagreement switches this object to `check_flags`
        for good; later disagreement under a trusted pattern is still counted on the device (`asymmetric_steps()`)."""
        self.model, self.world, self.check_flags = model, world, check_flags
        self.force_collective = force_collective   # run the all-reduce even with one rank (exercises the backend on a 1-GPU box)
        nerf = model.nerf
        self.nets = [nerf.nerf_coarse, nerf.nerf_fine]
        for n in self.nets:
            n.flat_params()
        self.net_sizes = [n.flat_params().numel() for n in self.nets]
        self.cam_params = [p for name, p in model.named_parameters() if not name.startswith("nerf.")]
        self.cam_sizes = [p.numel() for p in self.cam_params]
        self.n_grad = sum(self.net_sizes) + sum(self.cam_sizes)
        # one flag per parameter tensor rides at the end of the same message: "some rank produced a gradient for it"
        self.n_flags = sum(len(n.ordered_parameters()) for n in self.nets) + len(self.cam_params)
        self.total = self.n_grad + self.n_flags
        dev = self.nets[0].flat_params().device
        self.arena = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self._flag_cache = {}                      # tuple of local flags -> device tensor (no per-step host -> device copy)
        self._trusted = set()                      # local flag patterns verified to be the same on every rank
        self._asym = torch.zeros((), dtype=torch.int32, device=dev)
        self._stage = None                         # the model's stage (opt_idx) of the last sync: trust does not survive a stage change

    def broadcast_parameters(self):
        """Rank 0's parameters to everyone (what the DDP constructor does, main.py:61)."""
        if self.world <= 1:
            return
        for n in self.nets:
            dist.broadcast(n.flat_params(), src=0)
        for p in self.cam_params:
            dist.broadcast(p.data, src=0)

    def prepare(self):
        """Call before the step's backward().  Every managed parameter's `.grad` is dropped first: after `sync()` the
        grads are VIEWS of the arena, and the HIP backward writes the next step's gradients into that same arena, so
        a surviving `.grad` (zero_grad(set_to_none=False), a second backward) would make autograd's accumulation
        add the arena to itself and silently double the gradient."""
        for n in self.nets:
            for p in n.ordered_parameters():
                p.grad = None
        for p in self.cam_params:
            p.grad = None
        self.arena.zero_()
        self.model.nerf.grad_arena = self.arena

    def sync(self):
        """Call once after the step's single backward()."""
        nerf = self.model.nerf
        used = getattr(nerf, "grad_arena_used", False)
        if self.world <= 1 and not self.force_collective:      # nothing to reduce: the gradients stay where backward() put them
            nerf.grad_arena = None
            nerf.grad_arena_used = False
            return
        pairs, have = [], []
        o = 0
        for n, sz in zip(self.nets, self.net_sizes):
            for p, po in zip(n.ordered_parameters(), n._offsets):
                v = self.arena[o + po:o + po + p.numel()].view(p.shape)
                if p.grad is not None and not used:      # backward ran without the arena: gather
                    v.copy_(p.grad)
                pairs.append((p, v))
                have.append(1.0 if p.grad is not None else 0.0)
            o += sz
        for p, sz in zip(self.cam_params, self.cam_sizes):
            v = self.arena[o:o + sz].view(p.shape)
            if p.grad is not None:
                v.copy_(p.grad)
            pairs.append((p, v))
            have.append(1.0 if p.grad is not None else 0.0)
            o += sz
        key = tuple(have)
        stage = getattr(self.model, "opt_idx", None)
        if stage != self._stage:                   # a new stage (main.py:79 switches optimisers on it): every rank re-verifies its
            self._stage = stage                    # pattern on the stage's first step, so no rank keeps trusting across a boundary
            self._trusted.clear()                  # where another rank's pattern changed
        local = self._flag_cache.get(key)
        if local is None:                          # (first step of a stage only)
            local = self._flag_cache[key] = torch.tensor(have, dtype=torch.float32, device=self.arena.device)
        self.arena[self.n_grad:].copy_(local)
        dist.all_reduce(self.arena, op=dist.ReduceOp.SUM)     # the step's ONE collective
        trusted = not (self.check_flags or key not in self._trusted)
        fused = trusted and self.arena.is_cuda                # average + flag check in one launch (mcnerf_sync_finish)
        if fused:
            from . import ops
            ops.sync_finish(self.arena, self.n_grad, self.world, local, self._asym)
        else:
            self.arena[:self.n_grad].div_(self.world)
        if not trusted:
            reduced = self.arena[self.n_grad:].tolist()       # (device -> host synchronisation)
            if all((r > 0) == (h > 0) and (r == 0 or r == self.world) for r, h in zip(reduced, have)):
                self._trusted.add(key)
            elif not self.check_flags:
                import logging
                logging.warning("FlatGradSync: the ranks disagree on which parameters have gradients; reading the flags back every step from now on")
                self.check_flags = True
            have = reduced
        elif not fused:                            # trusted pattern: same flags on every rank <=> reduced == world * local; count the steps where not
            self._asym += (self.arena[self.n_grad:] != local * self.world).any().to(torch.int32)
        # A reduced slice becomes its parameter's gradient whenever a rank produced one (DDP semantics: a parameter unused
        # on this rank but used elsewhere still receives the averaged gradient, otherwise the replicas diverge -- with
        # check_flags; without it the ranks are assumed to run the same stage and a violation is counted); parameters
        # without a gradient (a stage that does not touch them) keep `.grad = None`, so the optimiser skips them exactly
        # as in the reference.
        for (p, v), h in zip(pairs, have):
            p.grad = v if h > 0 else None
        nerf.grad_arena = None
        nerf.grad_arena_used = False

    def asymmetric_steps(self) -> int:
        """Steps (since construction) in which the ranks disagreed on which parameters have a gradient under a pattern that
        had been verified earlier, i.e. steps whose flags were not read back (one device -> host read; call it at an epoch
        boundary).  Must be 0."""
        return int(self._asym.item())
