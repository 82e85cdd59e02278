"""Optimiser of the reference's train loop (reference: model/net_utils.py:10-101): Rectified Adam with
the 10-slot step-size cache, the N_sma >= 5 switch and the `p -= wd*lr*p` decay applied before the Adam
update.  On the GPU every tensor of a param group is updated by ONE launch of the fused multi-tensor
kernel (`mcnerf_radam_step`, csrc/optim.hip) instead of ~10 ATen ops per tensor; the
parameters it updates are views of the flat buffers the HIP render kernels read.  Host tensors (the CPU
plumbing tests of the camera-only stage) take the same arithmetic through plain torch ops."""
import ctypes
import math

import torch
from torch.optim.optimizer import Optimizer

from .. import _lib


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid RAdam hyper-parameter")
        self.degenerated_to_sgd = degenerated_to_sgd
        self._guard = {}            # device -> int32[2]: [0] raised by a non-finite gradient this step, [1] skipped steps so far
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                        buffer=[[None, None, None] for _ in range(10)])
        super().__init__(params, defaults)

    @staticmethod
    def _rectification(step, beta1, beta2, degenerate):
        beta2_t = beta2 ** step
        n_max = 2.0 / (1.0 - beta2) - 1.0
        n_sma = n_max - 2.0 * step * beta2_t / (1.0 - beta2_t)
        if n_sma >= 5:
            r = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2))
            return n_sma, r / (1 - beta1 ** step)
        if degenerate:
            return n_sma, 1.0 / (1 - beta1 ** step)
        return n_sma, -1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        buckets = []            # (tensors, group, n_sma, step_size): the fused launches of this step
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            fused = {}          # step count -> tensors updated by one fused launch
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                slot = group["buffer"][st["step"] % 10]
                if slot[0] != st["step"]:
                    slot[0] = st["step"]
                    slot[1], slot[2] = self._rectification(st["step"], beta1, beta2, self.degenerated_to_sgd)
                if p.is_cuda:
                    fused.setdefault(st["step"], []).append(p)
                else:
                    self._step_host(p, p.grad, st, group, slot[1], slot[2])
            for step, ps in fused.items():
                slot = group["buffer"][step % 10]
                buckets.append((ps, group, slot[1], slot[2]))
        # the overflow guard covers the WHOLE step: every bucket's gradients are checked before any bucket is updated, so a
        # non-finite gradient anywhere refuses the step everywhere (counted once).  (The host-side step counters have
        # advanced by then: after a refused step the bias corrections run one step ahead -- a refused step is an error
        # condition, see raise_on_overflow, not a mode of operation.)
        first, last = {}, {}    # (one guard per device)
        for i, b in enumerate(buckets):
            first.setdefault(b[0][0].device, i)
            last[b[0][0].device] = i
        tables = [self._bucket_tables(ps) for ps, _, _, _ in buckets]          # (built once: both passes use the same pointer tables)
        for i, (ps, group, n_sma, ss) in enumerate(buckets):
            self._step_fused(ps, group, n_sma, ss, phase=(1 if first[ps[0].device] == i else 0) | 2, tables=tables[i])
        for i, (ps, group, n_sma, ss) in enumerate(buckets):
            self._step_fused(ps, group, n_sma, ss, phase=4 | (8 if last[ps[0].device] == i else 0), tables=tables[i])
        return loss

    def skipped_steps(self) -> int:
        """Steps the fused kernel refused because a gradient was inf / NaN (overflow of a reduced-precision mode): the
        parameters and moments of such a step are left untouched.  Synchronises; call it at epoch ends / checkpoints."""
        return int(sum(int(g[1].item()) for g in self._guard.values()))

    def raise_on_overflow(self):
        """Raises if the guard has refused steps, NAMING the weight tensors that left their precision mode's operand range
        (the packing kernels flag them per tensor: ops.range_report); with every weight in range it is an activation or a
        gradient that overflowed (f16: 65504; f16x3: activations 8188)."""
        n = self.skipped_steps()
        if n:
            from .. import ops
            bad = ops.range_report()
            if bad:
                why = "weights outside the operand range of their precision mode: " + "; ".join(
                    f"{label} {name} (mode {prec}: {ops.RANGE_TEXT[prec]})" for label, name, prec in bad)
            else:
                why = ("every packed weight is inside its mode's range, so an activation or a gradient overflowed the 16-bit operand "
                       "format (f16: 65504; f16x3: |activation| < 8188)")
            raise _lib.McnerfError(f"{n} optimiser step(s) skipped: non-finite gradients -- {why}.  Switch `precision` to 'f32' or "
                                   "lower the learning rate / add weight decay")

    @staticmethod
    def _step_host(p, g, st, group, n_sma, step_size):
        beta1, beta2 = group["betas"]
        lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
        m, v = st["exp_avg"], st["exp_avg_sq"]
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        if n_sma >= 5:
            if wd != 0:
                p.add_(p, alpha=-wd * lr)
            p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size * lr)
        elif step_size > 0:
            if wd != 0:
                p.add_(p, alpha=-wd * lr)
            p.add_(m, alpha=-step_size * lr)

    def _bucket_tables(self, ps):
        """The device-pointer tables of one fused launch: parameters, gradients, both moments, sizes (+ the gradient tensors
        they point into, kept alive until the step's last launch)."""
        n = len(ps)
        PtrArr, SizeArr = ctypes.c_void_p * n, ctypes.c_longlong * n
        grads = []
        for p in ps:
            g = p.grad
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.McnerfError("fused RAdam needs contiguous fp32 parameters")
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            grads.append(g)
        args = (PtrArr(*[p.data_ptr() for p in ps]), PtrArr(*[g.data_ptr() for g in grads]),
                PtrArr(*[self.state[p]["exp_avg"].data_ptr() for p in ps]),
                PtrArr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in ps]),
                SizeArr(*[p.numel() for p in ps]))
        return args, grads

    def _step_fused(self, ps, group, n_sma, step_size, phase=15, tables=None):
        n = len(ps)
        args, _keep = tables if tables is not None else self._bucket_tables(ps)
        beta1, beta2 = group["betas"]
        dev = ps[0].device
        if dev not in self._guard:
            self._guard[dev] = torch.zeros(2, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("mcnerf_radam_step", n, *[ctypes.cast(a, ctypes.c_void_p) for a in args],
                      float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]),
                      float(step_size), int(n_sma >= 5), self._guard[dev].data_ptr(), int(phase), torch.cuda.current_stream().cuda_stream)
