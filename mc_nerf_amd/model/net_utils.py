"""Optimiser of the reference's train loop (reference: model/net_utils.py:10-101): Rectified Adam with
the 10-slot step-size cache, the N_sma >= 5 switch and the `p -= wd*lr*p` decay applied before the Adam
update.  Per-tensor torch ops for now (the fused flat-buffer version is a later
widening step); the parameters it updates are views of the flat buffers the HIP kernels read."""
import math

import torch
from torch.optim.optimizer import Optimizer


class RAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid RAdam hyper-parameter")
        self.degenerated_to_sgd = degenerated_to_sgd
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
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, wd, eps = group["lr"], group["weight_decay"], group["eps"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
                m.mul_(beta1).add_(g, alpha=1 - beta1)
                st["step"] += 1
                slot = group["buffer"][st["step"] % 10]
                if slot[0] != st["step"]:
                    slot[0] = st["step"]
                    slot[1], slot[2] = self._rectification(st["step"], beta1, beta2, self.degenerated_to_sgd)
                n_sma, step_size = slot[1], slot[2]
                if n_sma >= 5:
                    if wd != 0:
                        p.add_(p, alpha=-wd * lr)
                    p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size * lr)
                elif step_size > 0:
                    if wd != 0:
                        p.add_(p, alpha=-wd * lr)
                    p.add_(m, alpha=-step_size * lr)
        return loss
