"""Achievable HBM write / read / copy bandwidth with plain torch kernels (context for the workspace-bound kernels)."""
import torch
dev = torch.device("cuda:0")
n = 4 * 1024 ** 3            # 16 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.zero_()); print(f"fill  16 GiB: {ms:.2f} ms  {x.numel()*4/ms/1e9:.2f} TB/s write")
ms = t(lambda: x.sum()); print(f"sum   16 GiB: {ms:.2f} ms  {x.numel()*4/ms/1e9:.2f} TB/s read")
y = torch.empty_like(x)
ms = t(lambda: y.copy_(x)); print(f"copy  16 GiB: {ms:.2f} ms  {2*x.numel()*4/ms/1e9:.2f} TB/s read+write")
h = x.view(torch.float16)
ms = t(lambda: h.fill_(1.0)); print(f"fill f16 16 GiB: {ms:.2f} ms  {x.numel()*4/ms/1e9:.2f} TB/s write")
