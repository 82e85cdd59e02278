"""HBM write / read / copy rates of plain torch kernels on this box (context for the roofline fractions)."""
import torch
dev = torch.device("cuda:0")
n = 2 * 1024**3          # 8 GB of fp32
x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
s = t(lambda: x.fill_(1.0)); print(f"fill  (write 8 GB): {n*4/s/1e12:.2f} TB/s")
s = t(lambda: x.zero_()); print(f"memset (write 8 GB): {n*4/s/1e12:.2f} TB/s")
s = t(lambda: y.copy_(x)); print(f"copy  (read 8 + write 8 GB): {2*n*4/s/1e12:.2f} TB/s total")
s = t(lambda: x.sum()); print(f"sum   (read 8 GB): {n*4/s/1e12:.2f} TB/s")
s = t(lambda: x.add_(1.0)); print(f"add_  (read 8 + write 8 GB): {2*n*4/s/1e12:.2f} TB/s total")
