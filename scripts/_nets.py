"""Shared helper of the timing scripts: a flat parameter buffer with nn.Linear's default initialisation (no oracle import)."""
import math
import torch
from mc_nerf_amd import ops

NETS = {256: (8, 256, 4), 128: (4, 128, 2), 64: (8, 64, 4), 32: (4, 32, 2)}


def make_net(width, device, seed=7):
    net = ops.Net(*NETS[width])
    g = torch.Generator().manual_seed(seed)
    tensors = []
    for shp in net.shapes():
        fan_in = shp[1] if len(shp) == 2 else None
        if fan_in is not None:
            bound = 1.0 / math.sqrt(fan_in)
            last_fan = fan_in
        else:
            bound = 1.0 / math.sqrt(last_fan)
        tensors.append(((torch.rand(shp, generator=g) * 2 - 1) * bound).to(device))
    return net, ops.flatten_params(net, tensors, device)
