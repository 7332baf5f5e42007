#!/usr/bin/env python3
"""Wall time of one full training step (forward + backward + Adam) of the headline MNF/planar net, B=4096."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd

dev = torch.device("cuda:0")
bnn_amd.set_precision(os.environ.get("PREC", "bf16x3"))
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
x = torch.rand(4096, 1, 28, 28, device=dev); y = torch.randint(0, 10, (4096,), device=dev)


def step(backward=True):
    net.zero_grad(set_to_none=True)
    out = net(x, sample=True)
    loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + net.kl() / 15
    if backward:
        loss.backward()
        opt.step()
    return loss


for name, bw in (("forward only (autograd graph built)", False), ("forward + backward + Adam", True)):
    for _ in range(3):
        step(bw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step(bw)
    torch.cuda.synchronize()
    print("%s: %.3f ms/step" % (name, (time.perf_counter() - t0) / n * 1e3))
