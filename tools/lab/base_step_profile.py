"""Kernel list of one eager sample_elbo training step of the baseline LBBNN network (LBBNN-GP-MF.py, 784-400-600-10, B = 100):
how much of it is torch's own kernels (gate draws, Gamma rsamples, log-prob glue)?  Run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bnn_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bnn_amd.base.BayesianNetwork().to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
x = torch.rand(100, 784, device=dev); y = torch.randint(0, 10, (100,), device=dev)
for it in range(30):
    if it == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    loss, lp, lq, nll = net.sample_elbo(x, y)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("eager sample_elbo step: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
