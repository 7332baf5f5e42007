#!/usr/bin/env python3
"""HIP-graph replay time of one training step (forward + backward + Adam) of an LRT network (BASELINE configs[1]'s shape:
784-400-400-10, B = 1024; DIMS / B / PREC in the environment).  LBBNN_LRT_BIAS_HIP=0 selects the round-2 route of the bias
gradients (a torch autograd graph over the two bias vectors of every layer) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd

dev = torch.device("cuda:0")
dims = tuple(int(v) for v in os.environ.get("DIMS", "784,400,400,10").split(","))
B = int(os.environ.get("B", "1024"))
torch.manual_seed(0)
net = bnn_amd.lrt.BayesianNetwork(dims).to(dev).train()
net.set_precision(os.environ.get("PREC", "fp16x3f"))
opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
x = torch.rand(B, dims[0], device=dev); y = torch.randint(0, dims[3], (B,), device=dev)
lf = lambda n, a, b: bnn_amd.elbo_loss(n(a, sample=True), b, n.kl(), 60)
step = bnn_amd.graphs.make_graphed_train_step(net, opt, lf, x, y)
for _ in range(20):
    step(x, y)
torch.cuda.synchronize()
ts = []
for r in range(5):
    t0 = time.perf_counter()
    for _ in range(100):
        step.graph.replay()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 100 * 1e3)
l0 = float(step(x, y))
print("LRT %s B=%d %s  LBBNN_LRT_BIAS_HIP=%s: %.4f ms per step (median of 5 x 100 replays; min %.4f), loss %.1f"
      % (dims, B, net.precision, os.environ.get("LBBNN_LRT_BIAS_HIP", "1"), sorted(ts)[2], min(ts), l0))
