#!/usr/bin/env python3
"""Headline forward: eager launches from Python vs graphs.LaunchPlan (recorded C calls) vs HIP-graph replay: GPU time per step over
200 steps, over a 20-step region that starts from an idle queue (the driver's command), and the host's time to enqueue one step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import graphs
dev = torch.device("cuda:0")
bnn_amd.set_precision("bf16x3")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
x = torch.rand(4096, 1, 28, 28, device=dev)
def region(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    def eager():
        return net(x, sample=True), net.kl()
    for _ in range(300): eager()
    plan = graphs.LaunchPlan(net, x, sample=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eager()
    for name, fn in (("eager", eager), ("plan (%d C calls)" % len(plan), plan), ("graph replay", g.replay)):
        for _ in range(300): fn()
        r200 = min(region(fn, 200) for _ in range(3))
        r20 = sorted(region(fn, 20) for _ in range(7))[3]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        host = (time.perf_counter() - t0) / 20 * 1e6
        torch.cuda.synchronize()
        print("%-22s 200 steps: %.1f us/step | 20 steps from idle: %.1f us/step (median of 7) | host enqueue %.0f us/step" % (name, r200, r20, host))
