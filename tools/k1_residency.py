#!/usr/bin/env python3
"""Does the weight pass find its parameters in the Infinity Cache inside the forward?  (cache-policy hints for
the GEMMs' stores so that mu / rho / lambda survive between steps.)  The headline forward's recorded launch plan is replayed
(a) whole, (b) its flow + weight-pass call alone, back to back, (c) that call alone after a 512 MB device memset has swept the
caches; HIP events around the call, medians over 200 runs.  If (a) ~ (b) << (c) the parameters are already resident between
steps and there is nothing for a store hint to protect."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import graphs

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
net.set_precision("fp16x3f")
x = torch.rand(4096, 784, device=dev)
with torch.no_grad():
    for _ in range(3):
        net(x, sample=True)
    plan = graphs.LaunchPlan(net, x, sample=True)
names = [c[0] for c in plan._calls]
print("calls of one forward:", names)
first = plan._calls[0]


def run_first():
    name, fn, args = first
    plan._check(fn(*args), name)


def timed(pre, body, n=200):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); body(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for _ in range(50):
    plan()
torch.cuda.synchronize()
whole = timed(lambda: None, plan)
a = timed(plan, run_first)                       # the call right after a whole forward (as in a training / serving loop)
b = timed(run_first, run_first)                  # the call right after itself
sweep = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
c = timed(lambda: sweep.fill_(1), run_first)     # after 512 MB of stores have gone through the caches
print("whole forward (event bracket)                          %.1f us" % whole)
print("flows + weight pass, after a whole forward             %.1f us" % a)
print("flows + weight pass, after itself                      %.1f us" % b)
print("flows + weight pass, after a 512 MB memset             %.1f us" % c)
