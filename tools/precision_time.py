#!/usr/bin/env python3
"""Headline ELBO forward (784-1200-1200-10 MNF / planar, B = 4096, no autograd) under every GEMM precision, interleaved in
ONE process (cdna guide rule 24): rounds of [precision A, precision B, ...] x REPS replays of a recorded launch plan each;
prints the median and the minimum ms per forward of every precision.  PRECS=comma list, ROUNDS, REPS in the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import graphs

dev = torch.device("cuda:0")
precs = os.environ.get("PRECS", "bf16x3,fp16x3,fp16x3f,fp32").split(",")
rounds, reps = int(os.environ.get("ROUNDS", "7")), int(os.environ.get("REPS", "200"))
B = int(os.environ.get("B", "4096"))
torch.manual_seed(0)
x = torch.rand(B, 784, device=dev)
plans = {}
for p in precs:
    net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    net.set_precision(p)
    with torch.no_grad():
        for _ in range(3):
            net(x, sample=True)
        plans[p] = (graphs.LaunchPlan(net, x, sample=True), net)
torch.cuda.synchronize()
res = {p: [] for p in precs}
for r in range(rounds + 1):
    for p in precs:
        plan = plans[p][0]
        for _ in range(20):
            plan()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            plan()
        torch.cuda.synchronize()
        if r:                                            # round 0 warms the clocks
            res[p].append((time.perf_counter() - t0) / reps * 1e3)
for p in precs:
    v = sorted(res[p])
    print("%-8s median %.4f ms  min %.4f ms  (%.2f M samples/s at the median; %d calls per forward)"
          % (p, v[len(v) // 2], v[0], B / v[len(v) // 2] / 1e3, len(plans[p][0])))
