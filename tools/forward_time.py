#!/usr/bin/env python3
"""ELBO-forward time (no autograd) of the headline 784-1200-1200-10 MNF net at batch 4096 for a given flow family
(FLOW=Planar | RNVP | MNF | Householder ...), eager and replayed from a HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd

dev = torch.device("cuda:0")
bnn_amd.set_precision(os.environ.get("PREC", "bf16x3"))
FLOW = os.environ.get("FLOW", "RNVP")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type=FLOW, r_flow_type=FLOW).to(dev).train()
x = torch.rand(4096, 784, device=dev)


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    def fwd():
        out = net(x, sample=True)
        return out, net.kl()
    print("%s eager: %.3f ms/forward" % (FLOW, timed(fwd)))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        res = fwd()
    print("%s HIP graph: %.3f ms/forward (kl %.1f)" % (FLOW, timed(g.replay), float(res[1])))
