#!/usr/bin/env python3
"""Forward time of the other BASELINE.json configurations (parity-test cases, not bench lines): configs[1] LRT
784-400-400-10 at B=1024 and configs[4] variational dropout 3072-4096-4096-10 at B=4096 / 1024, eager and graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd

dev = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def graphed(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


for prec in os.environ.get("PRECS", "fp16x3f,fp16x3,bf16x3,fp32,bf16,fp16").split(","):
    bnn_amd.set_precision(prec)
    torch.manual_seed(0)
    with torch.no_grad():
        net = bnn_amd.lrt.BayesianNetwork((784, 400, 400, 10)).to(dev).train()
        x = torch.rand(1024, 1, 28, 28, device=dev)
        f = lambda: (net(x, sample=True), net.kl())
        e = timeit(f); gr = timeit(graphed(f))
        print("configs[1] LRT 784-400-400-10 B=1024 %s: eager %.3f ms, graph %.3f ms (%.2f M samples/s)" % (prec, e, gr, 1024 / gr / 1e3))
        for B in (4096, 1024):
            ls = [bnn_amd.vd.BayesianLayer(a, b).to(dev) for a, b in ((3072, 4096), (4096, 4096), (4096, 10))]
            xv = torch.rand(B, 3072, device=dev)
            f2 = lambda: ls[2](torch.relu(ls[1](torch.relu(ls[0](xv)))))
            e = timeit(f2, 20); gr = timeit(graphed(f2), 20)
            flops = 4.0 * B * (3072 * 4096 + 4096 * 4096 + 4096 * 10)
            print("configs[4] VD 3072-4096-4096-10 B=%d %s: eager %.3f ms, graph %.3f ms (%.0f algorithmic TFLOP/s, %.2f M samples/s)"
                  % (B, prec, e, gr, flops / gr / 1e9, B / gr / 1e3))
