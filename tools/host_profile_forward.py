#!/usr/bin/env python3
"""Host-side cost of the eager no-grad ELBO forward of the headline net: wall time per call with the GPU kept busy (queue never
empty => what the host needs to ENQUEUE one forward), and a cProfile of where it goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(os.environ.get("B", "4096"))
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
x = torch.rand(B, 784, device=dev)
bnn_amd.set_precision(os.environ.get("PREC", "bf16x3"))
def step():
    out = net(x, sample=True)
    return out, net.kl()
with torch.no_grad():
    for _ in range(50): step()
    torch.cuda.synchronize()
    N = 400
    t0 = time.perf_counter()
    for _ in range(N): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.1f us per forward (host), %.1f us per forward incl. drain (GPU-bound if larger)" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N): step()
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
