"""Timing experiment: what does the headline forward cost when a layer's eps_out is READ (explicit draws) instead of drawn in
the GEMM epilogue?  Variants: none / l1 / l1+l2 explicit.  Interleaved launch-plan replays in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bnn_amd
from bnn_amd import graphs, ops
dev = torch.device("cuda:0")
B = 4096
torch.manual_seed(0)
x = torch.rand(B, 784, device=dev)
plans = {}
for name, which in (("drawn", ()), ("l1 read", ("l1",)), ("l1+l2 read", ("l1", "l2"))):
    net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    net.set_precision("fp16x3f")
    for w in which:
        l = getattr(net, w)              # (all draws of the layer or none: the three small vectors come along)
        l.noise = {"eps_out": torch.randn(B, 1200, device=dev), "eps_z": torch.randn(1, l.in_features, device=dev),
                   "eps_z2": torch.randn(1, l.in_features, device=dev), "eps_act": torch.randn(1200, device=dev)}
    with torch.no_grad():
        for _ in range(3):
            net(x, sample=True)
        plans[name] = (graphs.LaunchPlan(net, x, sample=True), net)
torch.cuda.synchronize()
res = {p: [] for p in plans}
for r in range(6):
    for p in plans:
        plan = plans[p][0]
        for _ in range(20):
            plan()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            plan()
        torch.cuda.synchronize()
        if r:
            res[p].append((time.perf_counter() - t0) / 200 * 1e3)
for p in plans:
    v = sorted(res[p])
    print("%-12s median %.4f ms  min %.4f ms  (%d calls per forward)" % (p, v[len(v) // 2], v[0], len(plans[p][0])))
# the stand-alone fill of one layer's draws
rng = ops.RngState.get(dev).t
for _ in range(5):
    ops.philox_normal(rng, ops.STREAM_EPS_OUT * 64 + 0, B, 1200)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    ops.philox_normal(rng, ops.STREAM_EPS_OUT * 64 + 0, B, 1200)
torch.cuda.synchronize()
print("stand-alone philox_normal fill of (4096, 1200): %.2f us per call (host loop, allocation included)" % ((time.perf_counter() - t0) / 200 * 1e6))
