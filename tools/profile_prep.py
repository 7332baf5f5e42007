#!/usr/bin/env python3
"""Times the x-independent kernels of the headline network in isolation (HIP events, back-to-back launches):
K1 weight pass per layer and batched, K3 flows, the whole lbbnn_layers_prepare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops, _lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
bnn_amd.set_precision(os.environ.get("PREC", "bf16x3"))
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
x = torch.rand(4096, 784, device=dev)
with torch.no_grad():
    for _ in range(3):
        net(x, sample=True)
st = ops.RngState.get(dev)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, l in (("l1", net.l1), ("l2", net.l2), ("l3", net.l3)):
    ws = l._workspace()
    split = l._split_now

    def k1():
        ops.weight_pass(l.weight_mu, l.weight_rho, l.lambdal, z_fwd=ws.z_fwd, z_kl=ws.z_kl, r0_c=l.r0_c,
                        bias_rho=l.bias_rho, priors=l.priors, e_w=ws.e_w, var_w=ws.var_w, kl_rows=ws.kl_rows,
                        act_mu=ws.act_mu, act_var=ws.act_var, bias_var=ws.bias_var, split=split)

    def k3():
        ops.mnf_flow_planar(l.q0_mean, l.q0_log_var, l.z_flow.planar_params(), l.r_flow.planar_params(), rng=st.t,
                            layer_id=l._layer_id, z_fwd=ws.z_fwd, z_kl=ws.z_kl, scal=ws.scal, want_kl=True)

    def k5():
        ops.kl_finalize(ws.kl_rows, l.bias_mu, l.bias_rho, priors=l.priors, act_mu=ws.act_mu, act_var=ws.act_var,
                        r0_b1=l.r0_b1, r0_b2=l.r0_b2, scal=ws.scal, rng=st.t, layer_id=l._layer_id, kl_layer=ws.kl)
    w = l.weight_mu.numel()
    t1 = timeit(k1)
    print("%s (%dx%d, split=%s): K1 %.1f us (%.2f TB/s @20B/w)  K3 %.1f us  K5 %.1f us"
          % (name, l.out_features, l.in_features, split, t1, w * 20 / t1 / 1e6, timeit(k3), timeit(k5)))

layers = [net.l1, net.l2, net.l3]
kls = torch.empty(4, device=dev)
descs = (_lib.LayerDesc * 3)()
keep = [l._fill_desc(descs[i], (True, True, i < 2), kls[i]) for i, l in enumerate(layers)]


def prep():
    _lib.check(_lib.lib().lbbnn_layers_prepare(descs, 3, st.t.data_ptr(), torch.cuda.current_stream().cuda_stream), "prep")


print("lbbnn_layers_prepare (K3 all | K1 all | K5 all): %.1f us" % timeit(prep))
