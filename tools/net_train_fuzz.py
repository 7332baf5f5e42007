#!/usr/bin/env python3
"""Random three-layer MNF / planar networks through ONE training step's forward + backward on the HIP path (the network-level
training forward, the batched / deferred backward pieces of round 3: head dX / dW kernels, deferred column sums, batched V1,
register-form vector chains, the loss's logits hand-over) against fp64 autograd of the oracle on the same explicit draws:
loss and every parameter gradient (bar: 5e-4 of the largest entry, or 30 x what the oracle itself moves by when evaluated in fp32
-- printed next to the worst case).  Unaligned widths exercise the fall-backs (I % 4 != 0: generic weight-pass kernels; O * I
% 4 != 0: one head slab; > 4 flow steps: the LDS chain).  Usage: net_train_fuzz.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import layers as L
from oracle import lbbnn_oracle as orc

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


worst, worst_at = 0.0, None
for it in range(N):
    dims = (random.choice([20, 64, 100, 784]), random.choice([33, 64, 80, 130, 256]), random.choice([17, 64, 96, 256]),
            random.choice([3, 10, 16]))
    B = random.choice([64, 100, 130, 256])
    T = random.choice([1, 2, 3])
    prec = random.choice(["fp32", "bf16x3", "fp16x3", "fp16x3f"])
    defer = random.random() < 0.6
    fused_loss = random.random() < 0.6
    torch.manual_seed(it)
    net = bnn_amd.mnf.BayesianNetwork(dims, T, z_flow_type="Planar", r_flow_type="Planar")
    g = torch.Generator().manual_seed(900 + it)
    x = torch.rand(B, dims[0], generator=g)
    y = torch.randint(0, dims[3], (B,), generator=g)
    lay = [net.l1, net.l2, net.l3]
    noises = [{"eps_z": torch.randn(1, l.in_features, generator=g), "eps_out": torch.randn(B, l.out_features, generator=g),
               "eps_z2": torch.randn(1, l.in_features, generator=g), "eps_act": torch.randn(l.out_features, generator=g)} for l in lay]
    P = [{k: v.detach().clone().double().requires_grad_(True) for k, v in l.state_dict().items()} for l in lay]
    zf = [orc.flow_from_state("z_flow", "Planar", p, T) for p in P]
    rf = [orc.flow_from_state("r_flow", "Planar", p, T) for p in P]
    ref_out, ref_kl = orc.mnf_network_forward(x.double(), P, zf, rf, [{k: v.double() for k, v in n.items()} for n in noises])
    ref_loss = torch.nn.functional.nll_loss(ref_out, y, reduction="sum") + ref_kl / 10
    ref_loss.backward()
    # conditioning yardstick: the SAME oracle evaluated in fp32 (a gradient that is a small difference of large terms -- r0_b1 /
    # r0_b2 are proportional to the mean of O tanh values, which can cancel to ~0 -- moves in any fp32 evaluation)
    P32 = [{k: v.detach().clone().float().requires_grad_(True) for k, v in l.state_dict().items()} for l in lay]
    o32, k32 = orc.mnf_network_forward(x, P32, [orc.flow_from_state("z_flow", "Planar", p, T) for p in P32],
                                       [orc.flow_from_state("r_flow", "Planar", p, T) for p in P32], noises)
    (torch.nn.functional.nll_loss(o32, y, reduction="sum") + k32 / 10).backward()
    net = net.to(dev).train()
    net.set_precision(prec)
    for l, n in zip(lay, noises):
        l.noise = {k: v.to(dev) for k, v in n.items()}
    out = net(x.to(dev), sample=True)
    loss = bnn_amd.elbo_loss(out, y.to(dev), net.kl(), 10) if fused_loss else \
        torch.nn.functional.nll_loss(out, y.to(dev), reduction="sum") + net.kl() / 10
    if defer:
        with L.vector_backward_overlap():
            loss.backward()
    else:
        loss.backward()
    errs, e32 = {"loss": abs(float(loss.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach()))}, {}
    for li, l in enumerate(lay):
        for name, prm in l.named_parameters():
            r = P[li][name].grad
            if r is not None and float(r.abs().max()) > 0:
                errs["l%d.%s" % (li + 1, name)] = rel(prm.grad.cpu().double(), r)
                e32["l%d.%s" % (li + 1, name)] = rel(P32[li][name].grad.double(), r)
    k = max(errs, key=errs.get)
    case = dict(it=it, dims=dims, B=B, T=T, prec=prec, defer=defer, fused_loss=fused_loss)
    if errs[k] > worst:
        worst, worst_at = errs[k], (k, "fp32 oracle: %.2e" % e32.get(k, 0.0), case)
    bad = {n: (v, e32.get(n)) for n, v in errs.items() if not v < max(5e-4, 30 * e32.get(n, 0.0))}
    if bad:
        print("FAIL", case, bad); sys.exit(1)
    del loss, out
print("%d random training steps ok; worst relative error %.2e at %s" % (N, worst, worst_at))
