#!/usr/bin/env python3
"""Random three-layer MNF / planar networks through ONE training step's forward + backward on the HIP path (the network-level
training forward, the batched / deferred backward pieces of round 3: head dX / dW kernels, deferred column sums, batched V1,
register-form vector chains, the loss's logits hand-over) against fp64 autograd of the oracle on the same explicit draws:
loss and every parameter gradient (bar: 5e-4 of the largest entry, or 30 x what the oracle itself moves by when evaluated in fp32
-- printed next to the worst case).  Unaligned widths exercise the fall-backs (I % 4 != 0: generic weight-pass kernels; O * I
% 4 != 0: one head slab; > 4 flow steps: the LDS chain).  Families: planar / RNVP / MNF-flow
networks and the LRT network (no flows).  Usage: net_train_fuzz.py [seed] [cases] [Planar|RNVP|MNF|LRT|any]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import layers as L
from oracle import lbbnn_oracle as orc

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
FAMILY = sys.argv[3] if len(sys.argv) > 3 else "any"
ONLY = int(os.environ["NTF_ONLY"]) if "NTF_ONLY" in os.environ else None


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


worst, worst_at, kinks = 0.0, None, 0
for it in range(N):
    dims = (random.choice([20, 64, 100, 784]), random.choice([33, 64, 80, 130, 256]), random.choice([17, 64, 96, 256]),
            random.choice([3, 10, 16]))
    B = random.choice([64, 100, 130, 256])
    T = random.choice([1, 2, 3])
    prec = random.choice(["fp32", "bf16x3", "fp16x3", "fp16x3f"])
    defer = random.random() < 0.6
    fused_loss = random.random() < 0.6
    kind = random.choice(["Planar", "Planar", "RNVP", "MNF", "LRT"]) if FAMILY == "any" else FAMILY
    hard = random.random() < 0.3                    # flows and weights away from their small initial values
    if ONLY is not None:                            # NTF_ONLY=<it> [NTF_PREC= NTF_DEFER= NTF_FUSED=]: one case again, with overrides
        if it != ONLY:
            continue
        prec = os.environ.get("NTF_PREC", prec)
        defer = bool(int(os.environ.get("NTF_DEFER", int(defer))))
        fused_loss = bool(int(os.environ.get("NTF_FUSED", int(fused_loss))))
    torch.manual_seed(it)
    net = bnn_amd.lrt.BayesianNetwork(dims) if kind == "LRT" else bnn_amd.mnf.BayesianNetwork(dims, T, z_flow_type=kind, r_flow_type=kind)
    if hard:
        with torch.no_grad():
            for l in net._layers():
                l.weight_mu.mul_(5)
                if kind == "Planar":
                    for fl in (l.z_flow, l.r_flow):
                        for tr in fl.transforms:
                            tr.u.mul_(5.0); tr.w.mul_(5.0); tr.bias.mul_(5.0)
                if kind != "LRT":
                    l.q0_mean.add_(0.5)
    g = torch.Generator().manual_seed(900 + it)
    x = torch.rand(B, dims[0], generator=g)
    y = torch.randint(0, dims[3], (B,), generator=g)
    lay = [net.l1, net.l2, net.l3]
    if kind == "LRT":
        noises = [{"eps_out": torch.randn(B, l.out_features, generator=g)} for l in lay]
    else:
        noises = [{"eps_z": torch.randn(1, l.in_features, generator=g), "eps_out": torch.randn(B, l.out_features, generator=g),
                   "eps_z2": torch.randn(1, l.in_features, generator=g), "eps_act": torch.randn(l.out_features, generator=g)} for l in lay]
    if kind in ("RNVP", "MNF"):
        for l, n in zip(lay, noises):
            bern = lambda: torch.bernoulli(torch.full((l.in_features,), 0.5), generator=g)
            n.update(zmask=[bern() for _ in range(T)], zmask2=[bern() for _ in range(T)], rmask=[bern() for _ in range(T)])
    cast = lambda n, f: {k: ([f(m) for m in v] if isinstance(v, list) else f(v)) for k, v in n.items()}

    def oracle(dt):
        """BayesianNetwork.forward + kl() layer by layer (orc.lrt_network_forward / orc.mnf_network_forward spelled out, to see
        the hidden pre-activations): returns parameters with .grad, the loss, and per hidden layer the sorted |pre| / max|pre|
        of the elements closest to the ReLU kink."""
        P = [{k: v.detach().clone().to(dt).requires_grad_(True) for k, v in l.state_dict().items()} for l in lay]
        h, k, near = x.to(dt), 0, []
        for i, p in enumerate(P):
            if kind == "LRT":
                h, kk, _ = orc.lrt_forward(h, p, noises[i]["eps_out"].to(dt))
            else:
                h, kk, _ = orc.mnf_forward(h, p, orc.flow_from_state("z_flow", kind, p, T), orc.flow_from_state("r_flow", kind, p, T),
                                           cast(noises[i], lambda v: v.to(dt)))
            k = k + kk
            if i < 2:
                near.append(float((h.detach().abs() / h.detach().abs().max()).min()))
                h = torch.relu(h)
        ls = torch.nn.functional.nll_loss(torch.log_softmax(h, dim=1), y, reduction="sum") + k / 10
        ls.backward()
        return P, ls.detach(), near

    P, ref_loss, near = oracle(torch.float64)
    # conditioning yardstick: the SAME oracle evaluated in fp32 (a gradient that is a small difference of large terms -- r0_b1 /
    # r0_b2 are proportional to the mean of O tanh values, which can cancel to ~0 -- moves in any fp32 evaluation)
    P32, _, _ = oracle(torch.float32)
    net = net.to(dev).train()
    net.set_precision(prec)
    for l, n in zip(lay, noises):
        l.noise = cast(n, lambda v: v.to(dev))
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
    case = dict(it=it, kind=kind, hard=hard, dims=dims, B=B, T=T, prec=prec, defer=defer, fused_loss=fused_loss)
    bad = {n: (v, e32.get(n)) for n, v in errs.items() if not v < max(5e-4, 30 * e32.get(n, 0.0))}
    if errs[k] > worst and not bad:
        worst, worst_at = errs[k], (k, "fp32 oracle: %.2e" % e32.get(k, 0.0), case)
    if ONLY is not None:
        print(case, {n: "%.2e" % v for n, v in errs.items()})
    if bad:
        # a hidden pre-activation within the forward's own error of 0 sits on the ReLU kink: the 16-bit formats (forward error
        # ~1e-5 of max) may take the other branch there, and the "gradient" then differs by that element's whole contribution
        # (one row of dX, one row of dW) -- the function is not differentiable there; counted, not failed
        kink_tol = 1e-6 if prec == "fp32" else 5e-5
        if min(near) < kink_tol and all(n.startswith(("l1.", "l2.")) for n in bad):
            kinks += 1
            print("kink", case, "closest hidden pre-activation %.1e of max" % min(near), {n: "%.1e" % v[0] for n, v in bad.items()})
        else:
            print("FAIL", case, bad); sys.exit(1)
    del loss, out
print("%d random training steps ok (%d on a ReLU kink, see above); worst relative error %.2e at %s" % (N, kinks, worst, worst_at))
