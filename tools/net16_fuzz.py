#!/usr/bin/env python3
"""Random three-layer MNF networks through the fused no-grad fp16 forward (plane hand-over, head fold, per-layer choice of the
3 + 1 / 3 + 3 form, K tails, unaligned batches) against the fp64 oracle on the same explicit draws: log-probabilities and KL.
Usage: net16_fuzz.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from oracle import lbbnn_oracle as orc

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = {"fp16x3": 0.0, "fp16x3f": 0.0}
T = 2
for it in range(N):
    dims = (random.choice([64, 200, 784, 1000]), random.choice([64, 80, 96, 160, 256, 400, 1200]),
            random.choice([64, 80, 96, 256, 400, 1200]), random.choice([2, 10, 16, 17, 40]))
    B = random.choice([1, 7, 33, 128, 130, 512, 1000])
    prec = random.choice(["fp16x3", "fp16x3f"])
    torch.manual_seed(it)
    net = bnn_amd.mnf.BayesianNetwork(dims, T, z_flow_type="Planar", r_flow_type="Planar")
    g = torch.Generator().manual_seed(500 + it)
    x = torch.rand(B, dims[0], generator=g)
    layers = [net.l1, net.l2, net.l3]
    noises = [{"eps_z": torch.randn(B, l.in_features, generator=g), "eps_out": torch.randn(B, l.out_features, generator=g),
               "eps_z2": torch.randn(1, l.in_features, generator=g), "eps_act": torch.randn(l.out_features, generator=g)} for l in layers]
    P = [{k: v.detach().clone().double() for k, v in l.state_dict().items()} for l in layers]
    zf = [orc.flow_from_state("z_flow", "Planar", p, T) for p in P]
    rf = [orc.flow_from_state("r_flow", "Planar", p, T) for p in P]
    ref_out, ref_kl = orc.mnf_network_forward(x.double(), P, zf, rf, [{k: v.double() for k, v in n.items()} for n in noises])
    net = net.to(dev).train()
    net.set_precision(prec)
    for l, n in zip(layers, noises):
        l.noise = {k: v.to(dev) for k, v in n.items()}
    with torch.no_grad():
        out = net(x.to(dev), sample=True)
        kl = net.kl()
    # log-probabilities: absolute error against the spread of the logits row (a log_softmax output has no natural max-norm)
    err = float((out.cpu().double() - ref_out).abs().max() / ref_out.abs().max().clamp_min(1e-30))
    ekl = abs(float(kl) - float(ref_kl)) / abs(float(ref_kl))
    fmts = tuple(l._split_now for l in layers)
    worst[prec] = max(worst[prec], err)
    bar = 1e-4 if prec == "fp16x3f" else 5e-6
    if not (err < bar and ekl < 2e-5):
        print("FAIL", dict(dims=dims, B=B, prec=prec, formats=fmts), err, ekl); sys.exit(1)
print("%d random networks ok; worst relative error of the log-probabilities fp16x3 %.2e, fp16x3f %.2e" % (N, worst["fp16x3"], worst["fp16x3f"]))
