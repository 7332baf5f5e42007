#!/usr/bin/env python3
"""Random-shape sweep of a whole MNF layer (forward, KL, every gradient) against fp64 autograd of the oracle:
unaligned I / O / B exercise the scalar paths of K1, K1b, the output-gradient kernel and the GEMM fallbacks.
FLOW=Planar (default) | RNVP | MNF | any: the flow family (dense flows: masks drawn per case, injected on both sides)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from oracle import lbbnn_oracle as orc

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
FLOW = os.environ.get("FLOW", "Planar")


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


worst, worst_at = 0.0, None
for it in range(N):
    B = random.choice([1, 3, 16, 33, 64, 100, 130])
    I = random.choice([5, 8, 33, 64, 100, 200, 257, 784, 1201])
    O = random.choice([1, 7, 10, 17, 40, 64, 130])
    T = random.choice([1, 2, 3])
    prec = random.choice(["fp32", "bf16x3", "fp16x3", "fp16x3f"])      # (shapes a 16-bit format does not take run the fp32 kernels)
    relu = random.random() < 0.5
    bnn_amd.set_precision(prec)
    torch.manual_seed(it)
    kind = random.choice(["Planar", "RNVP", "MNF"]) if FLOW == "any" else FLOW
    layer = bnn_amd.mnf.BayesianLinear(I, O, T, z_flow_type=kind, r_flow_type=kind)
    with torch.no_grad():
        if kind == "Planar":
            for fl in (layer.z_flow, layer.r_flow):
                for tr in fl.transforms:
                    tr.u.mul_(6.0); tr.w.mul_(6.0); tr.bias.mul_(6.0)
        layer.q0_mean.add_(1.0); layer.weight_mu.mul_(10)
    g = torch.Generator().manual_seed(1000 + it)
    noise = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g)}
    if kind != "Planar":
        bern = lambda: torch.bernoulli(torch.full((I,), 0.5), generator=g)
        noise.update(zmask=[bern() for _ in range(T)], zmask2=[bern() for _ in range(T)], rmask=[bern() for _ in range(T)])
    x = torch.rand(B, I, generator=g)
    wgt = torch.randn(B, O, generator=g)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train()
    layer.noise = {k: ([m.to(dev) for m in v] if isinstance(v, list) else v.to(dev)) for k, v in noise.items()}
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg, sample=True, _relu=relu)
    ((out * wgt.to(dev)).sum() + layer.kl / 60).backward()
    pc = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xc = x.double().requires_grad_(True)
    zf = orc.flow_from_state("z_flow", kind, pc, T); rf = orc.flow_from_state("r_flow", kind, pc, T)
    o, kl, _ = orc.mnf_forward(xc, pc, zf, rf, {k: ([m.double() for m in v] if isinstance(v, list) else v.double()) for k, v in noise.items()})
    if relu:
        o = torch.relu(o)
    ((o * wgt.double()).sum() + kl / 60).backward()
    errs = {"out": rel(out.detach().cpu().double(), o.detach()), "kl": abs(float(layer.kl.detach()) - float(kl.detach())) / abs(float(kl.detach())),
            "dx": rel(xg.grad.cpu().double(), xc.grad)}
    for name, prm in layer.named_parameters():
        ref = pc[name].grad
        if ref is not None and float(ref.abs().max()) > 0:
            errs[name] = rel(prm.grad.cpu().double(), ref)
    bad = {k: v for k, v in errs.items() if not v < (2e-4 if kind == "Planar" else 5e-4)}
    k = max(errs, key=errs.get)
    if errs[k] > worst:
        worst, worst_at = errs[k], (k, kind, prec, B, I, O, T)
    if bad:
        print("FAIL", dict(kind=kind, B=B, I=I, O=O, T=T, prec=prec, relu=relu), bad)
        sys.exit(1)
bnn_amd.set_precision("fp32")
print("%d random layers ok; worst relative error %.2e at %s" % (N, worst, worst_at))
