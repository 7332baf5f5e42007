#!/usr/bin/env python3
"""Diagnostic for VERDICT r02 weak #1b: the fp32-mode gradient of r0_c at a (B, I, O, T) = (3, 5, 7, 3) planar layer was
1.21e-4 off in the round-2 fuzz campaign.  Repeats tools/layer_fuzz.py's case construction at that shape over SEEDS seeds
and prints, per seed, the error of every gradient AND the conditioning of the r0_c sum in fp64: dr0_c[i] = sum_o t[o, i]
with t = da_mu[o] (z_k mu alpha)[o, i] + 2 r0_c[i] da_var[o] var_w[o, i]; cond = max_i sum_o |t| / max_i |sum_o t|."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from oracle import lbbnn_oracle as orc

dev = torch.device("cuda:0")
B, I, O, T = [int(v) for v in os.environ.get("SHAPE", "3,5,7,3").split(",")]
SEEDS = int(os.environ.get("SEEDS", "40"))
bnn_amd.set_precision("fp32")


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


worst = {}
for it in range(SEEDS):
    torch.manual_seed(it)
    layer = bnn_amd.mnf.BayesianLinear(I, O, T, z_flow_type="Planar", r_flow_type="Planar")
    with torch.no_grad():
        for fl in (layer.z_flow, layer.r_flow):
            for tr in fl.transforms:
                tr.u.mul_(6.0); tr.w.mul_(6.0); tr.bias.mul_(6.0)
        layer.q0_mean.add_(1.0); layer.weight_mu.mul_(10)
    g = torch.Generator().manual_seed(1000 + it)
    noise = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g)}
    x = torch.rand(B, I, generator=g)
    wgt = torch.randn(B, O, generator=g)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train()
    layer.noise = {k: v.to(dev) for k, v in noise.items()}
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg, sample=True, _relu=False)
    ((out * wgt.to(dev)).sum() + layer.kl / 60).backward()
    pc = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xc = x.double().requires_grad_(True)
    zf = orc.flow_from_state("z_flow", "Planar", pc, T); rf = orc.flow_from_state("r_flow", "Planar", pc, T)
    o, kl, aux = orc.mnf_forward(xc, pc, zf, rf, {k: v.double() for k, v in noise.items()})
    ((o * wgt.double()).sum() + kl / 60).backward()
    errs = {}
    for name, prm in layer.named_parameters():
        ref = pc[name].grad
        if ref is not None and float(ref.abs().max()) > 0:
            errs[name] = rel(prm.grad.cpu().double(), ref)
    k = max(errs, key=errs.get)
    for n, v in errs.items():
        worst[n] = max(worst.get(n, 0.0), v)
    # conditioning of the r0_c gradient: element-wise max-norm error relative to the LARGEST entry, so what matters is how much
    # the largest entry itself cancels
    gr = pc["r0_c"].grad
    # ... and of the auxiliary mean m = mean_o tanh(act_mu + sqrt(act_var) eps) (LBBNN-GP-MF-MNF.py:218-221): the gradients of
    # r0_b1 / r0_b2 are proportional to it, so their relative error is the relative error of a sum of O tanh values
    ai = (aux["act_mu"] + aux["act_var"].sqrt() * noise["eps_act"].double()).detach()
    cond_m = float(torch.tanh(ai).abs().mean() / torch.tanh(ai).mean().abs().clamp_min(1e-300))
    print("seed %2d  worst %-22s %.2e   r0_c err %.2e  r0_b2 err %.2e  cond(m) %.1e  |dr0_c| = %s" % (
          it, k, errs[k], errs.get("r0_c", 0.0), errs.get("r0_b2", 0.0), cond_m, ["%.2e" % float(v) for v in gr.abs()][:4]))
print("worst per parameter over %d seeds:" % SEEDS)
for n, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]:
    print("  %-28s %.2e" % (n, v))
