#!/usr/bin/env python3
"""Random shapes through the two "next" rows of SURVEY.md section 8: the baseline LBBNN layer (LBBNN-GP-MF.py: relaxed gate x
Gaussian weight sample, one GEMM, log-prior / log-posterior sums) in its full sample_elbo graph, and the variational-dropout
layer (variational_dropout.py).  Output, log-probabilities and every gradient from the HIP path against fp64 autograd of the
oracle on the same draws; all four arithmetics (shapes a 16-bit format does not take run the fp32 kernels).
Usage: base_vd_fuzz.py [seed] [cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from oracle import lbbnn_oracle as orc

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def rel(a, b):
    return float((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-30))


worst, worst_at = 0.0, None
for it in range(N):
    B = random.choice([1, 3, 16, 33, 64, 100, 130])
    I = random.choice([5, 8, 33, 64, 100, 200, 257, 784])
    O = random.choice([1, 7, 10, 17, 40, 64, 130, 400])
    prec = random.choice(["fp32", "bf16x3", "fp16x3", "fp16x3f"])
    which = random.choice(["base", "vd"])
    case = dict(it=it, which=which, B=B, I=I, O=O, prec=prec)
    torch.manual_seed(it)
    g = torch.Generator().manual_seed(300 + it)
    x = torch.randn(B, I, generator=g)
    wgt = torch.randn(B, O, generator=g)
    errs = {}
    bnn_amd.set_precision(prec)
    try:
        if which == "vd":
            layer = bnn_amd.vd.BayesianLayer(I, O).to(dev)
            zeta = torch.randn(B, O, generator=g)
            layer.noise = {"zeta": zeta.to(dev)}
            xd = x.to(dev).requires_grad_(True)
            out = layer(xd)
            (out * wgt.to(dev)).sum().backward()
            x64 = x.double().requires_grad_(True)
            th64 = layer.theta.detach().cpu().double().requires_grad_(True)
            ref = orc.vd_forward(x64, th64, layer.alpha.cpu().double(), zeta.double())
            (ref * wgt.double()).sum().backward()
            errs = {"out": rel(out, ref), "dx": rel(xd.grad, x64.grad), "dtheta": rel(layer.theta.grad, th64.grad)}
        else:
            layer = bnn_amd.base.BayesianLinear(I, O, 1)
            p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
            layer = layer.to(dev).train()
            logistic = torch.log(torch.rand(O, I, generator=g).clamp(1e-6, 1 - 1e-6))
            logistic = logistic - torch.log1p(-torch.exp(logistic))
            eps_w, eps_b = torch.randn(O, I, generator=g), torch.randn(O, generator=g)
            gam_w, gam_b = torch.rand(1, generator=g) + 0.5, torch.rand(O, generator=g) + 0.5

            def build(P, lib):
                alpha = 1 / (1 + torch.exp(-P["lambdal"]))
                cg = torch.sigmoid((torch.log(alpha) - torch.log1p(-alpha) + lib(logistic)) / 0.5)
                return alpha, cg, lib(gam_w) * P["weight_a"] / P["weight_b"], lib(gam_b) * P["bias_a"] / P["bias_b"]

            P = dict(layer.named_parameters())
            xd = x.to(dev).requires_grad_(True)
            alpha, cg, tau_w, tau_b = build(P, lambda t: t.to(dev))
            layer.gamma.alpha = alpha
            layer.noise = {"eps_w": eps_w.to(dev), "eps_b": eps_b.to(dev), "tau_w": tau_w, "tau_b": tau_b}
            out = layer(xd, cg, sample=True)
            ((out * wgt.to(dev)).sum() + (layer.log_variational_posterior - layer.log_prior) / 60).backward()
            P64 = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
            x64 = x.double().requires_grad_(True)
            a64, cg64, tw64, tb64 = build(P64, lambda t: t.double())
            o, lp, lq = orc.base_forward(x64, P64, cg64, {"eps_w": eps_w.double(), "eps_b": eps_b.double(), "tau_w": tw64, "tau_b": tb64},
                                         mode="sample", gamma_alpha=a64)
            ((o * wgt.double()).sum() + (lq - lp) / 60).backward()
            errs = {"out": rel(out, o), "log_prior": rel(layer.log_prior, lp), "log_q": rel(layer.log_variational_posterior, lq),
                    "dx": rel(xd.grad, x64.grad)}
            for name, prm in layer.named_parameters():
                if P64[name].grad is not None and float(P64[name].grad.abs().max()) > 0:
                    errs["d" + name] = rel(prm.grad, P64[name].grad)
    finally:
        bnn_amd.set_precision("fp32")
    bad = {k: v for k, v in errs.items() if not v < (2e-5 if k in ("out", "log_prior", "log_q") else 5e-4)}
    k = max(errs, key=errs.get)
    if errs[k] > worst:
        worst, worst_at = errs[k], (k, case)
    if bad:
        print("FAIL", case, bad); sys.exit(1)
print("%d random baseline / variational-dropout layers ok; worst relative error %.2e at %s" % (N, worst, worst_at))
