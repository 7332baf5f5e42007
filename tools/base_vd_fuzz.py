#!/usr/bin/env python3
"""Random shapes through the two "next" rows of SURVEY.md section 8: the baseline LBBNN layer (LBBNN-GP-MF.py: relaxed gate x
Gaussian weight sample, one GEMM, log-prior / log-posterior sums) in its full sample_elbo graph, and the variational-dropout
layer (variational_dropout.py).  Output, log-probabilities and every gradient from the HIP path against fp64 autograd of the
oracle on the same draws; all four arithmetics (shapes a 16-bit format does not take run the fp32 kernels).  A gradient's bar is
5e-4 of its largest entry, or 30 x what torch's own fp32 evaluation of the same graph moves by (the scalar prior parameters'
gradients are near-cancelling sums).
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
    errs, e32 = {}, {}
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
            def oracle(dt):
                Pd = {k: v.to(dt).clone().requires_grad_(True) for k, v in p.items()}
                xd_ = x.to(dt).requires_grad_(True)
                a_, cg_, tw_, tb_ = build(Pd, lambda t: t.to(dt))
                o_, lp_, lq_ = orc.base_forward(xd_, Pd, cg_, {"eps_w": eps_w.to(dt), "eps_b": eps_b.to(dt), "tau_w": tw_, "tau_b": tb_},
                                                mode="sample", gamma_alpha=a_)
                ((o_ * wgt.to(dt)).sum() + (lq_ - lp_) / 60).backward()
                return Pd, xd_, o_, lp_, lq_

            P64, x64, o, lp, lq = oracle(torch.float64)
            # conditioning yardstick: the same graph evaluated by torch in fp32 (the scalar gradients of pa / pb / a / b are sums of
            # O x I digamma-sized terms that can cancel to ~0: any fp32 evaluation moves there)
            P32 = oracle(torch.float32)[0]
            for name in P64:
                if P64[name].grad is not None and P32[name].grad is not None and float(P64[name].grad.abs().max()) > 0:
                    e32["d" + name] = float((P32[name].grad.double() - P64[name].grad).abs().max() / P64[name].grad.abs().max())
            errs = {"out": rel(out, o), "log_prior": rel(layer.log_prior, lp), "log_q": rel(layer.log_variational_posterior, lq),
                    "dx": rel(xd.grad, x64.grad)}
            for name, prm in layer.named_parameters():
                if P64[name].grad is not None and float(P64[name].grad.abs().max()) > 0:
                    errs["d" + name] = rel(prm.grad, P64[name].grad)
    finally:
        bnn_amd.set_precision("fp32")
    bad = {k: (v, e32.get(k)) for k, v in errs.items()
           if not v < (2e-5 if k in ("out", "log_prior", "log_q") else max(5e-4, 30 * e32.get(k, 0.0)))}
    k = max(errs, key=errs.get)
    if errs[k] > worst and not bad:
        worst, worst_at = errs[k], (k, "fp32 torch: %.1e" % e32[k] if k in e32 else "", case)
    if bad:
        print("FAIL", case, bad); sys.exit(1)
print("%d random baseline / variational-dropout layers ok; worst relative error %.2e at %s" % (N, worst, worst_at))
