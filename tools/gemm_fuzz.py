#!/usr/bin/env python3
"""Random-shape sweep of the dual-moment GEMM (both precisions, mean-only and stochastic, ReLU on/off) against fp64:
catches tail / clamp / alignment mistakes that fixed test shapes miss."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
worst = {"fp32": 0.0, "bf16x3": 0.0}
for it in range(N):
    B = random.choice([1, 2, 7, 16, 33, 100, 127, 128, 129, 255, 300, 513, 1024])
    I = random.choice([8, 16, 24, 40, 64, 72, 96, 104, 200, 328, 784, 1000, 1200, 2048, 2056])
    O = random.choice([17, 20, 33, 64, 79, 80, 81, 96, 160, 161, 400, 1200])
    relu, mean_only = random.random() < 0.5, random.random() < 0.3
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, I, generator=g)
    mu = 0.2 * (torch.rand(O, I, generator=g) - 0.5); rho = -5 + 2 * torch.rand(O, I, generator=g); lam = torch.randn(O, I, generator=g)
    bm = torch.randn(O, generator=g); brho = -5 + torch.rand(O, generator=g); eps = torch.randn(B, O, generator=g)
    alpha, sigma = torch.sigmoid(lam.double()), torch.log1p(torch.exp(rho.double()))
    ew, vw = mu.double() * alpha, sigma ** 2 * alpha ** 2
    ref = x.double() @ ew.T + bm.double()
    if not mean_only:
        ref = ref + torch.sqrt((x.double() ** 2) @ vw.T + torch.log1p(torch.exp(brho.double())) ** 2) * eps.double()
    if relu:
        ref = torch.relu(ref)
    for prec in ("fp32", "bf16x3"):
        split = prec == "bf16x3" and ops.split_eligible(I, O)
        ld = ops.operand_ld(I)
        e_w, v_w = torch.empty(O, ld, device=dev), torch.empty(O, ld, device=dev)
        bvar = torch.empty(O, device=dev)
        ops.weight_pass(mu.to(dev), rho.to(dev), lam.to(dev), bias_rho=brho.to(dev), priors=bnn_amd.Priors(), e_w=e_w, var_w=v_w,
                        bias_var=bvar, split=split)
        out = ops.lrt_gemm(x.to(dev), e_w, v_w, I=I, O=O, bias_mean=bm.to(dev), bias_var=bvar, eps=None if mean_only else eps.to(dev),
                           relu=relu, mean_only=mean_only, split=split)
        err = float((out.cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        worst[prec] = max(worst[prec], err)
        if err > 1e-4 or not torch.isfinite(out).all():
            print("FAIL", prec, dict(B=B, I=I, O=O, relu=relu, mean_only=mean_only, split=split), err)
            sys.exit(1)
print("%d random shapes ok; worst relative error fp32 %.2e, bf16x3 %.2e" % (N, worst["fp32"], worst["bf16x3"]))
