#!/usr/bin/env python3
"""Error of the split-precision (bf16x3) GEMM path vs fp64, by shape and by component."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops
from oracle import lbbnn_oracle as orc
from tests.conftest import rel_err

dev = torch.device("cuda:0")
for (B, I, O) in [(128, 64, 80), (256, 256, 80), (1024, 400, 400), (1024, 784, 400), (4096, 784, 1200), (4096, 1200, 1200)]:
    g = torch.Generator().manual_seed(B + I + O)
    x = torch.rand(B, I, generator=g)
    p = orc.init_mnf_params(I, O, g)
    z = 1 + 0.1 * torch.randn(I, generator=g)
    d = {k: v.to(dev) for k, v in p.items()}
    ld = ops.operand_ld(I)
    eps = torch.randn(B, O, generator=g)
    alpha = orc.alpha_of(p["lambdal"].double()); sigma = orc.sigma_of(p["weight_rho"].double())
    ew = p["weight_mu"].double() * alpha * z.double(); vw = sigma ** 2 * alpha ** 2
    x64 = x.double()
    mean = x64 @ ew.T + p["bias_mu"].double()
    var = (x64 ** 2) @ vw.T + orc.sigma_of(p["bias_rho"].double()) ** 2
    ref = mean + torch.sqrt(var) * eps.double()
    res = {}
    for split in (False, True):
        e_w = torch.empty(O, ld, device=dev); var_w = torch.empty(O, ld, device=dev); bias_var = torch.empty(O, device=dev)
        ops.weight_pass(d["weight_mu"], d["weight_rho"], d["lambdal"], z_fwd=z.to(dev), bias_rho=d["bias_rho"],
                        priors=bnn_amd.Priors(), e_w=e_w, var_w=var_w, bias_var=bias_var, split=split)
        out = ops.lrt_gemm(x.to(dev), e_w, var_w, I=I, O=O, bias_mean=d["bias_mu"], bias_var=bias_var, eps=eps.to(dev), split=split)
        om = ops.lrt_gemm(x.to(dev), e_w, var_w, I=I, O=O, bias_mean=d["bias_mu"], mean_only=True, split=split)
        res[split] = (rel_err(out, ref), rel_err(om, mean))
    print("B=%d I=%d O=%d | fp32: out %.2e mean %.2e | split: out %.2e mean %.2e | max|out| %.2f sqrt(var) med %.3f"
          % (B, I, O, res[False][0], res[False][1], res[True][0], res[True][1], float(ref.abs().max()), float(var.sqrt().median())))
