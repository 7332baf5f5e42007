#!/usr/bin/env python3
"""Random-shape sweep of the row-scaled fp16 dual-moment GEMM (lbbnn_lrt_gemm_ex through ops.lrt_gemm16) against fp64:
unaligned B / O, K tails (I % 32 != 0), x as fp32 rows or as lbbnn_format_x planes, ReLU on / off, planes written for the
next layer, both product forms (3 + 3: max-norm bar 2e-6; 3 + 1: 4e-5 from I = 391 up, 2.5e-5 sqrt(1024 / I) below), in-kernel noise == the same draws handed in.
Usage: gemm16_fuzz.py [seed] [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops
from oracle import lbbnn_oracle as orc

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 80
# 3 + 1 products: the 2^-12 roundings of s and var_w average out as 1 / sqrt(I); the layers use that form from I = 256 up
# (ops.F16_VAR1_MIN_I), the raw entry point takes any I
bar = lambda fmt, I: 2e-6 if fmt == 2 else max(4e-5, 2.5e-5 * (1024.0 / I) ** 0.5)
worst = {2: 0.0, 3: 0.0}
for it in range(N):
    B = random.choice([1, 2, 7, 16, 33, 100, 127, 128, 129, 255, 300, 513, 1024])
    I = random.choice([8, 16, 24, 40, 64, 72, 96, 104, 200, 328, 784, 1000, 1200, 1272, 1280])
    O = random.choice([17, 20, 33, 64, 79, 80, 81, 96, 160, 161, 400, 1200])
    fmt = random.choice([2, 3])
    relu = random.random() < 0.5
    planes_in = random.random() < 0.5
    planes_out = O % 8 == 0 and random.random() < 0.5
    g = torch.Generator().manual_seed(10000 + it)
    p = orc.init_mnf_params(I, O, g)
    z = 1 + 0.1 * torch.randn(I, generator=g)
    d = {k: v.to(dev) for k, v in p.items()}
    ld = ops.operand_ld(I)
    ws = {k: torch.empty(O, ld, device=dev) for k in ("e_w", "var_w")}
    ws.update({k: torch.empty(O, device=dev) for k in ("bias_var", "e_scale", "v_scale")})
    ops.weight_pass(d["weight_mu"], d["weight_rho"], d["lambdal"], z_fwd=z.to(dev), bias_rho=d["bias_rho"], priors=bnn_amd.Priors(),
                    e_w=ws["e_w"], var_w=ws["var_w"], bias_var=ws["bias_var"], split=fmt, e_scale=ws["e_scale"], v_scale=ws["v_scale"])
    x = (4.0 * torch.rand(B, I, generator=g) - 1.0)
    eps = torch.randn(B, O, generator=g)
    alpha, sigma = orc.alpha_of(p["lambdal"].double()), orc.sigma_of(p["weight_rho"].double())
    ew, vw = p["weight_mu"].double() * alpha * z.double(), sigma ** 2 * alpha ** 2
    x64 = x.double()
    ref = x64 @ ew.T + p["bias_mu"].double() + torch.sqrt((x64 ** 2) @ vw.T + orc.sigma_of(p["bias_rho"].double()) ** 2) * eps.double()
    if relu:
        ref = torch.relu(ref)
    xd = x.to(dev)
    xin = ops.format_x(xd) if planes_in else xd
    pl = torch.zeros(B, ops.plane_ld(O), device=dev) if planes_out else None
    out, _ = ops.lrt_gemm16(xin, ws["e_w"], ws["var_w"], ws["e_scale"], ws["v_scale"], I=I, O=O, bias_mean=d["bias_mu"],
                            bias_var=ws["bias_var"], eps=eps.to(dev), relu=relu, var1=(fmt == 3), x_planes=planes_in, out_planes=pl)
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    worst[fmt] = max(worst[fmt], err)
    case = dict(B=B, I=I, O=O, fmt=fmt, relu=relu, planes_in=planes_in, planes_out=planes_out)
    if not err < bar(fmt, I):
        print("FAIL", case, err); sys.exit(1)
    if pl is not None and not torch.equal(pl, ops.format_x(out)):
        print("FAIL planes != format_x(out)", case); sys.exit(1)
    # in-kernel noise == the same draws handed in
    rng = torch.tensor([77 + it, 3, 0, 0], dtype=torch.int64, device=dev)
    e2 = ops.philox_normal(rng, 9, B, O, 5)
    a, _ = ops.lrt_gemm16(xin, ws["e_w"], ws["var_w"], ws["e_scale"], ws["v_scale"], I=I, O=O, bias_mean=d["bias_mu"],
                          bias_var=ws["bias_var"], eps=e2, relu=relu, var1=(fmt == 3), x_planes=planes_in)
    b, _ = ops.lrt_gemm16(xin, ws["e_w"], ws["var_w"], ws["e_scale"], ws["v_scale"], I=I, O=O, bias_mean=d["bias_mu"],
                          bias_var=ws["bias_var"], rng=rng, rng_stream=9, row_offset=5, relu=relu, var1=(fmt == 3), x_planes=planes_in)
    if not torch.equal(a, b):
        print("FAIL in-kernel noise != explicit draws", case); sys.exit(1)
print("%d random shapes ok; worst relative error fp16x3 %.2e, fp16x3f %.2e" % (N, worst[2], worst[3]))
