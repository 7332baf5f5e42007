#!/usr/bin/env python3
"""GEMM-only driver for rocprofv3 passes: launches the dual-moment GEMM of the headline layers
(B x 784 x 1200 and B x 1200 x 1200) N times with in-kernel noise.  Usage:
    rocprofv3 --kernel-trace --pmc <counters> -d out -- python3 tools/profile_gemm.py [B] [N]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SPLIT = os.environ.get("SPLIT", "0") == "1"
FMT = int(os.environ.get("FMT", "0"))        # 2: row-scaled fp16, 3 + 3 products; 3: 3 + 1 products (as the forward runs them:
                                             # layer 1 on fp32 x writing planes, layer 2 on planes writing fp32)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
st = ops.RngState.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
res = []
for (I, O) in [(784, 1200), (1200, 1200)]:
    ld = ops.operand_ld(I)
    x = torch.rand(B, I, device=dev, generator=g)
    ew = torch.zeros(O, ld, device=dev); ew[:, :I] = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5)
    vw = torch.zeros(O, ld, device=dev); vw[:, :I] = 1e-4 * torch.rand(O, I, device=dev, generator=g)
    if SPLIT:   # operands in split format via the weight pass (values irrelevant for timing, must be finite)
        mu = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5); rho = -5 + torch.rand(O, I, device=dev, generator=g)
        lam = torch.rand(O, I, device=dev, generator=g)
        ops.weight_pass(mu, rho, lam, priors=bnn_amd.Priors(), e_w=ew, var_w=vw, split=True)
    bm = torch.rand(O, device=dev, generator=g); bv = 1e-4 * torch.rand(O, device=dev, generator=g)
    out = torch.empty(B, O, device=dev)
    if FMT >= 2:
        mu = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5); rho = -5 + torch.rand(O, I, device=dev, generator=g)
        lam = torch.rand(O, I, device=dev, generator=g)
        es, vs = torch.empty(O, device=dev), torch.empty(O, device=dev)
        ops.weight_pass(mu, rho, lam, priors=bnn_amd.Priors(), e_w=ew, var_w=vw, split=FMT, e_scale=es, v_scale=vs)
        first = I == 784
        xin = x if first else ops.format_x(x)
        pl = torch.zeros(B, ops.plane_ld(O), device=dev) if first else None
        kw = dict(I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=True, var1=(FMT == 3), x_planes=not first,
                  out=None if first else out, want_out=not first, out_planes=pl)
        for _ in range(3):
            ops.lrt_gemm16(xin, ew, vw, es, vs, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            ops.lrt_gemm16(xin, ew, vw, es, vs, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / N
        res.append((I, O, us, 4.0 * B * I * O / us / 1e6))
        continue
    for _ in range(3):
        ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=True, out=out, split=SPLIT)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=True, out=out, split=SPLIT)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    res.append((I, O, us, 4.0 * B * I * O / us / 1e6))
for r in res:
    print("B=%d I=%d O=%d  %.1f us/launch  %.1f TFLOP/s (algorithmic, 2 GEMMs)" % ((B,) + r))
