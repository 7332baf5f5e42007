#!/usr/bin/env python3
"""Launch time of the dual-moment GEMM (bf16x3) against K at the headline B x O: separates the per-K-step cost from the fixed
part of a launch (dispatch, first fill, epilogue).  Usage: python3 tools/gemm_ksweep.py [B] [O]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bnn_amd
from bnn_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
O = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
dev = torch.device("cuda:0")
st = ops.RngState.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
rows = []
for I in [64, 256, 512, 784, 1024, 1200, 1600, 2400, 4800]:
    ld = ops.operand_ld(I)
    x = torch.rand(B, I, device=dev, generator=g)
    ew = torch.zeros(O, ld, device=dev); vw = torch.zeros(O, ld, device=dev)
    mu = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5); rho = -5 + torch.rand(O, I, device=dev, generator=g)
    lam = torch.rand(O, I, device=dev, generator=g)
    ops.weight_pass(mu, rho, lam, priors=bnn_amd.Priors(), e_w=ew, var_w=vw, split=True)
    bm = torch.rand(O, device=dev, generator=g); bv = 1e-4 * torch.rand(O, device=dev, generator=g)
    out = torch.empty(B, O, device=dev)
    def run(n):
        for _ in range(n):
            ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=True, out=out, split=True)
    run(20); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(100); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10.0)
    steps = (I + 31) // 32
    rows.append((I, steps, best))
    print("I=%5d steps=%3d  %.1f us/launch" % (I, steps, best), flush=True)
s = np.array([r[1] for r in rows], float); t = np.array([r[2] for r in rows], float)
A = np.vstack([s, np.ones_like(s)]).T
(slope, icpt), *_ = np.linalg.lstsq(A, t, rcond=None)
print("fit: %.3f us per K step + %.1f us fixed" % (slope, icpt))
