#!/usr/bin/env python3
"""Ring GEMM (LBBNN_GEMM_RING=1/2) against the 128x80 kernel: bitwise on several shapes, then the K sweep of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops
dev = torch.device("cuda:0")
st = ops.RngState.get(dev)
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for (B, I, O) in [(4096, 1200, 1200), (4096, 784, 1200), (4096, 1208, 1200), (4133, 1000, 1187), (8192, 264, 640), (2048, 2400, 1203)]:
    ld = ops.operand_ld(I)
    x = torch.rand(B, I, device=dev, generator=g) - 0.3
    ew = torch.zeros(O, ld, device=dev); vw = torch.zeros(O, ld, device=dev)
    mu = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5); rho = -5 + torch.rand(O, I, device=dev, generator=g)
    lam = torch.rand(O, I, device=dev, generator=g)
    ops.weight_pass(mu, rho, lam, priors=bnn_amd.Priors(), e_w=ew, var_w=vw, split=True)
    bm = torch.rand(O, device=dev, generator=g); bv = 1e-4 * torch.rand(O, device=dev, generator=g)
    eps = torch.randn(B, O, device=dev, generator=g)
    outs = {}
    for ring in ("0", "1", "2"):
        os.environ["LBBNN_GEMM_RING"] = ring
        a = ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, eps=eps, relu=True, split=True)
        b = ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, mean_only=True, split=True)
        off = st.t.clone()
        c = ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=False, split=True)
        torch.cuda.synchronize()
        outs[ring] = (a.clone(), b.clone(), c.clone())
    for ring in ("1", "2"):
        for k, name in enumerate(("eps", "mean_only", "philox")):
            same = torch.equal(outs["0"][k], outs[ring][k])
            if not same:
                bad += 1
                d = (outs["0"][k] - outs[ring][k]).abs().max().item()
            print("B=%d I=%d O=%d ring=%s %-9s %s" % (B, I, O, ring, name, "bitwise equal" if same else "DIFFERENT max %g" % d))
print("RESULT", "OK" if bad == 0 else "%d MISMATCHES" % bad)
