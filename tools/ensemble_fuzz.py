#!/usr/bin/env python3
"""Random networks (LRT / planar MNF, unaligned widths, ragged batches, 1-6 members) through the batched ensemble
(evaluate.ensemble_forward: one K3, one K1 and one GEMM launch per layer for all members) against the loop of fused single
forwards from the same Philox state: bitwise under fp32 / bf16x3; under the fp16 settings the batched form keeps the bf16
hi | lo operands (its member dimension exists in that format), so there the two agree to the formats' error (bar 1e-4 of max).
Usage: ensemble_fuzz.py [seed] [cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nb = 0
for it in range(N):
    dims = (random.choice([20, 64, 100, 784]), random.choice([33, 64, 80, 130, 256, 1200]), random.choice([17, 64, 96, 256]),
            random.choice([3, 10, 16, 24]))
    B = random.choice([1, 37, 64, 130, 1000])
    S = random.choice([1, 2, 3, 6])
    T = random.choice([1, 2, 4])
    prec = random.choice(["fp32", "bf16x3", "fp16x3", "fp16x3f"])
    kind = random.choice(["Planar", "LRT"])
    case = dict(it=it, kind=kind, dims=dims, B=B, S=S, T=T, prec=prec)
    torch.manual_seed(it)
    net = (bnn_amd.lrt.BayesianNetwork(dims) if kind == "LRT" else
           bnn_amd.mnf.BayesianNetwork(dims, T, z_flow_type="Planar", r_flow_type="Planar")).to(dev).eval()
    net.set_precision(prec)
    data = torch.rand(B, dims[0], device=dev)
    st = bnn_amd.ops.RngState.get(dev)
    with torch.no_grad():
        bnn_amd.manual_seed(3, 5)
        loop = torch.stack([net(data, sample=True) for _ in range(S)])
        off_loop = int(st.t[1])
        bnn_amd.manual_seed(3, 5)
        bat = bnn_amd.evaluate.ensemble_forward(net, data, S)
        off_bat = int(st.t[1])
    batched = bnn_amd.evaluate._batched_ok(net, data)
    nb += int(batched)
    same = torch.equal(bat, loop)
    err = float((bat - loop).abs().max() / loop.abs().max())
    fp16 = prec.startswith("fp16") and batched
    if off_loop != off_bat or (not fp16 and not same) or not err < 1e-4 or not bool(torch.isfinite(bat).all()):
        print("FAIL", case, dict(batched=batched, same=same, err=err, offsets=(off_loop, off_bat))); sys.exit(1)
    print("ok", case, "batched" if batched else "loop", "bitwise" if same else "%.1e" % err, flush=True)
print("%d random ensembles ok (%d through the batched launches)" % (N, nb))
