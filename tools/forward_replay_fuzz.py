#!/usr/bin/env python3
"""Random networks (every family, unaligned widths, ragged batches, all four arithmetics) through the fused no-grad forward with
IN-KERNEL noise, four ways from the same Philox {seed, offset}: eager, as two row shards with set_row_offset (the data-parallel
contract), as a recorded graphs.LaunchPlan, as a HIP-graph replay -- outputs and KL bitwise equal.
Usage: forward_replay_fuzz.py [seed] [cases]"""
import gc, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import graphs

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for it in range(N):
    dims = (random.choice([20, 64, 100, 784]), random.choice([33, 64, 80, 130, 256, 1200]), random.choice([17, 64, 96, 256]),
            random.choice([3, 10, 16]))
    B = random.choice([2, 64, 100, 130, 256, 1000])
    T = random.choice([1, 2, 3])
    prec = random.choice(["fp32", "bf16x3", "fp16x3", "fp16x3f"])
    kind = random.choice(["Planar", "Planar", "RNVP", "MNF", "LRT"])
    train = random.random() < 0.7
    case = dict(it=it, kind=kind, dims=dims, B=B, T=T, prec=prec, train=train)
    torch.manual_seed(it)
    net = (bnn_amd.lrt.BayesianNetwork(dims) if kind == "LRT" else
           bnn_amd.mnf.BayesianNetwork(dims, T, z_flow_type=kind, r_flow_type=kind)).to(dev)
    net.train(train)
    net.set_precision(prec)
    x = torch.rand(B, dims[0], device=dev)
    with torch.no_grad():
        bnn_amd.manual_seed(7 + it, 3)
        full = net(x, sample=True).clone()
        kl_full = net.kl().clone() if train else None
        cut = (B // 2) if B > 2 else 1
        parts = []
        for lo, hi in ((0, cut), (cut, B)):
            bnn_amd.manual_seed(7 + it, 3)
            net.set_row_offset(lo)
            parts.append(net(x[lo:hi].contiguous(), sample=True).clone())
            if train and not torch.equal(net.kl(), kl_full):
                print("FAIL (shard KL)", case); sys.exit(1)
        net.set_row_offset(0)
        ok_sh = torch.equal(torch.cat(parts), full)
        plan = graphs.LaunchPlan(net, x, sample=True)
        bnn_amd.manual_seed(7 + it, 3)
        o, k = plan()
        ok_plan = torch.equal(o, full) and (not train or torch.equal(k, kl_full))
        net(x, sample=True)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            og = net(x, sample=True)
            kg = net.kl() if train else None
        bnn_amd.manual_seed(7 + it, 3)
        gr.replay()
        torch.cuda.synchronize()
        ok_gr = torch.equal(og, full) and (not train or torch.equal(kg, kl_full))
    if not (ok_sh and ok_plan and ok_gr and bool(torch.isfinite(full).all())):
        print("FAIL", case, dict(shards=ok_sh, plan=ok_plan, graph=ok_gr)); sys.exit(1)
    print("ok", case, flush=True)
    del net, plan, gr, og, o
    gc.collect()
print("%d random forwards: shards == plan == graph replay == eager, bitwise" % N)
