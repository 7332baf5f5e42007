#!/usr/bin/env python3
"""Random networks (every family, unaligned widths -- parameter sizes that are not multiples of 4 floats, so the flat gradient
bucket's slices start at odd offsets) through the data-parallel step at world size 1: DataParallelELBO.make_graphed_step
(graph A: forward + backward + bucket pack | collective | graph B: Adam on the bucket) against the eager bucket step from the
same state and Philox seeds, 3 steps: losses and all parameters bitwise equal.  LBBNN_DP_FORCE_COLLECTIVE=1 with an nccl
group of one rank puts the RCCL call between the graphs.  Usage: dp_step_fuzz.py [seed] [cases]"""
import copy, gc, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import layers as L
from bnn_amd.parallel import DataParallelELBO

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for it in range(N):
    dims = (random.choice([20, 64, 100, 784]), random.choice([33, 64, 80, 130, 256]), random.choice([17, 64, 96, 256]),
            random.choice([3, 10, 16]))
    B = random.choice([64, 100, 130, 256])
    T = random.choice([1, 2, 3])
    prec = random.choice(["fp32", "bf16x3", "fp16x3", "fp16x3f"])
    kind = random.choice(["Planar", "Planar", "RNVP", "MNF", "LRT"])
    case = dict(it=it, kind=kind, dims=dims, B=B, T=T, prec=prec)
    torch.manual_seed(it)
    net = (bnn_amd.lrt.BayesianNetwork(dims) if kind == "LRT" else
           bnn_amd.mnf.BayesianNetwork(dims, T, z_flow_type=kind, r_flow_type=kind)).to(dev).train()
    net.set_precision(prec)
    init = copy.deepcopy(net.state_dict())
    x = torch.rand(B, dims[0], device=dev); y = torch.randint(0, dims[3], (B,), device=dev)
    res = []
    for mode in ("graph", "eager"):
        net.load_state_dict(init)
        opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
        dp = DataParallelELBO(net)
        if mode == "graph":
            step = dp.make_graphed_step(opt, x, y, 10, warmup=2)
            net.load_state_dict(init)
            for st in opt.state.values():
                st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
            for g in opt.param_groups:
                g["step_dev"].zero_()
        losses = []
        for s in range(3):
            bnn_amd.manual_seed(50 + s)
            if mode == "eager":
                opt.zero_grad(set_to_none=True)
                loss = dp.loss(net(x, sample=True), y, 10)
                with L.vector_backward_overlap():
                    loss.backward()
                dp.all_reduce_grads(unpack=False)
                opt.step(grads=dp.reduced_grads())
            else:
                loss = step(x, y)
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        res.append(({k: v.detach().clone() for k, v in net.named_parameters()}, losses))
        if mode == "graph":
            del step
    bad = [k for k in res[0][0] if not torch.equal(res[0][0][k], res[1][0][k])]
    if res[0][1] != res[1][1] or bad or not all(v == v for v in res[0][1]):
        print("FAIL", case, "losses graph / eager", res[0][1], res[1][1], "parameters that differ:", bad[:6]); sys.exit(1)
    print("ok", case, "bucket %d floats, loss %.3f -> %.3f" % (dp.bucket_numel(), res[0][1][0], res[0][1][-1]), flush=True)
    del net, opt, dp, res, loss
    gc.collect()
print("%d random data-parallel graphed steps bitwise equal to the eager bucket step" % N)
