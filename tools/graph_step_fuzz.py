#!/usr/bin/env python3
"""Random networks (planar / RNVP / MNF-flow / LRT, unaligned widths, all four arithmetics, 1-3 transforms, the torch or the
fused loss) through the CAPTURED training step (graphs.make_graphed_train_step: forward + backward + Adam in one HIP graph,
vector chains deferred) against the same steps run eagerly from the same state and Philox seeds: losses and every parameter
after 3 steps must be bitwise the same -- whatever fall-back a shape takes has to be capture-safe and order-independent.
Usage: graph_step_fuzz.py [seed] [cases]"""
import copy, gc, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import graphs, layers as L

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
    fused_loss = random.random() < 0.6
    case = dict(it=it, kind=kind, dims=dims, B=B, T=T, prec=prec, fused_loss=fused_loss)
    torch.manual_seed(it)
    net = (bnn_amd.lrt.BayesianNetwork(dims) if kind == "LRT" else
           bnn_amd.mnf.BayesianNetwork(dims, T, z_flow_type=kind, r_flow_type=kind)).to(dev).train()
    net.set_precision(prec)
    init = copy.deepcopy(net.state_dict())
    x = torch.rand(B, dims[0], device=dev); y = torch.randint(0, dims[3], (B,), device=dev)
    if fused_loss:
        lf = lambda n, a, b: bnn_amd.elbo_loss(n(a, sample=True), b, n.kl(), 10)
    else:
        lf = lambda n, a, b: torch.nn.functional.nll_loss(n(a, sample=True), b, reduction="sum") + n.kl() / 10
    res = []
    for mode in ("graph", "eager"):       # graph first: an eager autograd graph alive on the default stream breaks a later capture
        net.load_state_dict(init)
        opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
        if mode == "graph":
            step = graphs.make_graphed_train_step(net, opt, lf, x, y, warmup=2)
            net.load_state_dict(init)                      # the warm-up steps moved the parameters: start over
            for st in opt.state.values():
                st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
            for g in opt.param_groups:
                g["step_dev"].zero_()
        losses = []
        for s in range(3):
            bnn_amd.manual_seed(50 + s)
            if mode == "eager":
                opt.zero_grad(set_to_none=True)
                loss = lf(net, x, y)
                with L.vector_backward_overlap():
                    loss.backward()
                opt.step()
            else:
                loss = step(x, y)
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        res.append(({k: v.detach().clone() for k, v in net.named_parameters()}, losses))
        if mode == "graph":
            del step
    bad = [k for k in res[0][0] if not torch.equal(res[0][0][k], res[1][0][k])]
    if res[0][1] != res[1][1] or bad:
        print("FAIL", case, "losses graph / eager", res[0][1], res[1][1], "parameters that differ:", bad[:6])
        sys.exit(1)
    if not all(v == v for v in res[0][1]):
        print("FAIL (nan)", case, res[0][1]); sys.exit(1)
    print("ok", case, "loss %.3f -> %.3f" % (res[0][1][0], res[0][1][-1]), flush=True)
    del net, opt, res, loss
    gc.collect()
print("%d random captured training steps bitwise equal to eager" % N)
