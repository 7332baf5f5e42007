#!/usr/bin/env python3
"""Wall time of one full training step (forward + backward + Adam) of the headline MNF/planar net, B=4096."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd

dev = torch.device("cuda:0")
bnn_amd.set_precision(os.environ.get("PREC", "bf16x3"))
torch.manual_seed(0)
FLOW = os.environ.get("FLOW", "Planar")      # Planar (headline) | RNVP (the reference default) | MNF
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type=FLOW, r_flow_type=FLOW).to(dev).train()
FUSED = os.environ.get("FUSED_ADAM", "1") == "1"
opt = (bnn_amd.optim.Adam if FUSED else torch.optim.Adam)(net.parameters(), lr=1e-3)
x = torch.rand(4096, 1, 28, 28, device=dev); y = torch.randint(0, 10, (4096,), device=dev)


FUSED_LOSS = os.environ.get("LOSS", "fused") == "fused"     # bnn_amd.elbo_loss (one launch) | the reference's spelling in torch ops


def elbo(out, kl, tgt=None):
    tgt = y if tgt is None else tgt
    if FUSED_LOSS:
        return bnn_amd.elbo_loss(out, tgt, kl, 15)
    return torch.nn.functional.nll_loss(out, tgt, reduction="sum") + kl / 15


def step(backward=True):
    net.zero_grad(set_to_none=True)
    out = net(x, sample=True)
    loss = elbo(out, net.kl())
    if backward:
        loss.backward()
        opt.step()
    return loss


MODE = sys.argv[1] if len(sys.argv) > 1 else "eager"
for name, bw in ((("forward only (autograd graph built)", False), ("forward + backward + Adam", True)) if MODE == "eager" else ()):
    for _ in range(3):
        step(bw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step(bw)
    torch.cuda.synchronize()
    print("%s: %.3f ms/step" % (name, (time.perf_counter() - t0) / n * 1e3))

# ---- the same step captured in a HIP graph (fresh process: `train_step_time.py graph`)
if MODE != "graph":
    sys.exit(0)
opt2 = bnn_amd.optim.Adam(net.parameters(), lr=1e-3) if FUSED else torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
lf = lambda n, a, b: elbo(n(a, sample=True), n.kl(), b)
gstep = bnn_amd.graphs.make_graphed_train_step(net, opt2, lf, x, y)
if os.environ.get("RESIDENT", "1") == "1":       # the batch lies in the graph's own input buffers (no device-to-device copy per step)
    x, y = gstep.inputs
for _ in range(3):
    gstep(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 50
for _ in range(n):
    gstep(x, y)                                  # (nothing between two replays: a .clone() of the loss here is a device copy per step)
torch.cuda.synchronize()
print("HIP-graph replay of forward + backward + Adam: %.3f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
losses = []
for _ in range(n):                               # the check, outside the timed loop
    losses.append(gstep(x, y).clone())
torch.cuda.synchronize()
ls = [float(l) for l in losses]
print("loss first/last: %.1f -> %.1f (must decrease; distinct values => fresh noise per replay: %s)" % (ls[0], ls[-1], len(set(ls)) > 40))
