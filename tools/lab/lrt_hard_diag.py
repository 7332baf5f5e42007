"""Diagnosis of the net_train_fuzz LRT failure (seed 15, it 18): layer-by-layer, fp32 kernels against bf16x3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bnn_amd
dev = torch.device("cuda:0")
dims, B = (20, 80, 256, 3), 256
SCALE = float(os.environ.get("SCALE", "5"))
torch.manual_seed(18)
net = bnn_amd.lrt.BayesianNetwork(dims)
with torch.no_grad():
    for l in net._layers():
        l.weight_mu.mul_(SCALE)
g = torch.Generator().manual_seed(918)
x = torch.rand(B, dims[0], generator=g)
y = torch.randint(0, dims[3], (B,), generator=g)
lay = [net.l1, net.l2, net.l3]
noises = [{"eps_out": torch.randn(B, l.out_features, generator=g)} for l in lay]
net = net.to(dev).train()
for l, n in zip(lay, noises):
    l.noise = {k: v.to(dev) for k, v in n.items()}
res = {}
for prec in ("fp32", "bf16x3"):
    net.set_precision(prec)
    net.zero_grad()
    xx = x.to(dev).requires_grad_(True)
    h1 = net.l1(xx, sample=True, _relu=True); h1.retain_grad()
    h2 = net.l2(h1, sample=True, _relu=True); h2.retain_grad()
    o = net.l3(h2, sample=True); o.retain_grad()
    lp = torch.log_softmax(o, dim=1)
    loss = torch.nn.functional.nll_loss(lp, y.to(dev), reduction="sum") + (net.l1.kl + net.l2.kl + net.l3.kl) / 10
    loss.backward()
    r = {"h1": h1.detach(), "h2": h2.detach(), "o": o.detach(), "g_o": o.grad, "g_h2": h2.grad, "g_h1": h1.grad, "g_x": xx.grad}
    for n, p in net.named_parameters():
        r["d " + n] = p.grad.clone()
    res[prec] = r
    print(prec, "splits", [l._split_now for l in lay], "loss %.6f" % float(loss))
for k in res["fp32"]:
    a, b = res["bf16x3"][k].double(), res["fp32"][k].double()
    d = (a - b).abs()
    i = int(d.argmax())
    print("%-18s max|fp32| %.3e  rel diff %.2e   (at flat %d: %.6e vs %.6e)" % (k, float(b.abs().max()), float(d.max() / b.abs().max()), i,
          float(a.flatten()[i]), float(b.flatten()[i])))
