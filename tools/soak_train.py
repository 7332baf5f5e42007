import os, sys, torch
sys.path.insert(0, ".")
import bnn_amd
dev = torch.device("cuda:0")
bnn_amd.set_precision(os.environ.get("PREC", "fp16x3f"))
torch.manual_seed(0)
FLOW = os.environ.get("FLOW", "Planar")       # Planar (headline) | RNVP (reference default) | MNF
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type=FLOW, r_flow_type=FLOW).to(dev).train()
opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
# a learnable synthetic task: labels = argmax of a fixed random projection of the input
g = torch.Generator(device=dev).manual_seed(1)
proj = torch.randn(784, 10, device=dev, generator=g)
x = torch.rand(4096, 1, 28, 28, device=dev, generator=g)
y = (x.view(-1, 784) @ proj).argmax(1)
lf = lambda n, a, b: torch.nn.functional.nll_loss(n(a, sample=True), b, reduction="sum") + n.kl() / 15
step = bnn_amd.graphs.make_graphed_train_step(net, opt, lf, x, y)
losses = []
for it in range(2000):
    l = step(x, y)
    if it % 200 == 0 or it == 1999:
        losses.append(float(l))
        assert l.isfinite(), it
print("losses every 200 steps:", ["%.0f" % v for v in losses])
net.eval()
with torch.no_grad():
    acc = float((net(x, sample=False).argmax(1) == y).float().mean())
print("train-set accuracy of the posterior-mean network after 2000 graphed steps: %.3f" % acc)
for n_, p in net.named_parameters():
    assert torch.isfinite(p).all(), n_
print("all parameters finite; rng offset:", int(bnn_amd.ops.RngState.get(dev).t[1]))
print("precision %s, flows %s, device memory allocated %.0f MB (max %.0f MB)" % (bnn_amd.get_precision(), FLOW, torch.cuda.memory_allocated() / 2 ** 20,
                                                                                 torch.cuda.max_memory_allocated() / 2 ** 20))
