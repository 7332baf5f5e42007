import sys, cProfile, pstats, torch
sys.path.insert(0, ".")
import bnn_amd
dev = torch.device("cuda:0")
bnn_amd.set_precision("bf16x3")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
x = torch.rand(4096, 1, 28, 28, device=dev); y = torch.randint(0, 10, (4096,), device=dev)
def step():
    net.zero_grad(set_to_none=True)
    loss = torch.nn.functional.nll_loss(net(x, sample=True), y, reduction="sum") + net.kl() / 15
    loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
