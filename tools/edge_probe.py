import sys
sys.path.insert(0, '.')
import torch, bnn_amd
from oracle import lbbnn_oracle as orc
dev = torch.device("cuda:0")
torch.manual_seed(0)
# empty batch, single row, odd sizes
for (B, I, O) in [(0, 33, 17), (1, 33, 17), (1, 1, 1), (3, 5, 1), (2, 1, 7), (5, 16383, 3)]:
    for kind in ("lrt", "mnf"):
        try:
            if kind == "lrt":
                l = bnn_amd.lrt.BayesianLinear(I, O).to(dev).train()
            else:
                l = bnn_amd.mnf.BayesianLinear(I, O, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
            x = torch.rand(B, I, device=dev)
            with torch.no_grad():
                out = l(x, sample=True)
            torch.cuda.synchronize()
            print(kind, (B, I, O), "ok", tuple(out.shape), float(l.kl), bool(torch.isfinite(out).all()))
        except Exception as e:
            print(kind, (B, I, O), "FAIL", type(e).__name__, str(e)[:150])
