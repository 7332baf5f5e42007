#!/usr/bin/env python3
"""GPU-side cost of ONE launch of the bf16x3 GEMM's grid (480 workgroups x 256 threads, 72 KB of LDS each) that does nothing (lab build
`empty`): 50 launches captured in a HIP graph, replayed -- the host is out of the picture."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops
dev = torch.device("cuda:0")
st = ops.RngState.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
B, I, O = 4096, 1200, 1200
ld = ops.operand_ld(I)
x = torch.rand(B, I, device=dev, generator=g)
ew = torch.zeros(O, ld, device=dev); vw = torch.zeros(O, ld, device=dev)
bm = torch.rand(O, device=dev, generator=g); bv = 1e-4 * torch.rand(O, device=dev, generator=g)
out = torch.empty(B, O, device=dev)
def run(n):
    for _ in range(n):
        ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=True, out=out, split=True)
run(3); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    run(50)
for _ in range(3): gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): gr.replay()
e1.record(); torch.cuda.synchronize()
print("%.2f us per launch (GPU side, 500 launches in 10 graph replays)" % (e0.elapsed_time(e1) / 500 * 1e3))
