#!/usr/bin/env python3
"""Would two row shards of the batch, run as two GEMM chains on two streams, fill each other's launch ramps and epilogues?
GEMM chain of the headline net (784->1200->1200->10, in-kernel noise, ReLU, log_softmax) on precomputed operands:
one stream x 4096 rows against two streams x (B0, 4096 - B0) rows, same total work.  HIP-graph replays, events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops
dev = torch.device("cuda:0")
st = ops.RngState.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
B = 4096
dims = [(784, 1200), (1200, 1200), (1200, 10)]
W = []
for (I, O) in dims:
    ld = ops.operand_ld(I)
    mu = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5); rho = -5 + torch.rand(O, I, device=dev, generator=g)
    lam = torch.rand(O, I, device=dev, generator=g)
    split = O > 16
    ew = torch.zeros(O, ld, device=dev); vw = torch.zeros(O, ld, device=dev)
    ops.weight_pass(mu, rho, lam, priors=bnn_amd.Priors(), e_w=ew, var_w=vw, split=split)
    W.append((ew, vw, torch.rand(O, device=dev, generator=g), 1e-4 * torch.rand(O, device=dev, generator=g), split))
x = torch.rand(B, 784, device=dev, generator=g)
h1 = torch.empty(B, 1200, device=dev); h2 = torch.empty(B, 1200, device=dev); out = torch.empty(B, 10, device=dev)

def chain(r0, r1):
    bufs = [x, h1, h2, out]
    for k, ((I, O), (ew, vw, bm, bv, split)) in enumerate(zip(dims, W)):
        ops.lrt_gemm(bufs[k][r0:r1], ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=(k < 2), out=bufs[k + 1][r0:r1],
                     split=split, log_softmax=(k == 2), row_offset=r0)

def timed(fn, n=200):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(n): gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

side = torch.cuda.Stream()
def one():
    chain(0, B)
def two(b0):
    def f():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        side.wait_event(ev)
        chain(0, b0)
        with torch.cuda.stream(side):
            chain(b0, B)
        ev2 = torch.cuda.Event(); ev2.record(side)
        cur.wait_event(ev2)
    return f
ref = None
one(); torch.cuda.synchronize(); ref = out.clone()
print("one stream, 4096 rows: %.1f us per chain" % timed(one))
for b0 in (2048, 2560, 3072, 1024):
    t = timed(two(b0))
    print("two streams, %d + %d rows: %.1f us per chain" % (b0, B - b0, t))
