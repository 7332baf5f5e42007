#!/usr/bin/env python3
"""Lab timing of the 'P' GEMM experiment (x delivered as four bf16 planes, 128x160 tile, 8 waves): run with
LBBNN_LIB_PATH=tools/lab/liblbbnn_planes.so.  Values are random (timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bnn_amd
from bnn_amd import ops, _lib

dev = torch.device("cuda:0")
B, N = 4096, 50
st = ops.RngState.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
for (I, O) in [(768, 1200), (1184, 1200)]:          # multiples of 32: the lab kernel has no K-tail handling
    ld = ops.operand_ld(I)
    planes = (torch.randn(4, B, I, device=dev, generator=g) * 0.3).bfloat16().contiguous()
    xview = planes.view(torch.float32).reshape(B, 2 * I)
    ew = torch.zeros(O, ld, device=dev); vw = torch.zeros(O, ld, device=dev)
    mu = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5); rho = -5 + torch.rand(O, I, device=dev, generator=g)
    lam = torch.rand(O, I, device=dev, generator=g)
    ops.weight_pass(mu, rho, lam, priors=bnn_amd.Priors(), e_w=ew, var_w=vw, split=True)
    bm = torch.rand(O, device=dev, generator=g); bv = 1e-4 * torch.rand(O, device=dev, generator=g)
    out = torch.empty(B, O, device=dev)
    flags = ops.F_RELU | ops.F_SPLIT16

    def run():
        rc = _lib.lib().lbbnn_lrt_gemm(xview.data_ptr(), 2 * I, ew.data_ptr(), vw.data_ptr(), ld, bm.data_ptr(), bv.data_ptr(),
                                       None, None, st.t.data_ptr(), 3, 0, out.data_ptr(), O, B, I, O, flags,
                                       torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "lbbnn_lrt_gemm")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / N
    print("B=%d I=%d O=%d  %.1f us/launch  %.1f TFLOP/s algorithmic (finite out: %s)" % (B, I, O, us, 4.0 * B * I * O / us / 1e6, bool(torch.isfinite(out).all())))
