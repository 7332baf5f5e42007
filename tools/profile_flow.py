#!/usr/bin/env python3
"""Latency breakdown of the single-workgroup kernels by varying what they do (HIP-event timing of back-to-back
launches; run under `rocprofv3 --kernel-trace --stats` for pure kernel durations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bnn_amd
from bnn_amd import ops, flows

dev = torch.device("cuda:0")
st = ops.RngState.get(dev)
I = 1200


def timeit(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("empty kernel (rng_advance): %.2f us" % timeit(lambda: st.advance(1)))
qm, lv = torch.randn(I, device=dev) * 0.1, torch.full((I,), -9.0, device=dev)
zf, zk, scal = torch.empty(I, device=dev), torch.empty(I, device=dev), torch.empty(8, device=dev)
eps1, eps2 = torch.randn(I, device=dev), torch.randn(I, device=dev)
for T in (0, 1, 2, 4, 8):
    fz = flows.PropagateFlow("Planar", I, T).to(dev)
    fr = flows.PropagateFlow("Planar", I, T).to(dev)
    for want_kl in (False, True):
        for given in (False, True):
            def k3():
                ops.mnf_flow_planar(qm, lv, fz.planar_params(), fr.planar_params() if want_kl else [],
                                    eps_fwd=eps1 if given else None, eps_kl=eps2 if given else None, rng=st.t, layer_id=1,
                                    z_fwd=zf, z_kl=zk, scal=scal, want_kl=want_kl)
            print("K3 planar I=%d T=%d want_kl=%d eps_given=%d: %.2f us" % (I, T, want_kl, given, timeit(k3)))
