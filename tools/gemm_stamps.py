#!/usr/bin/env python3
"""Phase stamps of the bf16x3 GEMM's main loop (lab build: make -C tools/lab liblbbnn_gstamps.so, then
    LBBNN_LIB_PATH=tools/lab/liblbbnn_gstamps.so python3 tools/gemm_stamps.py [I] [O]).
Every 32nd workgroup records, for each of its 4 waves and each K step, the shader-clock time at: step start (t0), after the
24 ds_read_b128 are issued (t1), after the next step's LDS-DMA pieces are issued (t2), after the conversions and the 60 MFMAs
are issued (t3), after the barrier (t4)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bnn_amd
from bnn_amd import ops, _lib

B = 4096
I = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
O = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
dev = torch.device("cuda:0")
st = ops.RngState.get(dev)
g = torch.Generator(device=dev).manual_seed(0)
ld = ops.operand_ld(I)
x = torch.rand(B, I, device=dev, generator=g)
ew = torch.zeros(O, ld, device=dev); vw = torch.zeros(O, ld, device=dev)
mu = 0.02 * (torch.rand(O, I, device=dev, generator=g) - 0.5); rho = -5 + torch.rand(O, I, device=dev, generator=g)
lam = torch.rand(O, I, device=dev, generator=g)
ops.weight_pass(mu, rho, lam, priors=bnn_amd.Priors(), e_w=ew, var_w=vw, split=True)
bm = torch.rand(O, device=dev, generator=g); bv = 1e-4 * torch.rand(O, device=dev, generator=g)
out = torch.empty(B, O, device=dev)
for _ in range(5):
    ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=True, out=out, split=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.lrt_gemm(x, ew, vw, I=I, O=O, bias_mean=bm, bias_var=bv, rng=st.t, relu=True, out=out, split=True)
e1.record(); torch.cuda.synchronize()
print("launch %.1f us (stamped build)" % (e0.elapsed_time(e1) * 1e3))
SLOTS, STEPS, PH = 64, 48, 8
buf = np.zeros(SLOTS * STEPS * PH, dtype=np.uint32)
lib = _lib.lib()
lib.lbbnn_lab_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.lbbnn_lab_gemm_stamps(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
s = buf.reshape(SLOTS, STEPS, PH).astype(np.int64)
print("slot lin wave | xcc se cu simd | loop_cycles total_cycles nsteps | per-step avg: ds_issue dma_issue conv+mfma barrier gap | step")
rows = []
for sl in range(SLOTS):
    hw, xcc, loop, tot, ns, lin = s[sl, 0, :6]
    if ns == 0:
        continue
    ns = min(int(ns), STEPS - 1)
    t = s[sl, 1:ns + 1, :5]
    d_ds = (t[:, 1] - t[:, 0]).mean(); d_dma = (t[:, 2] - t[:, 1])[:-1].mean(); d_mf = (t[:, 3] - t[:, 2]).mean()
    d_bar = (t[:, 4] - t[:, 3]).mean(); gap = (t[1:, 0] - t[:-1, 4]).mean(); step = (t[1:, 0] - t[:-1, 0]).mean()
    cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
    rows.append((d_ds, d_dma, d_mf, d_bar, gap, step, loop, tot))
    print("%3d %4d %d | %d %d %2d %d | %6d %6d %2d | %5.0f %5.0f %5.0f %5.0f %4.0f | %5.0f" %
          (sl, lin, sl % 4, xcc & 0xF, se, cu, simd, loop, tot, ns, d_ds, d_dma, d_mf, d_bar, gap, step))
r = np.array(rows)
print("MEAN over waves: ds_issue %.0f  dma_issue %.0f  conv+mfma %.0f  barrier %.0f  gap %.0f  step %.0f | loop %.0f total %.0f cycles"
      % tuple(r.mean(0)))
# one wave's step-by-step trace
sl = 0
ns = min(int(s[sl, 0, 4]), STEPS - 1)
print("trace of slot 0 (t0, then deltas ds / dma / conv+mfma / barrier):")
for c in range(1, ns + 1):
    t = s[sl, c]
    print("  step %2d  t0 %6d  %5d %5d %5d %5d" % (c - 1, t[0], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]))
