#!/usr/bin/env python3
"""CPU experiment (no GPU): output error of candidate 16-bit operand formats for the dual-moment GEMM against fp64, on the
headline layers' operands (reference init, x ~ U[0,1) for layer 1, ReLU activations of layer 1 for layer 2).  Every
candidate is emulated by rounding the operands the way the kernels would and accumulating in fp64 (the fp32 accumulate of
the MFMA adds ~2e-7 on top, first row).  Columns: max|err| / max|out|, and the element-wise violation of
|err| <= atol + 1e-4 |ref| with atol = 1e-6 max|out| (<= 1 passes the fp32-grade bar of the tests)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import lbbnn_oracle as orc

torch.manual_seed(0)
B = int(os.environ.get("B", "1024"))


def f16(t):
    return t.to(torch.float32).to(torch.float16).to(torch.float64)


def bf16_trunc(t):
    u = t.to(torch.float32).view(torch.int32) & -65536
    return u.view(torch.float32).to(torch.float64)


def bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def row_scale(w):
    """power-of-two scale per row so that max|w| lands in [2^13, 2^14)"""
    m = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(m))
    return torch.pow(2.0, 13 - e)


def split16(t):
    hi = f16(t)
    lo = f16(t - hi)
    return hi, lo


def pk_fma_f16(a, b, c):
    # one rounding to fp16 (fma)
    return f16(a * b + c)


def report(name, out, ref):
    err = (out - ref).abs()
    mx = ref.abs().max()
    viol = (err / (1e-6 * mx + 1e-4 * ref.abs())).max()
    print("  %-58s max|err|/max|out| %.2e   elementwise violation %.2f" % (name, float(err.max() / mx), float(viol)))


def layer(I, O, x, g, tag):
    p = orc.init_mnf_params(I, O, g)
    z = 1 + 0.1 * torch.randn(I, generator=g)
    eps = torch.randn(x.shape[0], O, generator=g).double()
    alpha = orc.alpha_of(p["lambdal"].double()); sigma = orc.sigma_of(p["weight_rho"].double())
    ew = (p["weight_mu"].double() * alpha * z.double()).float().double()      # the fp32 operands K1 computes
    vw = (sigma ** 2 * alpha ** 2).float().double()
    bm = p["bias_mu"].double(); bv = orc.sigma_of(p["bias_rho"].double()) ** 2
    x64 = x.double()

    def finish(m, v):
        return m + bm + torch.sqrt(v + bv) * eps

    ref = finish(x64 @ ew.T, (x64 ** 2) @ vw.T)
    print("%s  (B=%d, I=%d, O=%d; max|out| %.3f)" % (tag, x.shape[0], I, O, float(ref.abs().max())))
    report("fp32 accumulate (torch.mm)", (x @ ew.float().T).double() + bm + torch.sqrt(((x ** 2) @ vw.float().T).double() + bv) * eps, ref)
    # bf16x3 as shipped
    xh = bf16_trunc(x64); xl = bf16(x64 - xh); wh = bf16(ew); wl = bf16(ew - wh)
    s = (x * x).double(); sh = bf16_trunc(s); sl = bf16(s - sh); vh = bf16(vw); vl = bf16(vw - vh)
    report("bf16x3 (3+3), shipped", finish(xh @ wh.T + xh @ wl.T + xl @ wh.T, sh @ vh.T + sh @ vl.T + sl @ vh.T), ref)
    # fp16 formats, row-scaled weights
    se, sv = row_scale(ew), row_scale(vw)
    eh, el = split16(ew * se); vh, vl = split16(vw * sv)
    xh, xl = split16(x64)
    sh, sl = split16(s)
    mean3 = (xh @ eh.T + xh @ el.T + xl @ eh.T) / se.T
    report("fp16 3+3 (s from fp32 x^2)", finish(mean3, (sh @ vh.T + sh @ vl.T + sl @ vh.T) / sv.T), ref)
    report("fp16 3+2w (sh.vh + sh.vl)", finish(mean3, (sh @ vh.T + sh @ vl.T) / sv.T), ref)
    report("fp16 3+2x (sh.vh + sl.vh)", finish(mean3, (sh @ vh.T + sl @ vh.T) / sv.T), ref)
    report("fp16 3+1  (sh.vh)", finish(mean3, (sh @ vh.T) / sv.T), ref)
    # s from the fp16 planes by packed fp16 math: t = xh*xl (1 rounding), sh' = fma(xh, xh, 2t) (1 rounding); x pre-scaled by 2^-4
    k = 2.0 ** -4
    t = f16(xh * k * xl * k)
    shp = pk_fma_f16(xh * k, xh * k, 2 * t)
    report("fp16 3+1, sh = pk_fma_f16(xh,xh,2 xh xl) * 2^-8", finish(mean3, (shp @ vh.T) / sv.T / k / k), ref)
    report("fp16 3+2w, same sh", finish(mean3, (shp @ vh.T + shp @ vl.T) / sv.T / k / k), ref)
    sh1 = f16(xh * k * xh * k)
    report("fp16 3+1, sh = pk_mul_f16(xh,xh) * 2^-8 (xl ignored)", finish(mean3, (sh1 @ vh.T) / sv.T / k / k), ref)
    # second-order: sl' = residual of the packed product: fma(xh,xh,-sh') + 2t  (exactly representable residual pieces)
    slp = f16((xh * k) * (xh * k) + 2 * t - shp)       # what pk_fma(xh,xh,-sh')+2t would give up to one rounding
    report("fp16 3+3, sh/sl by packed fp16 math (3 pk ops + 1)", finish(mean3, (shp @ vh.T + shp @ vl.T + slp @ vh.T) / sv.T / k / k), ref)
    report("fp16 2+1 (xh.eh + xl.eh | sh.vh)  [mean without el]", finish((xh @ eh.T + xl @ eh.T) / se.T, (sh @ vh.T) / sv.T), ref)
    return torch.relu(ref).float()


g = torch.Generator().manual_seed(0)
x = torch.rand(B, 784, generator=g)
h1 = layer(784, 1200, x, g, "layer 784->1200")
h2 = layer(1200, 1200, h1, g, "layer 1200->1200 (x = ReLU activations of layer 1)")
