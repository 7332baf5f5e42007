"""Tensor-level wrappers over the C ABI (include/lbbnn.h).

Every function takes torch tensors that must already live on a HIP device (contiguous fp32),
enqueues on torch's current stream and returns without synchronising.  Nothing here computes
on the host; a CPU tensor or a missing library raises.
"""
import ctypes
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import Priors

F_RELU = 0x1
F_MEAN_ONLY = 0x2
F_SPLIT16 = 0x4
F_LOG_SOFTMAX = 0x8
F_SINGLE16 = 0x10
F_HALF16 = 0x20
F_F16S = 0x40
F_VAR1 = 0x80
F_XPLANES = 0x100

STREAM_EPS_OUT = 0
STREAM_EPS_Z = 1
STREAM_EPS_Z2 = 2
STREAM_EPS_ACT = 3
STREAM_EPS_W = 4
STREAM_EPS_B = 5
STREAM_MASK = 6
STREAM_ROW_MASK = 7


# GEMM arithmetic.  A name selects the operand FORMAT of the forward's dual-moment GEMM:
#   "fp32"     exact fp32 MFMA (v_mfma_f32_16x16x4_f32): the reference's own precision
#   "fp16x3"   row-scaled fp16 hi + lo operands, 3 + 3 products (LBBNN_F_F16S): 2-3e-8 of max|out| against fp64 -- tighter than an
#              fp32-accumulate torch.mm; the contract-grade 16-bit mode
#   "fp16x3f"  the same operands, ONE product for the variance GEMM (LBBNN_F_VAR1, 3 + 1): 1.4-1.8e-5 (contract 1e-4), 4 MFMAs
#              per tile step instead of 6; x^2 from the fp16 planes by packed fp16 math
#   "bf16x3"   bf16 hi + lo operands, 3 + 3 products (round 1-2 headline): 2.7e-6
#   "bf16" / "fp16"   ONE product per moment: reduced-precision modes OUTSIDE the contract (2e-3 / 3e-4), own tolerances
# The backward products and every mean-only product stay on the bf16x3 kernels whatever 16-bit mode is chosen.
# The process-wide default below applies to every module that has no ``precision`` attribute of its own:
# ``net.set_precision(name)`` (layers._NetworkBase) / ``layer.precision = name`` select per network / per layer.
PRECISIONS = ("fp32", "bf16x3", "bf16", "fp16", "fp16x3", "fp16x3f")
_PRECISION = "fp32"


def set_precision(name: str):
    """Process-wide DEFAULT precision (modules with their own ``precision`` attribute ignore it)."""
    global _PRECISION
    if name not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (PRECISIONS,))
    _PRECISION = name


def get_precision(module=None) -> str:
    """The precision in force for ``module`` (its own ``precision`` attribute if set, else the process default)."""
    own = getattr(module, "precision", None) if module is not None else None
    if own is not None:
        if own not in PRECISIONS:
            raise ValueError("precision must be one of %s" % (PRECISIONS,))
        return own
    return _PRECISION


def split_precision(module=None) -> bool:
    """Do the GEMMs of ``module`` take 16-bit operand planes (every mode but "fp32")?"""
    return get_precision(module) != "fp32"


def f16s_precision(module=None) -> bool:
    return get_precision(module) in ("fp16x3", "fp16x3f")


def split_eligible(I: int, O: int) -> bool:
    """Shapes the split-precision kernels accept (see lbbnn_lrt_gemm, LBBNN_F_SPLIT16); operands and x must also stay
    under 2 GiB each (32-bit buffer offsets), which the callers' batch sizes do by orders of magnitude."""
    return O > 16 and I % 8 == 0 and (I % 32 == 0 or operand_ld(I) - I >= 8) and O * operand_ld(I) * 4 < 0x7FFFFFF0


F16_VAR1_MIN_I = 256          # "fp16x3f": rows shorter than this keep three variance products (layers._split)


def f16s_eligible(I: int, O: int) -> bool:
    """Shapes the row-scaled fp16 format takes: what the split kernels take, and rows of at most 1280 weights (the weight
    pass scales a row by its maximum, which it holds in registers: weight_pass.hip)."""
    return split_eligible(I, O) and operand_ld(I) <= 1280


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor], name: str = "tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("bnn_amd: %s is on %s; the HIP path needs a GPU tensor (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32 and t.dtype != torch.int64:
        raise RuntimeError("bnn_amd: %s has dtype %s, expected float32" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("bnn_amd: %s must be contiguous" % name)
    return t.data_ptr()


def _ptr_rows(t: torch.Tensor, name: str) -> int:
    """Pointer of a 2-D fp32 GPU tensor whose rows are dense (stride(1) == 1); row stride is free."""
    if not t.is_cuda:
        raise RuntimeError("bnn_amd: %s is on %s; the HIP path needs a GPU tensor (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("bnn_amd: %s must be a 2-D float32 tensor with dense rows" % name)
    return t.data_ptr()


def operand_ld(I: int) -> int:
    return ((I + 31) // 32) * 32


# ----------------------------------------------------------------------------------------- RNG state
class RngState:
    """Per-device {seed, offset} pair in device memory, read by the kernels (graph-replay safe).

    The seed follows ``torch.initial_seed()`` (so ``torch.manual_seed(i)`` in a training script
    reseeds the in-kernel noise too); the offset is advanced on the device after every layer call.
    """
    _states = {}
    _requested = None          # (seed, offset) of the last manual_seed(): applies to states created afterwards too

    def __init__(self, device: torch.device):
        self.device = device
        self.seed = None
        # words 0-1: the live {seed, offset}; words 2-3: the copy a fused forward's later kernels read
        # (lbbnn_layers_operands_snap)
        self.t = torch.zeros(4, dtype=torch.int64, device=device)
        req = RngState._requested
        self.reseed(torch.initial_seed(), req[1] if (req is not None and req[0] == torch.initial_seed()) else 0)

    def reseed(self, seed: int, offset: int = 0):
        self.seed = int(seed)
        s = self.seed & 0xFFFFFFFFFFFFFFFF
        if s >= 1 << 63:
            s -= 1 << 64
        self.t[:2].copy_(torch.tensor([s, int(offset)], dtype=torch.int64))

    @classmethod
    def get(cls, device: torch.device) -> "RngState":
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        st = cls._states.get(key)
        if st is None:
            st = cls(device)
            cls._states[key] = st
        elif st.seed != torch.initial_seed() and not torch.cuda.is_current_stream_capturing():
            st.reseed(torch.initial_seed())
        return st

    def advance(self, delta: int = 1):
        _lib.check(_lib.lib().lbbnn_rng_advance(self.t.data_ptr(), delta, _stream()), "lbbnn_rng_advance")


def manual_seed(seed: int, offset: int = 0):
    """Reseed the in-kernel noise of every device that has a state (also done implicitly when
    ``torch.manual_seed`` changes ``torch.initial_seed()``)."""
    torch.manual_seed(seed)
    RngState._requested = (int(seed), int(offset))
    for st in RngState._states.values():
        st.reseed(seed, offset)


def philox_normal(rng: torch.Tensor, stream_id: int, rows: int, cols: int, row_base: int = 0) -> torch.Tensor:
    """The exact N(0,1) values the kernels draw (2-D if rows > 0, else 1-D of length cols)."""
    out = torch.empty((rows, cols) if rows > 0 else (cols,), dtype=torch.float32, device=rng.device)
    _lib.check(_lib.lib().lbbnn_philox_normal(rng.data_ptr(), stream_id, row_base, rows, cols, out.data_ptr(),
                                              _stream()), "lbbnn_philox_normal")
    return out


# ----------------------------------------------------------------------------------------- K1
def weight_pass(mu, rho, lambdal, *, z_fwd=None, z_kl=None, r0_c=None, bias_rho=None,
                priors: Priors, e_w=None, var_w=None, kl_rows=None, act_mu=None, act_var=None,
                bias_var=None, split=False, e_scale=None, v_scale=None):
    """lbbnn_weight_pass.  Output tensors are caller-allocated (see LayerWorkspace).  split: False / 0 fp32 operands,
    True / 1 bf16 hi | lo, 2 row-scaled fp16 hi | lo (lbbnn_weight_pass_f16: e_scale / v_scale (O) receive the row scales), 3 the same
    with var_w as plain fp16 rows (hi part only: the operands of the 3 + 1 product form)."""
    O, I = mu.shape
    ld = operand_ld(I)
    if int(split) >= 2:
        rc = _lib.lib().lbbnn_weight_pass_f16(
            _ptr(mu, "weight_mu"), _ptr(rho, "weight_rho"), _ptr(lambdal, "lambdal"),
            _ptr(z_fwd), _ptr(z_kl), _ptr(r0_c), _ptr(bias_rho), ctypes.byref(priors),
            _ptr(e_w), _ptr(var_w), ld, _ptr(e_scale, "e_scale"), _ptr(v_scale, "v_scale"),
            _ptr(kl_rows), _ptr(act_mu), _ptr(act_var), _ptr(bias_var), O, I, F_VAR1 if int(split) == 3 else 0, _stream())
        _lib.check(rc, "lbbnn_weight_pass_f16")
        return
    rc = _lib.lib().lbbnn_weight_pass(
        _ptr(mu, "weight_mu"), _ptr(rho, "weight_rho"), _ptr(lambdal, "lambdal"),
        _ptr(z_fwd), _ptr(z_kl), _ptr(r0_c), _ptr(bias_rho), ctypes.byref(priors),
        _ptr(e_w), _ptr(var_w), ld, _ptr(kl_rows), _ptr(act_mu), _ptr(act_var), _ptr(bias_var),
        O, I, F_SPLIT16 if split else 0, _stream())
    _lib.check(rc, "lbbnn_weight_pass")


def output_grad(g_out, *, out=None, std=None, eps=None, rng=None, rng_stream: int = 0, row_offset: int = 0,
                relu: bool = False, gv_scale=None, want_g: bool = True, defer_sums=None):
    """lbbnn_output_grad.  Returns (gm, gv, gmT, gvT, g_sum, gv_sum); the gv* are None for a posterior-mean forward.
    gv_scale (O,): per-column factor on G_v (the variational-dropout alpha).
    defer_sums: a list -- the column-sum partials stay in the workspace and a job for reduce_partials_flush is appended to
    it; g_sum / gv_sum are returned as (still unwritten) tensors that the flush fills."""
    B, O = g_out.shape
    f = dict(dtype=torch.float32, device=g_out.device)
    if g_out.stride(1) != 1:
        g_out = g_out.contiguous()
    stoch = std is not None
    # want_g=False: the caller needs no input gradient (a first layer): G_m / G_v themselves are not written, only G^T and sums
    gm, gmT, g_sum = (torch.empty((B, O), **f) if want_g else None), torch.empty((O, B), **f), torch.empty(O, **f)
    gv = gvT = gv_sum = None
    if stoch:
        gv, gvT, gv_sum = (torch.empty((B, O), **f) if want_g else None), torch.empty((O, B), **f), torch.empty(O, **f)
    if B == 0:
        g_sum.zero_()
        if stoch:
            gv_sum.zero_()
        return gm, gv, gmT, gvT, g_sum, gv_sum
    a = _lib.OutGradArgs()
    a.g_out = _ptr_rows(g_out, "g_out")
    a.out = _ptr_rows(out, "out") if (relu and out is not None) else None
    a.std = _ptr_rows(std, "std") if stoch else None
    ldo = out.stride(0) if (relu and out is not None) else (std.stride(0) if stoch else O)
    if relu and stoch and out.stride(0) != std.stride(0):
        raise ValueError("bnn_amd: out and std must share a row stride")
    a.eps = _ptr(eps, "eps") if (stoch and eps is not None) else None
    a.rng = rng.data_ptr() if rng is not None else None
    work = torch.empty(_lib.lib().lbbnn_output_grad_workspace(B, O), **f)
    a.gm, a.gmT, a.work = (gm.data_ptr() if gm is not None else None), gmT.data_ptr(), work.data_ptr()
    a.g_sum = g_sum.data_ptr() if defer_sums is None else None
    if stoch:
        a.gv, a.gvT = (gv.data_ptr() if gv is not None else None), gvT.data_ptr()
        a.gv_sum = gv_sum.data_ptr() if defer_sums is None else None
    if defer_sums is not None:
        nbt = (B + 63) // 64
        defer_sums.append(dict(work=work, outs=(g_sum, gv_sum, None), block_stride=O, q_stride=nbt * O, nblk=nbt, ncols=O,
                               nq=2 if stoch else 1))
    a.row_offset, a.rng_stream = row_offset, rng_stream
    a.gv_scale = _ptr(gv_scale, "gv_scale") if (stoch and gv_scale is not None) else None
    a.B, a.O, a.ldg, a.ldo, a.relu = B, O, g_out.stride(0), ldo, 1 if relu else 0
    _lib.check(_lib.lib().lbbnn_output_grad(ctypes.byref(a), _stream()), "lbbnn_output_grad")
    return gm, gv, gmT, gvT, g_sum, gv_sum


def head_dx(gm, gv, wmT, wvT, x, *, C: int, I: int):
    """lbbnn_head_dx: dX (B, I) of a <= 16-class head from G_m / G_v (B, C) and the transposed fp32 operands [I][ld]."""
    B = gm.shape[0]
    out = torch.empty((B, I), dtype=torch.float32, device=gm.device)
    rc = _lib.lib().lbbnn_head_dx(_ptr_rows(gm, "gm"), _ptr_rows(gv, "gv") if gv is not None else None, gm.stride(0),
                                  _ptr(wmT), _ptr(wvT), wmT.stride(0), _ptr_rows(x, "x") if gv is not None else None,
                                  x.stride(0) if gv is not None else 0, out.data_ptr(), I, B, C, I, _stream())
    _lib.check(rc, "lbbnn_head_dx")
    return out



def head_dw(gm, gv, x, *, nslabs: int = 16):
    """lbbnn_head_dw: (S, C, I) split-K slabs of dW_m = G_m^T x and dW_v = G_v^T x^2 of a <= 16-class head, from the row-major
    gradients (B, C) and the row-major layer input (B, I); gv None: (dWm, None)."""
    B, C = gm.shape
    I = x.shape[1]
    S = max(1, min(int(nslabs), B))
    dWm = torch.empty((S, C, I), dtype=torch.float32, device=gm.device)
    dWv = torch.empty((S, C, I), dtype=torch.float32, device=gm.device) if gv is not None else None
    rc = _lib.lib().lbbnn_head_dw(_ptr_rows(gm, "gm"), _ptr_rows(gv, "gv") if gv is not None else None, gm.stride(0),
                                  _ptr_rows(x, "x"), x.stride(0), dWm.data_ptr(), dWv.data_ptr() if dWv is not None else None,
                                  B, C, I, S, _stream())
    _lib.check(rc, "lbbnn_head_dw")
    return dWm, dWv


def matmul_splitk(a, w_op, *, K: int, N: int, kchunk: int):
    """lbbnn_matmul_splitk: slabs out[z] = a[:, Kz] @ w[Kz, :] (w given as its bf16x3 operand [N][ld(K)]);
    returns (S, M, N) with S = ceil(K / kchunk)."""
    M = a.shape[0]
    S = (K + kchunk - 1) // kchunk
    out = torch.empty((S, M, N), dtype=torch.float32, device=a.device)
    rc = _lib.lib().lbbnn_matmul_splitk(_ptr_rows(a, "a"), a.stride(0), _ptr(w_op), operand_ld(K), out.data_ptr(), N,
                                        M, K, N, kchunk, _stream())
    _lib.check(rc, "lbbnn_matmul_splitk")
    return out


def dx_combine(gx, gxv, x):
    """lbbnn_dx_combine: gx += 2 * x * gxv in place (gx, gxv dense (B,I); x may have a row stride)."""
    B, I = gx.shape
    rc = _lib.lib().lbbnn_dx_combine(_ptr(gx, "gx"), _ptr(gxv, "gxv"), _ptr_rows(x, "x"), x.stride(0), B, I, _stream())
    _lib.check(rc, "lbbnn_dx_combine")
    return gx


def weight_pass_backward(mu, rho, lambdal, dWm, dWv=None, *, z_fwd=None, z_kl=None, r0_c=None, da_mu=None,
                         da_var=None, g_kl=None, priors: Priors, work: Optional[torch.Tensor] = None, defer_sums=None):
    """lbbnn_weight_pass_backward (K1b).  Returns (dmu, drho, dlambdal, dz_fwd, dz_kl, dr0_c); the three
    vector gradients are None when the corresponding input vector was not given.
    defer_sums: a list -- as in output_grad: the three column sums are finished by reduce_partials_flush."""
    O, I = mu.shape
    a = _lib.WpbArgs()
    a.mu, a.rho, a.lambdal = _ptr(mu, "weight_mu"), _ptr(rho, "weight_rho"), _ptr(lambdal, "lambdal")
    nsplit = dWm.shape[0] if dWm.dim() == 3 else 1           # (S,O,I): split-K slabs, added inside the kernel
    for t, name in ((dWm, "dWm"), (dWv, "dWv")):
        if t is not None and (tuple(t.shape[-2:]) != tuple(mu.shape) or not t.is_contiguous()
                              or (t.shape[0] if t.dim() == 3 else 1) != nsplit):
            raise ValueError("bnn_amd: %s must be a contiguous (O,I) or (S,O,I) tensor" % name)
    a.dWm, a.dWv = _ptr(dWm, "dWm"), _ptr(dWv)
    a.z_fwd, a.z_kl, a.r0_c = _ptr(z_fwd), _ptr(z_kl), _ptr(r0_c)
    a.da_mu, a.da_var, a.g_kl = _ptr(da_mu), _ptr(da_var), _ptr(g_kl)
    a.priors = priors
    f = dict(dtype=torch.float32, device=mu.device)
    dmu, drho, dlam = torch.empty_like(mu), torch.empty_like(mu), torch.empty_like(mu)
    dz_fwd = torch.empty(I, **f) if z_fwd is not None else None
    dz_kl = torch.empty(I, **f) if (z_kl is not None and (g_kl is not None or da_mu is not None)) else None
    dr0_c = torch.empty(I, **f) if (r0_c is not None and da_mu is not None) else None
    need = _lib.lib().lbbnn_weight_pass_backward_workspace(O, I)
    if work is None or work.numel() < need:
        work = torch.empty(need, **f)
    a.dmu, a.drho, a.dlambdal = dmu.data_ptr(), drho.data_ptr(), dlam.data_ptr()
    deferred = defer_sums is not None and (dz_fwd is not None or dz_kl is not None or dr0_c is not None)
    a.work = work.data_ptr()
    a.dz_fwd, a.dz_kl, a.dr0_c = (None, None, None) if deferred else (_ptr(dz_fwd), _ptr(dz_kl), _ptr(dr0_c))
    a.O, a.I = O, I
    a.nsplit, a.split_stride = nsplit, O * I
    _lib.check(_lib.lib().lbbnn_weight_pass_backward(ctypes.byref(a), _stream()), "lbbnn_weight_pass_backward")
    if deferred:
        ld = (I + 3) & ~3
        # dr0_c is RETURNED as a parameter gradient: autograd adopts it as .grad only while nobody else holds a reference
        # (layers.join_vector_backward checks the adoption), so the job keeps its address, not the tensor
        defer_sums.append(dict(work=work, outs=(dz_fwd, dz_kl, None), ptrs=(_ptr(dz_fwd), _ptr(dz_kl), _ptr(dr0_c)),
                               block_stride=3 * ld, q_stride=ld, nblk=(O + 7) // 8, ncols=I, nq=3))
    return dmu, drho, dlam, dz_fwd, dz_kl, dr0_c


def reduce_partials_flush(pending):
    """lbbnn_reduce_partials_batch: finish the column sums that output_grad / weight_pass_backward left as partials
    (``defer_sums``), up to 8 jobs per launch, in filing order; clears the list."""
    i = 0
    while i < len(pending):
        grp = pending[i:i + 8]
        arr = (_lib.ReduceJob * len(grp))()
        for k, j in enumerate(grp):
            arr[k].work = j["work"].data_ptr()
            for q in range(3):
                arr[k].out[q] = j["ptrs"][q] if "ptrs" in j else (j["outs"][q].data_ptr() if j["outs"][q] is not None else None)
            arr[k].block_stride, arr[k].q_stride = j["block_stride"], j["q_stride"]
            arr[k].nblk, arr[k].ncols, arr[k].nq = j["nblk"], j["ncols"], j["nq"]
        _lib.check(_lib.lib().lbbnn_reduce_partials_batch(arr, len(grp), _stream()), "lbbnn_reduce_partials_batch")
        i += 8
    pending.clear()


# ----------------------------------------------------------------------------------------- K2
# When set to a GemmEventLog, every lrt_gemm launch is bracketed by HIP events recorded on the launch stream and
# (B, I, O, start, end) is logged -- bench.py's per-kernel roofline pass (run OUTSIDE its timed region).
# torch.cuda.Event creates the HIP event lazily at its first record(): the constructor therefore records every pooled
# event once and synchronises, so no event is created (or first touched by the runtime) between two bracketed launches.
class GemmEventLog(list):
    """``every``: bracket only every n-th launch group of ``group`` launches (the event records themselves cost host
    time: 6 per step made an otherwise GPU-bound eager loop host-bound)."""

    def __init__(self, launches: int, group: int = 1, every: int = 1):
        super().__init__()
        self._pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for s, e in self._pool:
            s.record()
            e.record()
        torch.cuda.synchronize()
        self._group, self._every, self._n = group, every, 0

    def take(self):
        k = self._n
        self._n += 1
        if (k // self._group) % self._every:
            return None
        return self._pool.pop() if self._pool else None


GEMM_EVENTS = None


def lrt_gemm(x, e_w, var_w, *, I: int, O: int, bias_mean=None, bias_var=None, var_scale=None,
             eps=None, rng: Optional[torch.Tensor] = None, rng_stream: int = 0, row_offset: int = 0,
             relu: bool = False, mean_only: bool = False, log_softmax: bool = False,
             split: bool = False, out: Optional[torch.Tensor] = None, std_out: Optional[torch.Tensor] = None,
             finalize=None, half: bool = False, single: Optional[bool] = None):
    """lbbnn_lrt_gemm: out = x.e_w^T + b [+ sqrt(x^2.var_w^T + bv) * eps] [ReLU].
    single: ONE 16-bit product per moment (the reduced "bf16" / "fp16" modes; default: what the process-wide precision says).
    finalize = (layer descriptors, n, rng pointer for K5, kl_total pointer[, live rng pointer, advance]):
    lbbnn_lrt_gemm_finalize_adv -- the KL finalize of the whole network (n may be 0) and the forward's RNG advance ride in
    this launch."""
    if x.dim() != 2 or x.shape[1] != I:
        raise RuntimeError("bnn_amd: input must be (B,%d), got %s" % (I, tuple(x.shape)))
    B = x.shape[0]
    if out is None:
        out = torch.empty((B, O), dtype=torch.float32, device=x.device)
    if eps is not None and tuple(eps.shape) != (B, O):
        raise RuntimeError("bnn_amd: eps must be (%d,%d), got %s" % (B, O, tuple(eps.shape)))
    flags = ((F_RELU if relu else 0) | (F_MEAN_ONLY if mean_only else 0) | (F_LOG_SOFTMAX if log_softmax else 0)
             | (F_SPLIT16 if split else 0)
             | (F_SINGLE16 if (split and not mean_only and (single if single is not None else _PRECISION in ("bf16", "fp16"))) else 0)
             | (F_HALF16 if (half and split and not mean_only) else 0))
    if B == 0 and finalize is None:    # empty batch: (0,O) activations, as torch.mm gives; the KL side is unaffected
        return out
    if x.stride(1) != 1 or (x.stride(0) < I):
        x = x.contiguous()
    ev = GEMM_EVENTS.take() if GEMM_EVENTS is not None else None
    if ev is not None:
        ev[0].record()
    if finalize is not None:
        # (descs, n, fin_rng, kl_total[, rng_live, advance]): with the last two the piggy workgroup also advances the live offset
        live, adv = (finalize[4], finalize[5]) if len(finalize) > 4 else (None, 0)
        rc = _lib.lib().lbbnn_lrt_gemm_finalize_adv(
            _ptr_rows(x, "input"), x.stride(0), _ptr(e_w), _ptr(var_w), operand_ld(I),
            _ptr(bias_mean), _ptr(bias_var), _ptr(var_scale), _ptr(eps, "eps"),
            rng.data_ptr() if rng is not None else None, rng_stream, row_offset,
            out.data_ptr(), out.stride(0), _ptr(std_out, "std_out"), B, I, O, flags,
            finalize[0], finalize[1], finalize[2], finalize[3], live, adv, _stream())
    elif std_out is None:
        rc = _lib.lib().lbbnn_lrt_gemm(
            _ptr_rows(x, "input"), x.stride(0), _ptr(e_w), _ptr(var_w), operand_ld(I),
            _ptr(bias_mean), _ptr(bias_var), _ptr(var_scale), _ptr(eps, "eps"),
            rng.data_ptr() if rng is not None else None, rng_stream, row_offset,
            out.data_ptr(), out.stride(0), B, I, O, flags, _stream())
    else:
        rc = _lib.lib().lbbnn_lrt_gemm_train(
            _ptr_rows(x, "input"), x.stride(0), _ptr(e_w), _ptr(var_w), operand_ld(I),
            _ptr(bias_mean), _ptr(bias_var), _ptr(var_scale), _ptr(eps, "eps"),
            rng.data_ptr() if rng is not None else None, rng_stream, row_offset,
            out.data_ptr(), out.stride(0), _ptr(std_out, "std_out"), B, I, O, flags, _stream())
    _lib.check(rc, "lbbnn_lrt_gemm")
    if ev is not None:
        ev[1].record()
        GEMM_EVENTS.append((B, I, O, ev[0], ev[1]))
    return out


def plane_ld(n: int) -> int:
    return operand_ld(n)


def head_slab_floats(B: int, O: int) -> int:
    return int(_lib.lib().lbbnn_head_slab_floats(B, O))


def format_x(x, planes=None):
    """lbbnn_format_x: fp32 rows (B, I) -> fp16 hi | lo planes, a (B, plane_ld(I)) fp32-sized buffer (tail zero)."""
    B, I = x.shape
    if x.stride(1) != 1:
        x = x.contiguous()
    if planes is None:
        planes = torch.empty((B, plane_ld(I)), dtype=torch.float32, device=x.device)
    rc = _lib.lib().lbbnn_format_x(_ptr_rows(x, "input"), x.stride(0), planes.data_ptr(), planes.stride(0), B, I, _stream())
    _lib.check(rc, "lbbnn_format_x")
    return planes


def lrt_gemm16(x, e_w, var_w, e_scale, v_scale, *, I: int, O: int, bias_mean=None, bias_var=None, var_scale=None,
               eps=None, rng: Optional[torch.Tensor] = None, rng_stream: int = 0, row_offset: int = 0, relu: bool = False,
               var1: bool = False, x_planes: bool = False, out: Optional[torch.Tensor] = None, want_out: bool = True,
               out_planes: Optional[torch.Tensor] = None, std_out: Optional[torch.Tensor] = None, finalize=None,
               head=None):
    """lbbnn_lrt_gemm_ex on LBBNN_F_F16S operands: the dual-moment GEMM + sampling epilogue of lrt_gemm in the row-scaled fp16
    format.  x: fp32 rows (B, I), or with ``x_planes`` the (B, plane_ld(I)) plane buffer a previous call / format_x wrote.
    Outputs: fp32 ``out`` (allocated unless ``want_out`` is False) and / or ``out_planes`` (B, plane_ld(O)) for the next layer.
    ``finalize`` as in lrt_gemm.  ``head``: dict(e_w, var_w (C, operand_ld(O)) fp32 operands of the <= 16-class layer that follows,
    bias_mean, bias_var, eps (or None), rng_stream, out (B, C), slab (head_slab_floats(B, O)), log_softmax) -- the head's GEMM
    is folded into this launch's epilogue + a small finalize launch.  Returns (out or None, out_planes or None)."""
    B = x.shape[0]
    if x_planes:
        if x.dim() != 2 or x.shape[1] != plane_ld(I) or x.stride(1) != 1:
            raise RuntimeError("bnn_amd: x planes must be (B,%d), got %s" % (plane_ld(I), tuple(x.shape)))
    elif x.dim() != 2 or x.shape[1] != I:
        raise RuntimeError("bnn_amd: input must be (B,%d), got %s" % (I, tuple(x.shape)))
    if out is None and want_out:
        out = torch.empty((B, O), dtype=torch.float32, device=x.device)
    if eps is not None and tuple(eps.shape) != (B, O):
        raise RuntimeError("bnn_amd: eps must be (%d,%d), got %s" % (B, O, tuple(eps.shape)))
    if B == 0 and finalize is None:
        return out, out_planes
    if x.stride(1) != 1 or (x.stride(0) < x.shape[1]):
        x = x.contiguous()
    d = _lib.GemmDesc()
    d.x, d.ldx = _ptr_rows(x, "input"), x.stride(0)
    d.e_w, d.var_w, d.ld = _ptr(e_w), _ptr(var_w), operand_ld(I)
    d.mean_scale, d.wvar_scale = _ptr(e_scale, "e_scale"), _ptr(v_scale, "v_scale")
    d.bias_mean, d.bias_var, d.var_scale, d.eps = _ptr(bias_mean), _ptr(bias_var), _ptr(var_scale), _ptr(eps, "eps")
    d.rng = rng.data_ptr() if rng is not None else None
    d.rng_stream, d.row_offset = rng_stream, row_offset
    d.out, d.ldo = (out.data_ptr(), out.stride(0)) if out is not None else (None, 0)
    d.out_planes, d.ldp = (out_planes.data_ptr(), out_planes.stride(0)) if out_planes is not None else (None, 0)
    d.std_out = _ptr(std_out, "std_out")
    d.B, d.I, d.O = B, I, O
    d.flags = F_F16S | (F_RELU if relu else 0) | (F_VAR1 if var1 else 0) | (F_XPLANES if x_planes else 0)
    if head is not None:
        C = head["out"].shape[1]
        d.head_e, d.head_v, d.head_ld, d.head_classes = _ptr(head["e_w"]), _ptr(head["var_w"]), head["e_w"].stride(0), C
        d.head_bias_mean, d.head_bias_var, d.head_eps = _ptr(head.get("bias_mean")), _ptr(head.get("bias_var")), _ptr(head.get("eps"), "eps")
        d.head_rng_stream = head.get("rng_stream", 0)
        d.head_out, d.head_ldo, d.head_slab = head["out"].data_ptr(), head["out"].stride(0), _ptr(head["slab"])
        d.head_flags = F_LOG_SOFTMAX if head.get("log_softmax") else 0
    if finalize is not None:
        live, adv = (finalize[4], finalize[5]) if len(finalize) > 4 else (None, 0)
        d.layers = finalize[0] if finalize[0] is not None else None
        d.n_layers, d.fin_rng, d.kl_total, d.rng_live, d.advance = finalize[1], finalize[2], finalize[3], live, adv
    ev = GEMM_EVENTS.take() if GEMM_EVENTS is not None else None
    if ev is not None:
        ev[0].record()
    rc = _lib.lib().lbbnn_lrt_gemm_ex(ctypes.byref(d), _stream())
    _lib.check(rc, "lbbnn_lrt_gemm_ex")
    if ev is not None:
        ev[1].record()
        GEMM_EVENTS.append((B, I, O, ev[0], ev[1]))
    return out, out_planes


def lrt_gemm_combine(a, op, *, K: int, N: int, comb_x, comb_add, split: bool = False):
    """lbbnn_lrt_gemm_combine: comb_add + 2 * comb_x (.) (a (M,K) @ op^T), op = [N][operand_ld(K)]; writes into comb_add's
    storage (returned)."""
    M = a.shape[0]
    if a.stride(1) != 1:
        a = a.contiguous()
    rc = _lib.lib().lbbnn_lrt_gemm_combine(
        _ptr_rows(a, "a"), a.stride(0), _ptr(op), operand_ld(K), _ptr_rows(comb_x, "comb_x"), comb_x.stride(0),
        _ptr_rows(comb_add, "comb_add"), comb_add.stride(0), comb_add.data_ptr(), comb_add.stride(0), M, K, N,
        F_SPLIT16 if split else 0, _stream())
    _lib.check(rc, "lbbnn_lrt_gemm_combine")
    return comb_add


# ----------------------------------------------------------------------------------------- K3
def _ptr_array(ts: Sequence[torch.Tensor]):
    arr = (ctypes.c_void_p * max(len(ts), 1))()
    for i, t in enumerate(ts):
        arr[i] = _ptr(t, "flow parameter")
    return arr


def mnf_flow_planar(q0_mean, q0_log_var, z_params, r_params, *, eps_fwd=None, eps_kl=None,
                    rng: Optional[torch.Tensor] = None, layer_id: int = 0, z_fwd, z_kl=None, scal=None,
                    want_kl: bool = True):
    """lbbnn_mnf_flow_planar.  z_params / r_params: lists of (u, w, bias) per transform."""
    I = q0_mean.shape[0]
    zu, zw, zb = (_ptr_array([p[k] for p in z_params]) for k in range(3))
    ru, rw, rb = (_ptr_array([p[k] for p in r_params]) for k in range(3))
    rc = _lib.lib().lbbnn_mnf_flow_planar(
        _ptr(q0_mean, "q0_mean"), _ptr(q0_log_var, "q0_log_var"),
        zu, zw, zb, len(z_params), ru, rw, rb, len(r_params),
        _ptr(eps_fwd), _ptr(eps_kl), rng.data_ptr() if rng is not None else None, layer_id,
        _ptr(z_fwd), _ptr(z_kl), _ptr(scal), I, 1 if want_kl else 0, _stream())
    _lib.check(rc, "lbbnn_mnf_flow_planar")


FLOW_PLANAR, FLOW_RADIAL, FLOW_HOUSEHOLDER, FLOW_SYLVESTER = 0, 1, 2, 3


def flow_chain(steps, *, I: int, z_in=None, q0_mean=None, q0_log_var=None, eps=None, rng=None, rng_stream: int = 0,
               z_out: Optional[torch.Tensor] = None, logdet=None, log_q0=None, z_last=None):
    """lbbnn_flow_chain.  steps: list of (type, M, p0, p1, p2) (tensors or None).  Returns (z_out, logdet)."""
    if len(steps) > _lib.MAX_FLOW_T:
        raise ValueError("bnn_amd: at most %d transforms per flow" % _lib.MAX_FLOW_T)
    ch = _lib.FlowChain()
    ch.n = len(steps)
    for k, (ty, M, p0, p1, p2) in enumerate(steps):
        st = ch.step[k]
        st.type, st.M = ty, M
        st.p0, st.p1, st.p2 = _ptr(p0, "flow parameter"), _ptr(p1, "flow parameter"), _ptr(p2, "flow parameter")
    ref = z_in if z_in is not None else q0_mean
    if z_out is None:
        z_out = torch.empty(I, dtype=torch.float32, device=ref.device)
    if logdet is None:
        logdet = torch.empty(1, dtype=torch.float32, device=ref.device)
    rc = _lib.lib().lbbnn_flow_chain(ctypes.byref(ch), _ptr(z_in, "z"), _ptr(q0_mean), _ptr(q0_log_var), _ptr(eps),
                                     rng.data_ptr() if rng is not None else None, rng_stream, I,
                                     z_out.data_ptr(), logdet.data_ptr(),
                                     log_q0.data_ptr() if log_q0 is not None else None,
                                     z_last.data_ptr() if z_last is not None else None, _stream())
    _lib.check(rc, "lbbnn_flow_chain")
    return z_out, logdet


def flow_chain_rows(steps, z: torch.Tensor):
    """lbbnn_flow_chain_rows: the 1-D chain applied to every row of z (R,I).  Returns (z_out (R,I), logdet (R,))."""
    if len(steps) > _lib.MAX_FLOW_T:
        raise ValueError("bnn_amd: at most %d transforms per flow" % _lib.MAX_FLOW_T)
    ch = _lib.FlowChain()
    ch.n = len(steps)
    for k, (ty, M, p0, p1, p2) in enumerate(steps):
        st = ch.step[k]
        st.type, st.M = ty, M
        st.p0, st.p1, st.p2 = _ptr(p0, "flow parameter"), _ptr(p1, "flow parameter"), _ptr(p2, "flow parameter")
    z = z.float()
    if z.stride(1) != 1:
        z = z.contiguous()
    R, I = z.shape
    z_out = torch.empty((R, I), dtype=torch.float32, device=z.device)
    logdet = torch.zeros(R, dtype=torch.float32, device=z.device)
    rc = _lib.lib().lbbnn_flow_chain_rows(ctypes.byref(ch), _ptr_rows(z, "z"), z.stride(0), R, I, z_out.data_ptr(), I,
                                          logdet.data_ptr(), _stream())
    _lib.check(rc, "lbbnn_flow_chain_rows")
    return z_out, logdet


def mnf_aux_backward(act_mu, act_var, eps_act, r0_b1, r0_b2, zb_last, g_kl, rng=None, layer_id: int = 0):
    """lbbnn_mnf_aux_backward -> (da_mu, da_var, aux); zb_last: 1-element view of the forward's scal[3]."""
    O, I = act_mu.shape[0], r0_b1.shape[0]
    da_mu, da_var = torch.empty_like(act_mu), torch.empty_like(act_mu)
    aux = torch.empty(4, dtype=torch.float32, device=act_mu.device)
    rc = _lib.lib().lbbnn_mnf_aux_backward(_ptr(act_mu, "act_mu"), _ptr(act_var, "act_var"), _ptr(eps_act, "eps_act"),
                                           _ptr(r0_b1, "r0_b1"), _ptr(r0_b2, "r0_b2"), zb_last.data_ptr(), _ptr(g_kl, "g_kl"),
                                           O, I, da_mu.data_ptr(), da_var.data_ptr(), aux.data_ptr(),
                                           rng.data_ptr() if rng is not None else None, layer_id, _stream())
    _lib.check(rc, "lbbnn_mnf_aux_backward")
    return da_mu, da_var, aux


def bias_backward(bias_mu, bias_rho, g_sum, gv_sum, g_kl, priors: Priors):
    """lbbnn_bias_backward -> (d_bias_mu, d_bias_rho) of a layer without flows (gv_sum / g_kl may be None)."""
    d_mu, d_rho = torch.empty_like(bias_mu), torch.empty_like(bias_rho)
    rc = _lib.lib().lbbnn_bias_backward(_ptr(bias_mu, "bias_mu"), _ptr(bias_rho, "bias_rho"), _ptr(g_sum, "g_sum"),
                                        _ptr(gv_sum), _ptr(g_kl), ctypes.byref(priors), d_mu.data_ptr(), d_rho.data_ptr(),
                                        bias_mu.shape[0], _stream())
    _lib.check(rc, "lbbnn_bias_backward")
    return d_mu, d_rho


def bias_backward_partials(job, B, bias_mu, bias_rho, g_kl, priors: Priors):
    """lbbnn_bias_backward_partials: `job` is what output_grad(defer_sums=[...]) filed (its partials wait in job["work"])."""
    d_mu, d_rho = torch.empty_like(bias_mu), torch.empty_like(bias_rho)
    rc = _lib.lib().lbbnn_bias_backward_partials(job["work"].data_ptr(), B, bias_mu.shape[0], 1 if job["nq"] == 2 else 0,
                                                 _ptr(bias_mu, "bias_mu"), _ptr(bias_rho, "bias_rho"), _ptr(g_kl),
                                                 ctypes.byref(priors), d_mu.data_ptr(), d_rho.data_ptr(), _stream())
    _lib.check(rc, "lbbnn_bias_backward_partials")
    return d_mu, d_rho


def mnf_aux_backward_batch(items, g_kl):
    """lbbnn_mnf_aux_backward_batch: V1 of several layers in one launch.  items: dicts with act_mu, act_var, eps_act (or
    None), r0_b1, r0_b2, zb_last, rng (or None), layer_id.  Returns [(da_mu, da_var, aux)] in the same order."""
    outs = []
    i = 0
    while i < len(items):
        grp = items[i:i + 4]
        arr = (_lib.AuxBwdArgs * len(grp))()
        for k, it in enumerate(grp):
            da_mu, da_var = torch.empty_like(it["act_mu"]), torch.empty_like(it["act_mu"])
            aux = torch.empty(4, dtype=torch.float32, device=da_mu.device)
            a = arr[k]
            a.act_mu, a.act_var, a.eps_act = _ptr(it["act_mu"], "act_mu"), _ptr(it["act_var"], "act_var"), _ptr(it.get("eps_act"))
            a.r0_b1, a.r0_b2, a.zb_last, a.g_kl = _ptr(it["r0_b1"]), _ptr(it["r0_b2"]), it["zb_last"].data_ptr(), _ptr(g_kl, "g_kl")
            a.da_mu, a.da_var, a.aux = da_mu.data_ptr(), da_var.data_ptr(), aux.data_ptr()
            a.rng = it["rng"].data_ptr() if it.get("rng") is not None else None
            a.O, a.I, a.layer_id = da_mu.shape[0], it["r0_b1"].shape[0], it["layer_id"]
            outs.append((da_mu, da_var, aux))
        _lib.check(_lib.lib().lbbnn_mnf_aux_backward_batch(arr, len(grp), _stream()), "lbbnn_mnf_aux_backward_batch")
        i += 4
    return outs


def mnf_flow_planar_backward(q0_mean, q0_log_var, z_params, r_params, *, eps_fwd=None, eps_kl=None, r0_b1=None, r0_b2=None,
                             aux=None, dz_fwd=None, dz_kl=None, g_kl=None, bias_mu, bias_rho, g_sum, gv_sum=None,
                             priors: Priors, rng=None, layer_id: int = 0, defer=None):
    """lbbnn_mnf_flow_planar_backward.  Returns a dict: q0_mean, q0_log_var, r0_b1, r0_b2, bias_mu, bias_rho and
    z_flow / r_flow = lists of (du, dw, dbias) per transform.  ``defer``: a list -- the launch is NOT issued; the filled
    argument struct (with everything it points to kept alive) is appended for mnf_flow_planar_backward_flush."""
    I, O = q0_mean.shape[0], bias_mu.shape[0]
    a = _lib.FlowBwdArgs()
    for name, t in (("q0_mean", q0_mean), ("q0_log_var", q0_log_var), ("eps_fwd", eps_fwd), ("eps_kl", eps_kl),
                    ("r0_b1", r0_b1), ("r0_b2", r0_b2), ("aux", aux), ("dz_fwd", dz_fwd), ("dz_kl", dz_kl),
                    ("g_kl", g_kl), ("bias_mu", bias_mu), ("bias_rho", bias_rho), ("g_sum", g_sum), ("gv_sum", gv_sum)):
        setattr(a, name, _ptr(t, name))
    a.priors = priors
    f = dict(dtype=torch.float32, device=q0_mean.device)
    out = {n: torch.empty(I, **f) for n in ("q0_mean", "q0_log_var", "r0_b1", "r0_b2")}
    out["bias_mu"], out["bias_rho"] = torch.empty(O, **f), torch.empty(O, **f)
    for n in ("q0_mean", "q0_log_var", "r0_b1", "r0_b2", "bias_mu", "bias_rho"):
        setattr(a, "d_" + n, out[n].data_ptr())
    for key, params, fl, gr in (("z_flow", z_params, a.z_flow, a.d_z_flow), ("r_flow", r_params, a.r_flow, a.d_r_flow)):
        fl.T = len(params)
        grads = []
        for t, (u, w, b) in enumerate(params):
            fl.u[t], fl.w[t], fl.b[t] = _ptr(u, "flow u"), _ptr(w, "flow w"), _ptr(b, "flow bias")
            g3 = (torch.empty_like(u), torch.empty_like(w), torch.empty_like(b))
            gr.u[t], gr.w[t], gr.b[t] = g3[0].data_ptr(), g3[1].data_ptr(), g3[2].data_ptr()
            grads.append(g3)
        out[key] = grads
    work = torch.empty(_lib.lib().lbbnn_mnf_flow_backward_workspace(I, len(z_params), len(r_params)), **f)
    a.work, a.O, a.I = work.data_ptr(), O, I
    a.rng, a.layer_id = (rng.data_ptr() if rng is not None else None), layer_id
    if defer is not None:
        keep = (q0_mean, q0_log_var, eps_fwd, eps_kl, r0_b1, r0_b2, aux, dz_fwd, dz_kl, g_kl, bias_mu, bias_rho, g_sum, gv_sum,
                z_params, r_params, work, rng)
        # (NOT the output tensors: autograd must hold the only reference so that AccumulateGrad adopts them instead of
        # copying them -- a copy made before the deferred launch would copy unwritten memory)
        defer.append((a, keep))
        return out
    _lib.check(_lib.lib().lbbnn_mnf_flow_planar_backward(ctypes.byref(a), _stream()), "lbbnn_mnf_flow_planar_backward")
    return out


def mnf_flow_planar_backward_flush(pending):
    """Issue the deferred V2 chains: ONE launch per group of up to 4 layers (lbbnn_mnf_flow_planar_backward_batch), or one
    launch each where a flow has more than 4 transforms."""
    i = 0
    while i < len(pending):
        grp = pending[i:i + 4]
        arr = (_lib.FlowBwdArgs * len(grp))()
        for k, (a, _) in enumerate(grp):
            ctypes.memmove(ctypes.byref(arr[k]), ctypes.byref(a), ctypes.sizeof(_lib.FlowBwdArgs))
        rc = _lib.lib().lbbnn_mnf_flow_planar_backward_batch(arr, len(grp), _stream())
        if rc == -2:                                  # LBBNN_E_SHAPE: a flow with more than 4 transforms
            for a, _ in grp:
                _lib.check(_lib.lib().lbbnn_mnf_flow_planar_backward(ctypes.byref(a), _stream()), "lbbnn_mnf_flow_planar_backward")
        else:
            _lib.check(rc, "lbbnn_mnf_flow_planar_backward_batch")
        i += 4
    pending.clear()


_DENSE_GRAD_FIELDS = {0: ("w_in", "b_in", ("w_mid", 0), ("b_mid", 0), ("w_mid", 1), ("b_mid", 1), ("w_mid", 2), ("b_mid", 2),
                          "w_a", "b_a", "w_b", "b_b"),
                      1: ("w_in", "b_in", "w_a", "b_a", "w_b", "b_b")}


def _dense_grad_array(kind_id, params, T):
    """lbbnn_dense_grad_t array + fresh gradient tensors for T transforms whose parameters come in module order
    (RNVP: network.0/2/4/6 weight, bias, t, s; MNF type: f, g, k -- the order dense_descs reads them in)."""
    fields = _DENSE_GRAD_FIELDS[kind_id]
    n = len(fields)
    arr = (_lib.DenseGrad * max(T, 1))()
    grads = [torch.empty_like(p) for p in params[:T * n]]
    for t in range(T):
        for f, g in zip(fields, grads[t * n:(t + 1) * n]):
            if isinstance(f, tuple):
                getattr(arr[t], f[0])[f[1]] = g.data_ptr()
            else:
                setattr(arr[t], f, g.data_ptr())
    return arr, grads


def mnf_flow_dense_backward(q0_mean, q0_log_var, z_descs, Tz, z_kind, z_params, r_descs, Tr, r_kind, r_params, *, save,
                            eps_fwd=None, eps_kl=None, r0_b1=None, r0_b2=None, aux=None, dz_fwd=None, dz_kl=None, g_kl=None,
                            bias_mu, bias_rho, g_sum, gv_sum=None, priors: Priors, rng=None, layer_id: int = 0, defer=None,
                            keep=()):
    """lbbnn_mnf_flow_dense_backward (``defer``: a list -- nothing is launched, the filled argument struct and what it
    points to are appended for mnf_flow_dense_backward_flush).  z_descs / r_descs: the forward's lbbnn_dense_transform_t arrays (same masks);
    z_params / r_params: the flows' parameter tensors in module order (shapes of the gradients).  Returns a dict:
    q0_mean, q0_log_var, r0_b1, r0_b2, bias_mu, bias_rho and z_flow / r_flow = flat gradient lists in parameter order."""
    I, O = q0_mean.shape[0], bias_mu.shape[0]
    a = _lib.DenseBwdArgs()
    for name, t in (("q0_mean", q0_mean), ("q0_log_var", q0_log_var), ("eps_fwd", eps_fwd), ("eps_kl", eps_kl),
                    ("r0_b1", r0_b1), ("r0_b2", r0_b2), ("aux", aux), ("dz_fwd", dz_fwd), ("dz_kl", dz_kl),
                    ("g_kl", g_kl), ("bias_mu", bias_mu), ("bias_rho", bias_rho), ("g_sum", g_sum), ("gv_sum", gv_sum),
                    ("save", save)):
        setattr(a, name, _ptr(t, name))
    a.priors = priors
    f = dict(dtype=torch.float32, device=q0_mean.device)
    out = {n: torch.empty(I, **f) for n in ("q0_mean", "q0_log_var", "r0_b1", "r0_b2")}
    out["bias_mu"], out["bias_rho"] = torch.empty(O, **f), torch.empty(O, **f)
    for n in ("q0_mean", "q0_log_var", "r0_b1", "r0_b2", "bias_mu", "bias_rho"):
        setattr(a, "d_" + n, out[n].data_ptr())
    gz, out["z_flow"] = _dense_grad_array(z_kind, z_params, Tz)
    gr, out["r_flow"] = _dense_grad_array(r_kind, r_params, Tr)
    a.zt = ctypes.cast(z_descs, ctypes.POINTER(_lib.DenseTransform))
    a.rt = ctypes.cast(r_descs, ctypes.POINTER(_lib.DenseTransform)) if r_descs is not None else None
    a.d_zt = ctypes.cast(gz, ctypes.POINTER(_lib.DenseGrad))
    a.d_rt = ctypes.cast(gr, ctypes.POINTER(_lib.DenseGrad))
    work = torch.empty(_lib.lib().lbbnn_mnf_flow_dense_backward_workspace(I), **f)
    a.work, a.Tz, a.Tr, a.O, a.I = work.data_ptr(), Tz, Tr, O, I
    a.rng, a.layer_id = (rng.data_ptr() if rng is not None else None), layer_id
    if defer is not None:
        # (the output tensors are NOT kept here: autograd must hold their only reference, see mnf_flow_planar_backward)
        alive = (q0_mean, q0_log_var, eps_fwd, eps_kl, r0_b1, r0_b2, aux, dz_fwd, dz_kl, g_kl, bias_mu, bias_rho, g_sum, gv_sum,
                 save, z_descs, r_descs, z_params, r_params, gz, gr, work, rng, keep)
        defer.append((a, alive, (Tz, Tr, g_kl is not None)))
        return out
    _lib.check(_lib.lib().lbbnn_mnf_flow_dense_backward(ctypes.byref(a), _stream()), "lbbnn_mnf_flow_dense_backward")
    return out


def mnf_flow_dense_backward_flush(pending):
    """Issue the deferred dense-flow chains: layers that agree on (Tz, Tr, KL branch) share their launches
    (lbbnn_mnf_flow_dense_backward_batch, up to 4 layers per call)."""
    groups = {}
    for item in pending:
        groups.setdefault(item[2], []).append(item)
    for items in groups.values():
        for i in range(0, len(items), 4):
            grp = items[i:i + 4]
            arr = (_lib.DenseBwdArgs * len(grp))()
            for k, (a, _, _) in enumerate(grp):
                ctypes.memmove(ctypes.byref(arr[k]), ctypes.byref(a), ctypes.sizeof(_lib.DenseBwdArgs))
            _lib.check(_lib.lib().lbbnn_mnf_flow_dense_backward_batch(arr, len(grp), _stream()), "lbbnn_mnf_flow_dense_backward_batch")
    pending.clear()


def q0_rows(q0_mean, q0_log_var, R: int, *, eps=None, rng=None, rng_stream: int = 0):
    """lbbnn_q0_rows: z0 (R,I) = q0_mean + exp(q0_log_var)^.5 * eps, eps (R,I) explicit or Philox."""
    I = q0_mean.shape[0]
    if eps is not None:
        eps = eps.reshape(R, I).float().contiguous()
    elif rng is None:
        raise RuntimeError("bnn_amd: q0_rows needs explicit eps or an rng state")
    z0 = torch.empty((R, I), dtype=torch.float32, device=q0_mean.device)
    _lib.check(_lib.lib().lbbnn_q0_rows(_ptr(q0_mean, "q0_mean"), _ptr(q0_log_var, "q0_log_var"), _ptr(eps, "eps"),
                                        rng.data_ptr() if rng is not None else None, rng_stream, R, I, z0.data_ptr(),
                                        _stream()), "lbbnn_q0_rows")
    return z0


def flow_dense_rows(descs, T: int, z: torch.Tensor, *, masks: Optional[torch.Tensor] = None, rng: Optional[torch.Tensor] = None,
                    rng_stream: int = STREAM_ROW_MASK * 64, row_base: int = 0, want_masks: bool = False, keep=()):
    """lbbnn_flow_dense_rows: a chain of T dense coupling transforms (``descs`` = PropagateFlow.dense_descs(None, None)[0])
    on the R rows of z (R,I), each row with its own masks.  masks: (T,R,I) in {0,1} or None (drawn in-kernel from ``rng``).
    Returns (z_out (R,I), logdet_rows (R,), masks used (T,R,I) or None)."""
    if z.dim() != 2:
        raise RuntimeError("bnn_amd: flow_dense_rows wants z as (R,I), got %s" % (tuple(z.shape),))
    R, I = z.shape
    if I > _lib.lib().lbbnn_flow_dense_rows_max_dim():
        raise RuntimeError("bnn_amd: row-batched dense flows hold 16 rows of z in LDS: dim %d exceeds the limit %d"
                           % (I, _lib.lib().lbbnn_flow_dense_rows_max_dim()))
    z = z.float()
    if z.stride(1) != 1:
        z = z.contiguous()
    f = dict(dtype=torch.float32, device=z.device)
    if masks is not None:
        if tuple(masks.shape) != (T, R, I):
            raise RuntimeError("bnn_amd: masks must be (%d,%d,%d), got %s" % (T, R, I, tuple(masks.shape)))
        masks = masks.float().contiguous()
    elif rng is None:
        raise RuntimeError("bnn_amd: flow_dense_rows needs explicit masks or an rng state")
    z_out = torch.empty((R, I), **f)
    logdet = torch.zeros(R, **f)
    mask_out = torch.empty((T, R, I), **f) if (want_masks and masks is None) else None
    _lib.check(_lib.lib().lbbnn_flow_dense_rows(
        descs, T, _ptr(masks, "masks"), _ptr(mask_out), rng.data_ptr() if rng is not None else None, rng_stream,
        int(row_base), _ptr_rows(z, "z"), z.stride(0), R, I, z_out.data_ptr(), I, logdet.data_ptr(), _stream()),
        "lbbnn_flow_dense_rows")
    del keep
    return z_out, logdet, (masks if masks is not None else mask_out)


def flow_dense_save_size(I: int, Tz: int, Tr: int) -> int:
    return int(_lib.lib().lbbnn_flow_dense_save_size(I, Tz, Tr))


# ----------------------------------------------------------------------------------------- K4
def mnf_flow_dense(q0_mean, q0_log_var, z_descs, Tz, r_descs, Tr, *, eps_fwd=None, eps_kl=None,
                   rng: Optional[torch.Tensor] = None, layer_id: int = 0, z_fwd, z_kl=None, scal=None, work=None,
                   want_kl: bool = True):
    """lbbnn_mnf_flow_dense.  z_descs / r_descs: ctypes arrays of _lib.DenseTransform (see flows.dense_descs)."""
    I = q0_mean.shape[0]
    rc = _lib.lib().lbbnn_mnf_flow_dense(
        _ptr(q0_mean, "q0_mean"), _ptr(q0_log_var, "q0_log_var"),
        z_descs, Tz, r_descs, Tr,
        _ptr(eps_fwd), _ptr(eps_kl), rng.data_ptr() if rng is not None else None, layer_id,
        _ptr(z_fwd), _ptr(z_kl), _ptr(scal), _ptr(work), I, 1 if want_kl else 0, _stream())
    _lib.check(rc, "lbbnn_mnf_flow_dense")


def flow_dense_workspace(I: int) -> int:
    return int(_lib.lib().lbbnn_flow_dense_workspace(I))


# ----------------------------------------------------------------------------------------- K5
def kl_finalize(kl_rows, bias_mu, bias_rho, *, priors: Priors, act_mu=None, act_var=None, eps_act=None,
                r0_b1=None, r0_b2=None, scal=None, rng: Optional[torch.Tensor] = None, layer_id: int = 0,
                kl_out=None, kl_layer=None, accumulate: bool = False):
    O = bias_mu.shape[0]
    I = r0_b1.shape[0] if r0_b1 is not None else 0
    rc = _lib.lib().lbbnn_kl_finalize(
        _ptr(kl_rows), _ptr(bias_mu, "bias_mu"), _ptr(bias_rho, "bias_rho"), O,
        _ptr(act_mu), _ptr(act_var), _ptr(eps_act), _ptr(r0_b1), _ptr(r0_b2), I,
        _ptr(scal), ctypes.byref(priors), rng.data_ptr() if rng is not None else None, layer_id,
        _ptr(kl_out), _ptr(kl_layer), 1 if accumulate else 0, _stream())
    _lib.check(rc, "lbbnn_kl_finalize")


def transpose_operand(src: torch.Tensor, *, square: bool = False, split: bool = False,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """lbbnn_transpose_operand: (R,C) fp32 -> GEMM operand [C][operand_ld(R)] holding src^T (squared if asked)."""
    R, C = src.shape
    ld = operand_ld(R)
    if out is None:
        out = torch.empty((C, ld), dtype=torch.float32, device=src.device)
    rc = _lib.lib().lbbnn_transpose_operand(_ptr_rows(src, "src"), R, C, src.stride(0), out.data_ptr(), ld,
                                            1 if square else 0, F_SPLIT16 if split else 0, _stream())
    _lib.check(rc, "lbbnn_transpose_operand")
    return out


def format_operand(src: torch.Tensor, *, square: bool = False, split: bool = False) -> torch.Tensor:
    """lbbnn_format_operand: (R,C) fp32 -> GEMM operand [R][operand_ld(C)] holding src (squared if asked), no transpose."""
    R, C = src.shape
    ld = operand_ld(C)
    out = torch.empty((R, ld), dtype=torch.float32, device=src.device)
    rc = _lib.lib().lbbnn_format_operand(_ptr_rows(src, "src"), R, C, src.stride(0), out.data_ptr(), ld,
                                         1 if square else 0, F_SPLIT16 if split else 0, _stream())
    _lib.check(rc, "lbbnn_format_operand")
    return out


def log_softmax_backward(g: torch.Tensor, logp: torch.Tensor) -> torch.Tensor:
    """lbbnn_log_softmax_backward: g - exp(logp) * rowsum(g) for (B, C <= 64) tensors."""
    B, C = g.shape
    if g.stride(1) != 1:
        g = g.contiguous()
    out = torch.empty((B, C), dtype=torch.float32, device=g.device)
    rc = _lib.lib().lbbnn_log_softmax_backward(_ptr_rows(g, "g"), g.stride(0), _ptr_rows(logp, "logp"), logp.stride(0),
                                               out.data_ptr(), C, B, C, _stream())
    _lib.check(rc, "lbbnn_log_softmax_backward")
    return out


def log_softmax_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, O = x.shape
    if out is None:
        out = torch.empty_like(x)
    rc = _lib.lib().lbbnn_log_softmax_rows(_ptr(x, "input"), x.stride(0), out.data_ptr(), out.stride(0), B, O, _stream())
    _lib.check(rc, "lbbnn_log_softmax_rows")
    return out


# ----------------------------------------------------------------------------------------- workspaces
class LayerWorkspace:
    """Caller-owned device buffers of one layer, allocated once and reused every forward
    (the C ABI never allocates; fixed addresses also make the launch sequence graph-capturable)."""

    def __init__(self, O: int, I: int, device: torch.device, mnf: bool):
        ld = operand_ld(I)
        f = dict(dtype=torch.float32, device=device)
        self.device = device
        self.e_w = torch.empty((O, ld), **f)
        self.var_w = torch.empty((O, ld), **f)
        self.kl_rows = torch.empty(O, **f)
        self.bias_var = torch.empty(O, **f)
        self.e_scale = torch.ones(O, **f)          # LBBNN_F_F16S row scales (written by the weight pass)
        self.v_scale = torch.ones(O, **f)
        self.kl = torch.zeros((), **f)
        self._bw = None
        if mnf:
            self.act_mu = torch.empty(O, **f)
            self.act_var = torch.empty(O, **f)
            self.z_fwd = torch.empty(I, **f)
            self.z_kl = torch.empty(I, **f)
            self.scal = torch.empty(8, **f)

    def backward_operands(self):
        """fp32 operand pair the backward rebuilds for dX (allocated on first use)."""
        if self._bw is None:
            self._bw = (torch.empty_like(self.e_w), torch.empty_like(self.var_w))
        return self._bw
