"""Gaussian variational dropout, names as in variational_dropout.py: ``BayesianLayer(n, m)`` :55-68,
``BNN`` :72-86, ``loss_fn`` :89-106.  theta is (n, m) = (in, out), NN layout, as in the reference; the HIP
path transposes it into the GEMM operand format on the fly (lbbnn_vd_operands) and runs the same dual-moment
GEMM with ``var_scale = alpha`` and no bias."""
import itertools

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops

_ids = itertools.count(48)


class _VDFn(torch.autograd.Function):
    """Forward: lbbnn_vd_operands + the dual-moment GEMM.  Backward, all on the HIP kernels (round 2; round 1 recomputed the
    layer with torch ops):  with G = dL/dout, G_v = G zeta alpha / (2 sqrt(delta))  (lbbnn_output_grad, gv_scale = alpha)
        dX     = G . theta^T + 2 x (.) (G_v . (theta^2)^T)          operands theta, theta^2 as they lie (lbbnn_format_operand)
        dtheta = x^T . G + 2 theta (.) ((x^2)^T . G_v)              x^T | (x^2)^T from one pass over x (lbbnn_vd_operands),
                                                                    G^T, G_v^T operands (lbbnn_transpose_operand)
    both "+ 2 a (.) (second product)" combinations run in the second GEMM's epilogue (lbbnn_lrt_gemm_combine)."""

    @staticmethod
    def forward(ctx, layer, x, theta, alpha):
        std = torch.empty((x.shape[0], layer.m), dtype=torch.float32, device=x.device)
        out, zeta_src = layer._forward_hip(x, save_rng=True, std_out=std)
        ctx.layer, ctx.zeta_src = layer, zeta_src
        from . import graphs
        graphs.mark_autograd_node(ctx, layer)          # capture guard: graphs.assert_no_live_graph
        ctx.save_for_backward(x, theta, alpha, std)
        return out

    @staticmethod
    def backward(ctx, g):
        x, theta, alpha, std = ctx.saved_tensors
        layer = ctx.layer
        n, m, B = layer.n, layer.m, x.shape[0]
        kind, val = ctx.zeta_src
        gm, gv, _, _, _, _ = ops.output_grad(
            g, std=std, eps=val if kind == "explicit" else None, rng=val if kind != "explicit" else None,
            rng_stream=ops.STREAM_EPS_OUT * 64 + layer._layer_id, row_offset=layer.row_offset, relu=False, gv_scale=alpha)
        split = ops.split_precision()
        gx = gt = None
        if ctx.needs_input_grad[1]:
            sp = split and ops.split_eligible(m, n) and gm.stride(0) % 4 == 0
            op_t = ops.format_operand(theta, square=False, split=sp)            # [n][ld(m)]: theta rows, K = m
            op_t2 = ops.format_operand(theta, square=True, split=sp)
            gx = ops.lrt_gemm(gm, op_t, None, I=m, O=n, mean_only=True, split=sp)
            if n > 16 and x.stride(1) == 1:
                gx = ops.lrt_gemm_combine(gv, op_t2, K=m, N=n, comb_x=x, comb_add=gx, split=sp)
            else:
                gx = ops.dx_combine(gx, ops.lrt_gemm(gv, op_t2, None, I=m, O=n, mean_only=True, split=sp), x)
        if ctx.needs_input_grad[2]:
            # x^T and (x^2)^T as plain fp32 matrices [n][ld(B)] (the GEMM's A side is fp32; it splits it in registers)
            ld = ops.operand_ld(B)
            xc = x if x.is_contiguous() else x.contiguous()
            xt = torch.empty((n, ld), dtype=torch.float32, device=x.device)
            x2t = torch.empty((n, ld), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().lbbnn_vd_operands(xc.data_ptr(), xt.data_ptr(), x2t.data_ptr(), ld, B, n, 0, ops._stream()),
                       "lbbnn_vd_operands")
            sp = split and ops.split_eligible(B, m)
            gmT_op = ops.transpose_operand(gm, split=sp)                        # [m][ld(B)]: K = B
            gvT_op = ops.transpose_operand(gv, split=sp)
            gt = ops.lrt_gemm(xt[:, :B], gmT_op, None, I=B, O=m, mean_only=True, split=sp)
            if m > 16:
                gt = ops.lrt_gemm_combine(x2t[:, :B], gvT_op, K=B, N=m, comb_x=theta, comb_add=gt, split=sp)
            else:
                gt = ops.dx_combine(gt, ops.lrt_gemm(x2t[:, :B], gvT_op, None, I=B, O=m, mean_only=True, split=sp), theta)
        return None, gx, gt, None


class BayesianLayer(nn.Module):
    def __init__(self, n, m):
        super().__init__()
        low, high = -0.1, 0.1
        self.n, self.m = n, m
        self.theta = nn.Parameter((low - high) * torch.rand(size=(n, m)) + high)        # :58-60
        # :61 -- `nn.Parameter(zeros(m)) + 0.2` is a plain (non-leaf) tensor in the reference: never trained, not in
        # state_dict.  A non-persistent buffer keeps those properties and follows .to(device).
        self.register_buffer("alpha", torch.zeros(m) + 0.2, persistent=False)
        self.noise = None                 # {"zeta": (B, m)} for parity tests
        self.row_offset = 0
        self._layer_id = next(_ids) % 64
        self._ws = None

    def _forward_hip(self, x, relu=False, save_rng=False, std_out=None):
        dev = x.device
        ld = ops.operand_ld(self.n)
        if self._ws is None or self._ws[0].device != dev:
            self._ws = (torch.empty((self.m, ld), dtype=torch.float32, device=dev),
                        torch.empty((self.m, ld), dtype=torch.float32, device=dev))
        e_w, var_w = self._ws
        split = (ops.split_precision() and ops.split_eligible(self.n, self.m)
                 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)
        half = split and ops.get_precision() == "fp16"          # one fp16 product per moment (BASELINE configs[4]: "fp16 MFMA")
        _lib.check(_lib.lib().lbbnn_vd_operands(ops._ptr(self.theta.detach(), "theta"), e_w.data_ptr(), var_w.data_ptr(),
                                                ld, self.n, self.m,
                                                (ops.F_SPLIT16 if split else 0) | (ops.F_HALF16 if half else 0), ops._stream()),
                   "lbbnn_vd_operands")
        zeta = (self.noise or {}).get("zeta")
        rng, st = None, None
        if zeta is None:
            st = ops.RngState.get(dev)
            rng = st.t
        src = ("explicit", zeta) if zeta is not None else ("rng", rng.clone() if save_rng else None)
        out = ops.lrt_gemm(x, e_w, var_w, I=self.n, O=self.m, var_scale=self.alpha, eps=zeta, rng=rng,
                           rng_stream=ops.STREAM_EPS_OUT * 64 + self._layer_id, row_offset=self.row_offset,
                           relu=relu, split=split, std_out=std_out, half=half)
        if st is not None:
            st.advance(1)
        return out, src

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("bnn_amd: forward needs a HIP device tensor (input is on %s); there is no CPU path" % x.device)
        x = x.float()
        if torch.is_grad_enabled() and (x.requires_grad or self.theta.requires_grad):
            return _VDFn.apply(self, x, self.theta, self.alpha)
        return self._forward_hip(x)[0]


class BNN(nn.Module):
    """variational_dropout.py:72-86 (784-1200-1200-1200-10 there; ``dims=`` added)."""

    def __init__(self, dims=(28 * 28, 1200, 1200, 1200, 10)):
        super().__init__()
        self.dims = tuple(dims)
        self.l1 = BayesianLayer(dims[0], dims[1])
        self.l2 = BayesianLayer(dims[1], dims[2])
        self.l3 = BayesianLayer(dims[2], dims[3])
        self.l4 = BayesianLayer(dims[3], dims[4])
        for i, l in enumerate((self.l1, self.l2, self.l3, self.l4)):
            l._layer_id = 48 + i              # per-network Philox stream ids (not the process-wide counter)

    def forward(self, x):
        x = x.view(-1, self.dims[0])
        x = F.relu(self.l1(x))
        x = F.relu(self.l2(x))
        x = F.relu(self.l3(x))
        return F.log_softmax(self.l4(x), dim=1)


NUM_BATCHES = 600.0      # len(train_loader.dataset) / config['batch_size'] = 60000 / 100 (variational_dropout.py:90-95)


def loss_fn(prediction, target, model, *, num_batches=None):
    """variational_dropout.py:89-106, callable with the reference's three positional arguments.  There ``num_batches``
    = N / batch_size is computed from the module-level loaders; here it is a keyword (default: the module constant
    NUM_BATCHES = 600, the reference's MNIST value).  As in the reference, ``if model.train():`` (:91) is a CALL: it
    flips the model to training mode and is always truthy, so N is always the training set's size."""
    if num_batches is None:
        num_batches = NUM_BATCHES
    model.train()                                                    # :91 (side effect kept)
    KL = 0
    c1, c2, c3 = 1.16145124, -1.50204118, 0.58629921
    for layer in model.children():
        if isinstance(layer, BayesianLayer):
            a = layer.alpha
            KL = KL + (0.5 * torch.log(a) + c1 * a + c2 * a ** 2 + c3 * a ** 3).sum()
    return KL / num_batches + F.nll_loss(prediction, target, reduction="sum")
