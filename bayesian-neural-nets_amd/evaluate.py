"""Ensemble evaluation of a Bayesian network -- what ``test_ensemble`` computes per test batch
(LBBNN-GP-MF-LRT.py:229-272, LBBNN-GP-MF-MNF.py:277-334), minus the script's file output:

* ``outputs[s] = net(data, sample=True)`` for s < samples (each a fused no-grad HIP forward with its own draws),
* ensemble prediction = argmax of the mean log-probability over the samples (``outputs[0:10].mean(0)``),
* posterior-mean prediction = argmax of ``net(data, sample=False)``,
* density[s] = mean of one Bernoulli draw of every layer's inclusion probabilities (``layer.gamma.rsample()``).

The reference's own ``test_ensemble`` also runs unchanged on these modules; this is the batched convenience form.
"""
from typing import Dict, Optional

import torch

from . import ops


def _batched_ok(net, data) -> bool:
    """The one-launch-per-kernel ensemble applies to LRT networks and to MNF networks whose flows are planar with <= 4
    transforms, on a HIP device, without injected noise; anything else takes the loop of single forwards."""
    from . import layers as L
    if not isinstance(net, L._NetworkBase) or not data.is_cuda:
        return False
    for l in net._layers():
        if l.noise or getattr(l, "as_written", False):
            return False
        if l._mnf and (l._check_flows() != "planar" or len(l.z_flow.transforms) > 4):
            return False
        if l.in_features % 4 or ops.operand_ld(l.in_features) > 2048:
            return False
    return True


@torch.no_grad()
def ensemble_forward_batched(net, data: torch.Tensor, samples: int = 10) -> torch.Tensor:
    """``samples`` stochastic evaluation forwards of one batch (LBBNN-GP-MF-MNF.py:286-294: TEST_SAMPLES x net(data,
    sample=True)) in 2 + 3 launches instead of 5 per member: one K3 and one K1 launch produce every member's z and
    operands (the variance operand, z-free, once for all), then each layer's GEMM runs all members as gridDim.z slices of
    ONE launch (lbbnn_lrt_gemm_members) -- member m draws at Philox offset (live offset + m), exactly where the m-th of
    ``samples`` consecutive ``net(data, sample=True)`` calls would, so the result is bit-identical to that loop
    (tests/test_parity_gpu.py::test_ensemble_batched_equals_loop_bitwise) under the fp32 and bf16x3 settings.  The member
    dimension exists in the bf16 hi | lo operand format only: under "fp16x3" / "fp16x3f" the batched form still multiplies in
    bf16x3 (2.7e-6 of max|out| against fp64, tighter than fp16x3f's 1.4e-5) while the loop's single forwards take the
    row-scaled fp16 kernels -- same draws, results equal to the formats' error (tools/ensemble_fuzz.py)."""
    import ctypes
    from . import _lib
    net.eval()
    S = int(samples)
    layers = net._layers()
    n = len(layers)
    x = data.view(-1, net.dims[0])
    x = x.float() if x.dtype != torch.float32 else x
    if x.stride(1) != 1:
        x = x.contiguous()
    B, dev = x.shape[0], x.device
    st = ops.RngState.get(dev)
    rng = st.t
    f = dict(dtype=torch.float32, device=dev)
    descs = (_lib.LayerDesc * n)()
    keep, e_all, z_all = [], [], []
    for i, l in enumerate(layers):
        cfg = (True, False, i < n - 1)
        # (the member dimension of the batched ensemble exists in the bf16 hi | lo format: any 16-bit precision selects it)
        l._split_now = int(bool(l._split(x if i == 0 else None)) and (i == 0 or layers[i - 1].out_features % 4 == 0))
        keep.append(l._fill_desc(descs[i], cfg, None))
        ld = ops.operand_ld(l.in_features)
        e = torch.empty((S, l.out_features, ld), **f)
        e_all.append(e)
        descs[i].e_w = e.data_ptr()
        if l._mnf:
            z = torch.empty((S, ld), **f)
            z_all.append(z)
            descs[i].z_fwd = z.data_ptr()
        descs[i].eps_z = None
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().lbbnn_ensemble_operands(descs, n, S, rng.data_ptr(), 1, stream), "lbbnn_ensemble_operands")
    h, h_ms = x, 0                                             # the first layer reads the same rows for every member
    for i, l in enumerate(layers):
        O, I = l.out_features, l.in_features
        ws = l._workspace()
        o_ms = -(-(B * O) // 4) * 4                           # member stride padded to 16 B (vector loads / stores per member)
        out = torch.empty((S, o_ms), **f)[:, :B * O].view(S, B, O) if o_ms != B * O else torch.empty((S, B, O), **f)
        last = i == n - 1
        flags = (0 if last else ops.F_RELU) | (ops.F_SPLIT16 if l._split_now else 0) | \
                (ops.F_LOG_SOFTMAX if (last and O <= 16) else 0)
        rc = _lib.lib().lbbnn_lrt_gemm_members(
            h.data_ptr(), h.stride(-2), h_ms, e_all[i].data_ptr(), O * ops.operand_ld(I), ws.var_w.data_ptr(),
            ops.operand_ld(I), l.bias_mu.data_ptr(), ws.bias_var.data_ptr(), rng.data_ptr(),
            ops.STREAM_EPS_OUT * 64 + l._layer_id, l.row_offset, 1, out.data_ptr(), O, o_ms, B, I, O, flags, S, stream)
        _lib.check(rc, "lbbnn_lrt_gemm_members")
        h, h_ms = out, o_ms
    st.advance(S)                                              # as S single forwards would have
    if layers[-1].out_features > 16:
        h = torch.log_softmax(h, dim=-1)
    for l in layers:
        l.kl = 0
    net._kl_total = None
    del keep
    return h


@torch.no_grad()
def ensemble_forward(net, data: torch.Tensor, samples: int = 10, batched=None) -> torch.Tensor:
    """(samples, B, classes) log-probabilities of ``samples`` stochastic forwards (net left in eval mode).
    ``batched``: None = the one-launch-per-kernel form when the network qualifies (``_batched_ok``), else the loop of
    fused single forwards; True / False force one of them."""
    net.eval()
    if batched is None:
        batched = _batched_ok(net, data)
    if batched:
        return ensemble_forward_batched(net, data, samples)
    outs = [net(data, sample=True) for _ in range(samples)]
    return torch.stack(outs)


@torch.no_grad()
def ensemble_eval(net, data: torch.Tensor, target: Optional[torch.Tensor] = None, samples: int = 10) -> Dict[str, object]:
    outputs = ensemble_forward(net, data, samples)
    density = []
    for _ in range(samples):
        g = [l.gamma.rsample().flatten() for l in (net.l1, net.l2, net.l3)]
        density.append(torch.cat(g).mean())
    pred_ens = outputs.mean(0).argmax(1)
    pred_mean = net(data, sample=False).argmax(1)
    res = {"outputs": outputs, "pred_ensemble": pred_ens, "pred_posterior_mean": pred_mean,
           "density": torch.stack(density)}
    if target is not None:
        res["correct_ensemble"] = int(pred_ens.eq(target).sum())
        res["correct_posterior_mean"] = int(pred_mean.eq(target).sum())
    return res
