"""Drop-in ``torch.nn.Module``s for the reference's Bayesian linear layers, forward in HIP.

``LRTBayesianLinear``  <-> BayesianLinear of LBBNN-GP-MF-LRT.py:129-197
``MNFBayesianLinear``  <-> BayesianLinear of LBBNN-GP-MF-MNF.py:133-239
``LRTBayesianNetwork`` / ``MNFBayesianNetwork`` <-> BayesianNetwork (…LRT.py:199-214, …MNF.py:244-260)

Same constructor positional arguments, same ``forward(input, sample=False,
calculate_log_probs=False)``, same ``.kl`` attribute, same parameter names (so ``state_dict``s,
optimizer parameter groups by attribute and the eval code's ``layer.gamma.rsample()`` /
``layer.alpha_q`` accesses carry over).  Seeded construction consumes the torch generator in the
reference's order, so ``torch.manual_seed(s); BayesianLinear(...)`` gives the reference's values.

Additions (keyword-only, defaults = reference behaviour): ``priors=``, ``z_flow_type= /
r_flow_type=`` (module globals in the reference, LBBNN-GP-MF-MNF.py:46-47), and ``layer.noise``
-- a dict of explicit draws (``eps_out, eps_z, eps_z2, eps_act``) for parity tests; when
``None`` the kernels draw N(0,1) in-kernel (Philox) from the device RNG state.

There is no CPU path: parameters may be *constructed* on CPU (as the reference does) but
``forward`` needs them on a HIP device.
"""
import itertools
import weakref
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _grad, ops
from ._lib import Priors
from .distributions import Bernoulli, Gaussian
from .flows import PropagateFlow

import os as _os
# LBBNN_DENSE_TORCH_BWD=1: differentiate the dense (RNVP / MNF-type) flows with torch autograd on the GPU vectors, as
# before lbbnn_mnf_flow_dense_backward existed (kept for A/B timing and as a second opinion in the tests)
_DENSE_HIP_BWD = _os.environ.get("LBBNN_DENSE_TORCH_BWD", "0") != "1"
# LBBNN_TORCH_MASKS=1: draw the Bernoulli masks of the dense flows with torch's generator (one bernoulli_ launch per
# forward) instead of in the flow kernels from the layer's Philox state
_MASKS_IN_KERNEL = _os.environ.get("LBBNN_TORCH_MASKS", "0") != "1"
# LBBNN_DENSE_DEFER=0: keep the r flow and the flows' scalars of the dense flows on the forward's own stream, ahead of the weight
# pass (default: while a HIP graph is being captured they go on a side stream beside the weight pass and the first GEMM -- only
# the KL finalize needs them; "always": in eager launches too)
_DENSE_DEFER = {"0": False, "always": "always"}.get(_os.environ.get("LBBNN_DENSE_DEFER", "1"), True)
# LBBNN_F16_FIRST=planes: the first row-scaled-fp16 layer of a fused forward reads the network input as fp16 hi | lo planes,
# made by extra workgroups of the flow launch (lbbnn_layers_operands_x).  Default (f32): it takes the fp32 input as it is and
# splits it in registers.  Measured on the headline net (tools/precision_time.py, one process, interleaved): the in-register
# split costs the first GEMM's K loop 1.33 us per step instead of 1.08 (~6 us); the format pass costs 6.9 us as a launch of its
# own (0.1543 against 0.1487 ms per forward) and as much as it saves when it rides in the flow launch (0.1401 against
# 0.1392 ms): no form of the pre-pass beats the in-register split, so the simpler one is the default.
# LBBNN_HEAD_FOLD=0: keep the 10-class head a GEMM launch of its own (the skinny kernel) in the fused fp16 forward
_HEAD_FOLD = _os.environ.get("LBBNN_HEAD_FOLD", "1") != "0"
_ADV_RIDE = _os.environ.get("LBBNN_ADV_RIDE", "1") != "0"          # training forward: the RNG advance rides in the first GEMM's launch
_V1_BATCH = _os.environ.get("LBBNN_V1_BATCH", "1") != "0"          # all layers' V1 in one launch from the KL sum's backward (A/B knob)
_DEFER_SUMS = _os.environ.get("LBBNN_DEFER_SUMS", "1") != "0"      # column sums finished with the deferred vector chains (A/B knob)
_LRT_BIAS_HIP = _os.environ.get("LBBNN_LRT_BIAS_HIP", "1") != "0"   # the LRT layer's bias gradients through lbbnn_bias_backward (A/B knob)
_HEAD_DW = _os.environ.get("LBBNN_HEAD_DW", "1") != "0"           # the head's weight gradients through lbbnn_head_dw (A/B knob)
_F16_FIRST_PLANES = _os.environ.get("LBBNN_F16_FIRST", "f32") != "f32"
_SIDE = {}


def _side_stream(dev):
    """One side stream per device (the deferred part of the dense flows); created on first use."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev)
    return st


# Optional deferral of the vector-sized backward chains (V2 / lbbnn_mnf_flow_dense_backward: latency-bound, a handful of
# workgroups, feeding nothing but the optimizer): each layer's backward only FILES its chain, and when the backward pass
# is over all layers' chains are issued together -- one launch for the planar ones
# (lbbnn_mnf_flow_planar_backward_batch), one set of 2 + 3(Tz+Tr) launches for the dense ones
# (lbbnn_mnf_flow_dense_backward_batch) instead of one per layer.  OFF unless a caller that also FLUSHES before the
# optimizer step turns it on (graphs.make_graphed_train_step, or `with layers.vector_backward_overlap(): loss.backward()`):
# the gradients such a backward returns are only complete after join_vector_backward().
_OVERLAP = {"on": False, "planar": [], "dense": [], "adopt": [], "sums": []}


def _drop_deferred():
    for k in ("planar", "dense", "adopt", "sums"):
        _OVERLAP[k].clear()


def join_vector_backward():
    """Issue the deferred vector-sized backward chains of all layers (no-op if there are none).

    A deferred chain writes into the tensors its layer's backward RETURNED; that is only sound if autograd adopted each of
    them as the parameter's ``.grad`` (AccumulateGrad takes over a gradient it holds the only reference to).  This is
    checked here for every deferred gradient before anything is launched: a parameter whose ``.grad`` is not the very
    tensor the chain will write (it was copied, summed with another contribution, or the pass was ``autograd.grad``)
    would otherwise receive garbage while the kernel writes freed memory -- that raises instead."""
    bad = [name for (name, p, ptr) in _OVERLAP["adopt"] if p.grad is None or p.grad.data_ptr() != ptr]
    if bad:
        _drop_deferred()
        raise RuntimeError("bnn_amd: a deferred vector-backward gradient was not adopted as .grad (%s ...): the backward "
                           "pass ran under conditions vector_backward_overlap does not support; nothing was launched, "
                           "the gradients of this step are incomplete" % ", ".join(bad[:3]))
    _OVERLAP["adopt"].clear()
    # the second level of the layers' column sums (Sum_b G, dz_k, dz_2, dr0_c) first: the chains below read them
    if _OVERLAP["sums"]:
        ops.reduce_partials_flush(_OVERLAP["sums"])
    if _OVERLAP["planar"]:
        ops.mnf_flow_planar_backward_flush(_OVERLAP["planar"])
    if _OVERLAP["dense"]:
        ops.mnf_flow_dense_backward_flush(_OVERLAP["dense"])


class vector_backward_overlap:
    """Context manager: backward passes inside defer their vector-sized chains; issued (batched over layers) on exit.
    The gradients of the vector-sized parameters (q0_*, r0_*, flow parameters, biases) are complete only after the exit:
    do not combine with anything that READS gradients during the backward pass (gradient hooks, DistributedDataParallel's
    bucketed all-reduce); bnn_amd.parallel's explicit all_reduce_grads() after the block is fine."""

    def __enter__(self):
        self._was = _OVERLAP["on"]
        _OVERLAP["on"] = True
        return self

    def __exit__(self, exc_type, exc, tb):
        _OVERLAP["on"] = self._was
        if exc_type is not None:
            # the backward pass did not finish: autograd has released (some of) the tensors the filed chains would write
            _drop_deferred()
            return False
        join_vector_backward()
        return False


class _NodeToken:
    """One per autograd forward of a layer; ``multi`` is set on every live token of a layer as soon as a second one
    appears, i.e. when the layer takes part more than once in the graph a backward pass will walk."""
    __slots__ = ("multi", "__weakref__")

    def __init__(self):
        self.multi = False


def _has_grad_hooks(p):
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


def _deferral_ok(layer, token):
    """Deferred chains write their outputs AFTER the backward function has returned them, so autograd must ADOPT those
    tensors as ``.grad`` rather than read them.  That is the case only when
      * the backward is not building a graph,
      * every vector-sized parameter's ``.grad`` is None (``zero_grad(set_to_none=True)``),
      * the layer was run ONCE in the graph being walked (a second application -- a multi-sample ELBO, a loss that calls
        the net twice -- makes autograd SUM the two returned tensors before any flush: one unwritten operand, the other
        freed afterwards),
      * no tensor hook or post-accumulate hook would read the gradient during the pass.
    Otherwise the chains run in place, as without the context manager."""
    if not _OVERLAP["on"] or torch.is_grad_enabled() or token is None or token.multi:
        return False
    vec = layer._param_list()[3:]
    return all(p.grad is None and p.requires_grad and p.is_leaf and not _has_grad_hooks(p) for p in vec)


def _file_adoption_checks(layer, grads):
    """Remember (parameter, address of the returned gradient) for join_vector_backward's adoption check."""
    vec = layer._param_list()[3:]
    assert len(vec) == len(grads)
    for k, (p, g) in enumerate(zip(vec, grads)):
        _OVERLAP["adopt"].append(("%s[%d]" % (type(layer).__name__, k), p, g.data_ptr()))


_layer_ids = itertools.count()


class _BayesLinearFn(torch.autograd.Function):
    """Forward: HIP kernels.  Backward (hybrid): the four large products run on the same HIP GEMM kernels
    as the forward --
        dX   = G_m . W_m + 2 x (.) (G_v . W_v)        G_m = g (.) relu-mask,  G_v = G_m eps / (2 std)
        dW_m = G_m^T . x ,   dW_v = G_v^T . x^2
    -- and only the O(O*I) parameter chain (operands -> mu, rho, lambda, z, flows) and the KL are
    differentiated by torch autograd on the GPU (``_grad.*_param_graph``).  Measured on the headline net:
    see DESIGN.md."""

    @staticmethod
    def forward(ctx, layer, x, cfg, *params):
        if layer._mnf:
            # the backward needs this call's K3 / K1 by-products: give the call fresh output vectors instead of
            # copying them out of the (reused) workspace afterwards
            ws = layer._workspace()
            # (the network may already have run this call's dense flows for all layers at once: _preflow_dense)
            fresh = ("act_mu", "act_var") if layer._preflow is not None else ("z_fwd", "z_kl", "scal", "act_mu", "act_var")
            if layer._preprep is not None:
                fresh = ()                          # the network gave this call its buffers before it ran K3 / K1
            for name in fresh:
                setattr(ws, name, torch.empty_like(getattr(ws, name)))
        dense = layer._mnf and layer._check_flows() == "dense" and _DENSE_HIP_BWD
        layer._keep_dense = dense
        layer._last_flow_rng = None
        try:
            out, kl, saved = layer._forward_hip(x, cfg, advance=layer._advance_rng, save_rng=True, want_std=cfg[0])
        finally:
            layer._keep_dense = False
            layer._preflow = None
            layer._preprep = None
        if dense:
            saved["dense_save"], layer._last_dense_save = layer._last_dense_save, None
            saved["rng_flow"], layer._last_flow_rng = layer._last_flow_rng, None
        if layer._mnf:
            saved["z_fwd"] = ws.z_fwd
            if cfg[1]:
                saved["act_mu"], saved["act_var"], saved["z_kl"], saved["scal"] = ws.act_mu, ws.act_var, ws.z_kl, ws.scal
        layer._v1_slot = None
        if layer._mnf and cfg[1] and _V1_BATCH and not saved.get("noise") and saved.get("rng") is not None \
                and (layer._check_flows() == "planar" or dense) and "r0_b1" in layer._vec_names:
            # V1 (lbbnn_mnf_aux_backward) reads by-products of THIS forward only: the backward of the network's KL sum
            # (losses._SumKLFn), which learns d loss / d kl for all layers at once, runs all layers' V1 in one launch and
            # leaves the results in this slot for the layer's own backward
            slot = {"in": dict(act_mu=saved["act_mu"], act_var=saved["act_var"], zb_last=saved["scal"][3:4], rng=saved["rng"],
                               r0_b1=layer.r0_b1, r0_b2=layer.r0_b2, layer_id=layer._layer_id)}
            saved["v1_slot"] = layer._v1_slot = slot
        ctx.layer, ctx.cfg, ctx.saved = layer, cfg, saved
        from . import graphs
        graphs.mark_autograd_node(ctx, layer)          # capture guard: graphs.assert_no_live_graph
        # how many autograd nodes of this layer are alive: the deferral of its vector chain needs exactly one
        live = layer.__dict__.setdefault("_live_nodes", weakref.WeakSet())
        ctx.token = _NodeToken()
        if len(live):
            ctx.token.multi = True
            for t in live:
                t.multi = True
        live.add(ctx.token)
        std = saved.pop("std", None)
        ctx.has_std = std is not None
        ctx.save_for_backward(x, out, *([std] if std is not None else []), *params)
        if kl is None:
            kl = out.new_zeros(())
        return out, kl

    @staticmethod
    def backward(ctx, g_out, g_kl):
        layer, cfg = ctx.layer, ctx.cfg
        token = getattr(ctx, "token", None)
        can_defer = _deferral_ok(layer, token)
        if token is not None:
            layer.__dict__.get("_live_nodes", set()).discard(token)
        stochastic, want_kl, relu = cfg
        tens = list(ctx.saved_tensors)
        x, out = tens[0], tens[1]
        std = tens[2] if ctx.has_std else None
        params = tens[3 if ctx.has_std else 2:]
        mu, rho, lam = params[0], params[1], params[2]
        B = x.shape[0]
        # ---- (B,O) head: mask, G_v, transposes and column sums in one HIP pass; eps_out is re-created from the
        # forward's Philox state inside the kernel unless explicit draws were given
        explicit = ctx.saved.get("noise") or {}
        if ctx.saved.get("lsm"):
            from . import losses
            ent = losses._LOGITS_GRAD.pop(out.data_ptr(), None)
            if ent is not None and ent[0] == g_out.data_ptr():
                g_out = ent[1]                                        # formed by the loss's own backward launch (losses._LOGITS_GRAD)
            else:
                g_out = ops.log_softmax_backward(g_out, out)          # grad wrt log-probabilities -> grad wrt the logits
        planar = layer._mnf and layer._check_flows() == "planar"
        dense = layer._mnf and ctx.saved.get("dense_save") is not None
        # With the vector chains deferred (they are the only readers of the column sums Sum_b G_m / G_v, dz_k, dz_2, dr0_c)
        # the second level of those sums is deferred with them: one lbbnn_reduce_partials_batch launch for all layers
        # instead of two ~5 us launches per layer
        defer_sums = _OVERLAP["sums"] if (can_defer and (planar or dense) and _DEFER_SUMS) else None
        # an LRT layer: the sums' only reader is the bias backward, which takes the partials as they lie (one launch for both)
        lrt_job = [] if (not layer._mnf and _LRT_BIAS_HIP and B > 0) else None
        g, g_v, gT, g_vT, g_sum, gv_sum = ops.output_grad(
            g_out, out=out if relu else None, std=std if stochastic else None, eps=explicit.get("eps_out"),
            rng=ctx.saved.get("rng"), rng_stream=ops.STREAM_EPS_OUT * 64 + layer._layer_id, row_offset=layer.row_offset,
            relu=relu, want_g=bool(ctx.needs_input_grad[1]), defer_sums=lrt_job if lrt_job is not None else defer_sums)
        # planar / dense MNF layers re-create their small draws inside V1 / V2; the torch-graph paths need them as tensors
        rng_snap = ctx.saved.get("rng")
        in_kernel = (planar or dense) and not explicit and rng_snap is not None
        noise = {} if in_kernel else layer._noise_for_backward(ctx.saved, B, need_out=False)
        g_kl = g_kl.contiguous() if want_kl else None
        da_mu = da_var = aux = r0_c = vg = None
        if planar or dense:
            # ---- all-HIP chain: V1 (aux activations) -> GEMMs -> K1b -> V2 (flows, q0, r0_b, bias)
            P = dict(zip(layer._vec_names, params[3:3 + len(layer._vec_names)]))
            z_k = ctx.saved["z_fwd"]
            z2 = ctx.saved["z_kl"] if want_kl else None
            if want_kl:
                r0_c = P["r0_c"]
                slot = ctx.saved.get("v1_slot")
                if slot is not None and "out" in slot and slot["out"][3].data_ptr() == g_kl.data_ptr():
                    da_mu, da_var, aux = slot.pop("out")[:3]               # run with the other layers' by losses._SumKLFn
                else:
                    da_mu, da_var, aux = ops.mnf_aux_backward(ctx.saved["act_mu"], ctx.saved["act_var"], noise.get("eps_act"),
                                                              P["r0_b1"], P["r0_b2"], ctx.saved["scal"][3:4], g_kl,
                                                              rng=rng_snap, layer_id=layer._layer_id)
        elif not layer._mnf and _LRT_BIAS_HIP:
            z_k = z2 = None                          # an LRT layer: bias vectors only (lbbnn_bias_backward, below)
        else:
            # ---- vector-sized graph (flows, q/r densities, bias terms) under autograd: O(I + O) work
            with torch.enable_grad():
                vs = [p.detach().requires_grad_(True) for p in params[3:]]
                am = av = None
                if layer._mnf and want_kl:
                    am = ctx.saved["act_mu"].detach().requires_grad_(True)
                    av = ctx.saved["act_var"].detach().requires_grad_(True)
                vg = layer._vector_graph(vs, cfg, noise, am, av)
            z_k = vg["z_k"].detach() if vg["z_k"] is not None else None
            z2 = vg["z2"].detach() if vg["z2"] is not None else None
            if am is not None:
                r0_c = params[3 + layer._vec_names.index("r0_c")]
                da_mu, da_var = torch.autograd.grad(vg["kl"], [am, av], g_kl, retain_graph=True)
        # ---- the four big products on the HIP GEMM kernels
        gx = None
        if ctx.needs_input_grad[1]:
            # (e_w z)^T and var_w^T straight from the parameters in one pass (lbbnn_weight_operands_t)
            from . import _lib
            O, I = layer.out_features, layer.in_features
            split_m = _operand_split_ok(g, O, I, layer)
            split_v = _operand_split_ok(g_v, O, I, layer) if stochastic else split_m
            ld = ops.operand_ld(O)
            e_t = torch.empty((I, ld), dtype=torch.float32, device=x.device)
            v_t = torch.empty((I, ld), dtype=torch.float32, device=x.device) if stochastic else None
            if split_m == split_v:
                _lib.check(_lib.lib().lbbnn_weight_operands_t(
                    mu.data_ptr(), rho.data_ptr(), lam.data_ptr(), z_k.data_ptr() if z_k is not None else None,
                    e_t.data_ptr(), v_t.data_ptr() if v_t is not None else None, ld, O, I,
                    ops.F_SPLIT16 if split_m else 0, torch.cuda.current_stream(x.device).cuda_stream), "lbbnn_weight_operands_t")
                w_shape = torch.empty((O, I), device="meta")       # shape carrier for _hip_matmul_nt
                if O <= 16 and not split_m and x.stride(1) == 1 and g.stride(0) == (g_v.stride(0) if stochastic else g.stride(0)):
                    # the 10-class head: ten multiply-adds per output are one elementwise-shaped launch, not two GEMMs
                    gx = ops.head_dx(g, g_v if stochastic else None, e_t, v_t if stochastic else None, x, C=O, I=I)
                    stochastic_done = True
                else:
                    stochastic_done = False
                    gx = _hip_matmul_nt(g, None, w_shape, op=e_t, module=layer)
                if stochastic and not stochastic_done:
                    if I > 16 and x.stride(1) == 1:
                        # dX = G_m.W_m + 2 x (.) (G_v.W_v): the combination is the second product's epilogue
                        gx = ops.lrt_gemm_combine(g_v, v_t, K=O, N=I, comb_x=x, comb_add=gx, split=split_v)
                    else:
                        gx = ops.dx_combine(gx, _hip_matmul_nt(g_v, None, w_shape, op=v_t, module=layer), x)
            else:
                ws = layer._workspace()
                bw = ws.backward_operands()
                ops.weight_pass(mu, rho, lam, z_fwd=z_k, priors=layer.priors, e_w=bw[0],
                                var_w=bw[1] if stochastic else None)
                gx = _hip_matmul_nt(g, ops.transpose_operand, bw[0][:, :I], module=layer)
                if stochastic:
                    gx = ops.dx_combine(gx, _hip_matmul_nt(g_v, ops.transpose_operand, bw[1][:, :I], module=layer), x)
        if (_HEAD_DW and layer.out_features <= 16 and g is not None and (g_v is not None or not stochastic) and x.dim() == 2
                and x.stride(1) == 1 and x.dtype == torch.float32 and B >= 64):
            # the <= 16-class head: split-K slabs of both weight gradients straight from the row-major G and x (lbbnn_head_dw) --
            # no x^T | (x^2)^T operand pass, no 16-row GEMM tiles that are 37 % padding at 10 classes
            # (K1b adds slabs whose stride O * I is a multiple of 4 floats; other shapes get one slab)
            dWm, dWv = ops.head_dw(g, g_v if stochastic else None, x,
                                   nslabs=16 if (layer.out_features * layer.in_features) % 4 == 0 else 1)
            if dWm.shape[0] == 1:
                dWm, dWv = dWm[0], (dWv[0] if dWv is not None else None)
        else:
            pair = _x_operand_pair(x, gT, g_vT, layer) if stochastic else None
            dWm = _hip_matmul_nt(gT, ops.transpose_operand, x, allow_splitk=True, op=pair[0] if pair else None, module=layer)
            dWv = (_hip_matmul_nt(g_vT, ops.transpose_operand, x, square=True, allow_splitk=True, op=pair[1] if pair else None,
                                  module=layer) if stochastic else None)
        # ---- K1b: the whole (O,I) chain in one pass
        dmu, drho, dlam, dz_k, dz2, dr0c = ops.weight_pass_backward(
            mu, rho, lam, dWm, dWv, z_fwd=z_k, z_kl=z2, r0_c=r0_c, da_mu=da_mu, da_var=da_var, g_kl=g_kl,
            priors=layer.priors, defer_sums=defer_sums)
        if not layer._mnf and _LRT_BIAS_HIP:
            # the two bias gradients in one small launch (until round 3: a torch autograd graph over the bias vectors,
            # ~25 launches per layer inside a captured training step)
            if lrt_job:
                d_bmu, d_brho = ops.bias_backward_partials(lrt_job[0], B, params[3], params[4], g_kl, layer.priors)
            else:
                d_bmu, d_brho = ops.bias_backward(params[3], params[4], g_sum, gv_sum if stochastic else None, g_kl, layer.priors)
            return (None, gx, None, dmu, drho, dlam, d_bmu, d_brho)
        if planar:
            zp, rp = layer._planar_params_from(params)
            e1 = None if in_kernel else noise["eps_z"].contiguous()
            e2 = noise["eps_z2"].contiguous() if (want_kl and not in_kernel) else None
            G = ops.mnf_flow_planar_backward(
                P["q0_mean"], P["q0_log_var"], zp, rp, eps_fwd=e1, eps_kl=e2,
                rng=rng_snap, layer_id=layer._layer_id, r0_b1=P["r0_b1"], r0_b2=P["r0_b2"], aux=aux,
                dz_fwd=dz_k, dz_kl=dz2, g_kl=g_kl, bias_mu=P["bias_mu"], bias_rho=P["bias_rho"], g_sum=g_sum,
                gv_sum=gv_sum, priors=layer.priors, defer=_OVERLAP["planar"] if can_defer else None)
            G["r0_c"] = dr0c if dr0c is not None else torch.zeros_like(P["q0_mean"])
            vgrads = [G[n] for n in layer._vec_names]
            for key in ("z_flow", "r_flow"):
                for g3 in G[key]:
                    vgrads += list(g3)
            if can_defer:
                _file_adoption_checks(layer, vgrads)
            return (None, gx, None, dmu, drho, dlam, *vgrads)
        if dense:
            masks = ctx.saved["masks"]
            zd, Tz, k1 = layer.z_flow.dense_descs(masks["zmask"], masks.get("zmask2"))
            # no KL branch: the forward ran (and laid out what it kept) with Tr = 0; the r-flow gradients are zeros
            rd, Tr, k2 = layer.r_flow.dense_descs(None, masks.get("rmask")) if want_kl else (None, 0, None)
            rest = list(params[len(layer._names):])
            nz = len(list(layer.z_flow.parameters()))
            e1 = None if in_kernel else noise["eps_z"].contiguous()
            e2 = noise["eps_z2"].contiguous() if (want_kl and not in_kernel) else None
            rng_f = ctx.saved.get("rng_flow") if ctx.saved.get("rng_flow") is not None else rng_snap
            G = ops.mnf_flow_dense_backward(
                P["q0_mean"], P["q0_log_var"], zd, Tz, 0 if layer.z_flow.kind == "RNVP" else 1, rest[:nz],
                rd, Tr, 0 if layer.r_flow.kind == "RNVP" else 1, rest[nz:], save=ctx.saved["dense_save"],
                eps_fwd=e1, eps_kl=e2, rng=rng_f, layer_id=layer._layer_id, r0_b1=P["r0_b1"], r0_b2=P["r0_b2"], aux=aux,
                dz_fwd=dz_k, dz_kl=dz2, g_kl=g_kl, bias_mu=P["bias_mu"], bias_rho=P["bias_rho"], g_sum=g_sum,
                gv_sum=gv_sum, priors=layer.priors, defer=_OVERLAP["dense"] if can_defer else None,
                keep=(k1, k2))
            del k1, k2
            G["r0_c"] = dr0c if dr0c is not None else torch.zeros_like(P["q0_mean"])
            if not want_kl:
                G["r_flow"] = [torch.zeros_like(p) for p in rest[nz:]]
            vgrads = [*[G[n] for n in layer._vec_names], *G["z_flow"], *G["r_flow"]]
            if can_defer:
                _file_adoption_checks(layer, vgrads)
            return (None, gx, None, dmu, drho, dlam, *vgrads)
        outs, gouts = [vg["bmean"]], [g_sum]
        if stochastic:
            outs.append(vg["bvar"]); gouts.append(gv_sum)
        if z_k is not None:
            outs.append(vg["z_k"]); gouts.append(dz_k)
        if want_kl and vg["kl"] is not None:
            outs.append(vg["kl"]); gouts.append(g_kl)
            if z2 is not None:
                outs.append(vg["z2"]); gouts.append(dz2)
        vgrads = list(torch.autograd.grad(outs, vs, gouts, allow_unused=True))
        if dr0c is not None:
            k = layer._vec_names.index("r0_c")
            vgrads[k] = dr0c if vgrads[k] is None else vgrads[k] + dr0c
        return (None, gx, None, dmu, drho, dlam, *vgrads)


def _operand_split_ok(a, K, N, module=None):
    """bf16x3 operands for a backward / mean-only product of ``module`` (any 16-bit precision selects them)?"""
    return (ops.split_precision(module) and ops.split_eligible(K, N)
            and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0)


def _hip_matmul_nt(a, transpose_fn, w, square=False, allow_splitk=False, op=None, module=None):
    """a (M,K) @ f(w) (K,N) with f = identity or square, through lbbnn_lrt_gemm (mean-only):
    the operand is f(w)^T = [N][ld(K)], built by lbbnn_transpose_operand (or handed in as ``op``).  With
    ``allow_splitk`` a long contraction with few output tiles (the weight gradients: K = batch) is cut into k ranges
    that fill the chip; the result is then (S,M,N) slabs for the consumer (lbbnn_weight_pass_backward) to add."""
    M, K = a.shape
    N = w.shape[1]
    assert w.shape[0] == K
    split = _operand_split_ok(a, K, N, module)
    if op is None:
        op = transpose_fn(w if w.stride(1) == 1 else w.contiguous(), square=square, split=split)
    if allow_splitk and split and K >= 1024:
        tiles = ((N + 79) // 80) * ((M + 127) // 128 if M >= 96 else (M + 31) // 32)
        S = max(1, min(16, (400 + tiles - 1) // tiles, K // 256))
        if S > 1:
            kchunk = ((K + S - 1) // S + 31) // 32 * 32
            return ops.matmul_splitk(a, op, K=K, N=N, kchunk=kchunk)
    return ops.lrt_gemm(a, op, None, I=K, O=N, mean_only=True, split=split)


def _x_operand_pair(x, gT, g_vT, module=None):
    """x^T and (x^2)^T as GEMM operands from ONE pass over x (lbbnn_vd_operands: the variational-dropout kernel does
    exactly this for theta) when both weight-gradient products use the same operand format; else None."""
    from . import _lib
    B, I = x.shape
    if not x.is_contiguous():
        return None
    split = _operand_split_ok(gT, B, I, module)
    if split != _operand_split_ok(g_vT, B, I, module):
        return None
    ld = ops.operand_ld(B)
    xt = torch.empty((I, ld), dtype=torch.float32, device=x.device)
    x2t = torch.empty((I, ld), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().lbbnn_vd_operands(x.data_ptr(), xt.data_ptr(), x2t.data_ptr(), ld, B, I,
                                            ops.F_SPLIT16 if split else 0, torch.cuda.current_stream(x.device).cuda_stream),
               "lbbnn_vd_operands")
    return xt, x2t


class _BayesLinearBase(nn.Module):
    _mnf = False

    def _common_init(self, in_features, out_features, mu_range, priors):
        self.in_features = in_features
        self.out_features = out_features
        self.priors = priors if priors is not None else Priors()
        # creation order == reference (LBBNN-GP-MF-LRT.py:137-155): mu, rho, lambdal, alpha_q, bias_mu, bias_rho
        self.weight_mu = nn.Parameter(torch.Tensor(out_features, in_features).uniform_(-mu_range, mu_range))
        self.weight_rho = nn.Parameter(torch.Tensor(out_features, in_features).uniform_(-5, -4))
        self.weight = Gaussian(self.weight_mu, self.weight_rho)
        self.lambdal = nn.Parameter(torch.Tensor(out_features, in_features).uniform_(0, 1))
        self._alpha_q_init = torch.Tensor(out_features, in_features).uniform_(0.999, 0.9999)   # placeholder draw, :147
        self.gamma = Bernoulli(self._alpha_q_init)
        self.gamma.bind(self._alpha_now)
        self.bias_mu = nn.Parameter(torch.Tensor(out_features).uniform_(-0.2, 0.2))
        self.bias_rho = nn.Parameter(torch.Tensor(out_features).uniform_(-5, -4))
        self.bias = Gaussian(self.bias_mu, self.bias_rho)
        self.kl = 0
        self.noise: Optional[Dict[str, torch.Tensor]] = None
        self.row_offset = 0            # global row index of this rank's first batch row (data parallel)
        self._layer_id = next(_layer_ids) % 64
        self._ws = None
        self._split_now = False        # decided per forward (prep and GEMM must agree on the operand format)
        self._keep_dense = False       # set by the autograd forward: dense flows keep their intermediates
        self._last_dense_save = None
        self._preflow = None           # dense flows of this call already run by the network (batched over its layers)
        self._advance_rng = True       # False while a network drives the layers: it advances the shared offset once
        self._last_flow_rng = None
        self._mask_pool = None         # Bernoulli masks pre-drawn by the network for this call (one launch for all layers)
        self._preprep = None           # K3 / K1 of this call already run by the network (one launch per kind for all layers)
        self._lsm_now = False          # set by the network around the head's training forward: log_softmax in the GEMM epilogue
        self._last_masks = None

    # reference keeps the prior tensors as attributes; expose them lazily with the same names
    @property
    def mu_prior(self):
        return torch.full_like(self.weight_mu, self.priors.mu_prior).detach()

    @property
    def sigma_prior(self):
        return torch.full_like(self.weight_mu, self.priors.sigma_prior).detach()

    @property
    def alpha_prior(self):
        return torch.full_like(self.weight_mu, self.priors.alpha_prior).detach()

    def _workspace(self):
        dev = self.weight_mu.device
        if self._ws is None or self._ws.device != dev:
            self._ws = ops.LayerWorkspace(self.out_features, self.in_features, dev, self._mnf)
        return self._ws

    def _alpha_now(self):
        return 1 / (1 + torch.exp(-self.lambdal))

    precision = None              # None: the process-wide default (ops.set_precision); else one of ops.PRECISIONS

    def _split(self, x=None, cfg=None):
        """Operand format of this layer call: 0 fp32, 1 bf16 hi | lo (bf16x3 and the reduced single-product modes), 2
        row-scaled fp16 hi | lo (fp16x3), 3 the same with var_w as its hi part alone (fp16x3f: one variance product).  2 / 3
        serve the dual-moment GEMM only -- a posterior-mean call takes format 1."""
        if not ops.split_precision(self) or not ops.split_eligible(self.in_features, self.out_features):
            return 0
        if x is not None and (x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0 or x.stride(1) != 1
                              or x.shape[0] * x.stride(0) * 4 >= 0x7FFFFFF0):      # 32-bit buffer offsets in the split kernel
            return 0
        if (ops.f16s_precision(self) and (cfg is None or cfg[0]) and ops.f16s_eligible(self.in_features, self.out_features)
                and getattr(self, "_f16s_net_ok", True)):
            # the single variance product of "fp16x3f" rounds s and var_w to 11 bits each: that averages out over a long row
            # (1.4-1.8e-5 of max|out| at I = 784 ... 1200, ~1 / sqrt(I)) but not over a short one (6.5e-5 measured at I = 8,
            # 4.3-4.7e-5 at I = 72 ... 104: tools/gemm16_fuzz.py), so short rows -- whose GEMMs cost nothing -- take all three
            return 3 if (ops.get_precision(self) == "fp16x3f" and self.in_features >= ops.F16_VAR1_MIN_I) else 2
        return 1

    def _single(self):
        return ops.get_precision(self) in ("bf16", "fp16")

    @property
    def alpha_q(self):
        """sigmoid(lambdal) (…LRT.py:167).  The reference refreshes this attribute (and
        ``gamma.alpha``) inside forward; its eval code only reads them (…LRT.py:242-246), so they are
        computed on access with torch ops, off the hot path."""
        return self._alpha_now()

    def _param_list(self):
        raise NotImplementedError

    def _fusable(self):
        return True

    def _fill_desc(self, d, cfg, kl_layer):
        """Fill one lbbnn_layer_desc_t for lbbnn_layers_prepare (pointers into parameters / workspace)."""
        ws = self._workspace()
        noise = self.noise or {}
        d.weight_mu, d.weight_rho, d.lambdal = self.weight_mu.data_ptr(), self.weight_rho.data_ptr(), self.lambdal.data_ptr()
        d.bias_mu, d.bias_rho = self.bias_mu.data_ptr(), self.bias_rho.data_ptr()
        d.priors = self.priors
        d.O, d.I, d.layer_id = self.out_features, self.in_features, self._layer_id
        d.stochastic, d.want_kl = int(cfg[0]), int(cfg[1])
        d.split = int(self._split_now)
        d.e_w, d.var_w = ws.e_w.data_ptr(), ws.var_w.data_ptr()
        d.e_scale, d.v_scale = ws.e_scale.data_ptr(), ws.v_scale.data_ptr()
        d.kl_rows, d.bias_var = ws.kl_rows.data_ptr(), ws.bias_var.data_ptr()
        d.kl_layer = kl_layer.data_ptr() if kl_layer is not None else None
        d.q0_mean = None
        return ws, noise

    def _gemm(self, x, cfg, rng, log_softmax=False, std_out=None, finalize=None, out=None, x_planes=False,
              out_planes=None, want_out=True, head=None):
        """The layer's GEMM launch on the operands ``_prep`` (or the network's batched prepare) left in the workspace.
        Format 2 (row-scaled fp16): lbbnn_lrt_gemm_ex -- ``x`` may be the plane buffer the previous layer wrote
        (``x_planes``), and the call may write planes for the next layer (``out_planes``) instead of / beside fp32 ``out``
        (``want_out``); returns ``out`` (None when only planes were asked for)."""
        stochastic, _, relu = cfg
        ws = self._workspace()
        eps = (self.noise or {}).get("eps_out")
        stream_id = ops.STREAM_EPS_OUT * 64 + self._layer_id
        if self._split_now >= 2:
            assert stochastic and not log_softmax
            o, _ = ops.lrt_gemm16(x, ws.e_w, ws.var_w, ws.e_scale, ws.v_scale, I=self.in_features, O=self.out_features,
                                  bias_mean=self.bias_mu, bias_var=ws.bias_var, eps=eps, rng=rng, rng_stream=stream_id,
                                  row_offset=self.row_offset, relu=relu, var1=(self._split_now == 3), x_planes=x_planes, out=out,
                                  want_out=want_out, out_planes=out_planes, std_out=std_out, finalize=finalize, head=head)
            return o
        return ops.lrt_gemm(x, ws.e_w, ws.var_w, I=self.in_features, O=self.out_features,
                            bias_mean=self.bias_mu, bias_var=ws.bias_var, eps=eps, rng=rng,
                            rng_stream=stream_id, row_offset=self.row_offset,
                            relu=relu, mean_only=not stochastic, log_softmax=log_softmax, split=(self._split_now == 1),
                            std_out=std_out, finalize=finalize, out=out, single=self._single())

    def _forward_hip(self, x, cfg, advance=True, save_rng=False, want_std=False):
        """One layer, sequentially on the current stream: prep kernels, GEMM, RNG advance.
        ``save_rng`` snapshots the device RNG state so backward can re-create the in-kernel draws;
        ``want_std`` also stores sqrt(var) (needed by the backward)."""
        rng, st = None, None
        saved = {"noise": self.noise}
        if self._uses_rng(cfg):
            st = ops.RngState.get(x.device)
            rng = st.t
            pre_ = self._preprep
            # (a network that drives its layers takes ONE snapshot of the shared state for all of them)
            saved["rng"] = (pre_["snap"] if (pre_ is not None and pre_.get("snap") is not None) else rng.clone()) if save_rng else None
        kl = torch.empty((), dtype=torch.float32, device=x.device) if cfg[1] else None
        self._cur_B = x.shape[0]
        self._split_now = self._split(x, cfg)
        self._last_masks = None
        # the layer's KL tail (K5) depends on parameters only: it rides in the GEMM's launch (lbbnn_lrt_gemm_finalize)
        # instead of a launch of its own between the weight pass and the GEMM
        fin, hosted_all = None, False
        pre = self._preprep
        if pre is not None and pre["cfg"][:2] == tuple(cfg[:2]) and pre["split"] == self._split_now:
            # K3 / K1 of this call were run by the network for all layers at once (_NetworkBase._preprep_all)
            if self._preflow is not None:
                fl = self._preflow
                self._last_masks, self._last_dense_save, self._last_flow_rng = fl["masks"], fl["save"], fl["rng"]
            sh = pre["shared"]
            if cfg[1] and sh["fin_all"] is not None and (sh["hosted"] or pre["first"]):
                # every layer's KL tail rides in the FIRST layer's GEMM launch (one extra workgroup for the network)
                kl = pre["kl"]
                if pre["first"]:
                    fin, sh["hosted"] = sh["fin_all"], True
                    if _ADV_RIDE and pre.get("snap") is not None and rng is not None and sh.get("all_snap"):
                        # ... and so does the forward's RNG advance: every kernel of this forward from here on reads the
                        # SNAPSHOT the weight pass took (same {seed, offset}), so the live offset may move already
                        fin = (fin[0], fin[1], pre["snap"].data_ptr(), fin[3], rng.data_ptr(), 1)
                        sh["advanced"] = True
                hosted_all = True
            if sh.get("advanced"):
                rng = pre["snap"]                     # the GEMM's noise from the snapshot (the live offset is, or is being, advanced)
        else:
            self._prep(cfg, rng, kl_layer=kl, finalize=not cfg[1])
        if cfg[1] and not hosted_all:
            from . import _lib
            desc = (_lib.LayerDesc * 1)()
            saved["_keep"] = self._fill_desc(desc[0], cfg, kl)         # reshaped noise views: alive past the launch
            fin = (desc, 1, rng.data_ptr() if rng is not None else None, None)
        saved["masks"] = self._last_masks
        std = torch.empty((x.shape[0], self.out_features), dtype=torch.float32, device=x.device) if want_std else None
        out = self._gemm(x, cfg, rng, std_out=std, finalize=fin, log_softmax=bool(self._lsm_now and self._split_now == 0))
        saved["lsm"] = bool(self._lsm_now and self._split_now == 0)
        if self._lsm_now and not saved["lsm"]:
            out = F.log_softmax(out, dim=1)               # (a head the skinny kernel does not take: > 16 classes never get here)
        if std is not None:
            saved["std"] = std
        if st is not None and advance:
            st.advance(1)
        return out, kl, saved

    def forward(self, input, sample=False, calculate_log_probs=False, *, _relu=False):
        if not input.is_cuda:
            raise RuntimeError("bnn_amd: forward needs a HIP device tensor (input is on %s); there is no CPU path"
                               % input.device)
        cfg = (bool(self.training or sample), bool(self.training or calculate_log_probs), bool(_relu))
        x = input if input.dtype == torch.float32 else input.float()
        params = self._param_list()
        try:
            if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
                if getattr(self, "as_written", False):
                    raise RuntimeError("bnn_amd: as_written is a no-grad (evaluation / timing) mode")
                out, kl = _BayesLinearFn.apply(self, x, cfg, *params)
            else:
                self._preflow = None
                out, kl, _ = self._forward_hip(x, cfg, advance=self._advance_rng)
        finally:
            self._preflow = None
            self._preprep = None
            self._mask_pool = None
        self.kl = kl if cfg[1] else 0
        return out


class LRTBayesianLinear(_BayesLinearBase):
    """BayesianLinear(in_features, out_features) of LBBNN-GP-MF-LRT.py:129-197."""

    def __init__(self, in_features, out_features, *, priors: Optional[Priors] = None):
        super().__init__()
        self._common_init(in_features, out_features, 0.2, priors)

    def _param_list(self):
        return [self.weight_mu, self.weight_rho, self.lambdal, self.bias_mu, self.bias_rho]

    _names = ("weight_mu", "weight_rho", "lambdal", "bias_mu", "bias_rho")

    def _uses_rng(self, cfg):
        return cfg[0] and not (self.noise and "eps_out" in self.noise)

    def _prep(self, cfg, rng, kl_layer=None, kl_total=None, accumulate=False, finalize=True):
        """x-independent kernels: K1 weight pass (+ K5 KL finalize unless deferred to ``_finalize``)."""
        stochastic, want_kl, _ = cfg
        ws = self._workspace()
        ops.weight_pass(self.weight_mu, self.weight_rho, self.lambdal, bias_rho=self.bias_rho,
                        priors=self.priors, e_w=ws.e_w, var_w=ws.var_w if stochastic else None,
                        kl_rows=ws.kl_rows if want_kl else None, bias_var=ws.bias_var, split=self._split_now,
                        e_scale=ws.e_scale, v_scale=ws.v_scale)
        if want_kl and finalize:
            self._finalize(rng, kl_layer, kl_total, accumulate)

    def _finalize(self, rng, kl_layer=None, kl_total=None, accumulate=False):
        ws = self._workspace()
        ops.kl_finalize(ws.kl_rows, self.bias_mu, self.bias_rho, priors=self.priors, kl_layer=kl_layer,
                        kl_out=kl_total, accumulate=accumulate)

    def _noise_for_backward(self, saved, B, need_out=True):
        if saved.get("noise") and "eps_out" in saved["noise"]:
            return saved["noise"]
        if saved.get("rng") is None or not need_out:
            return {}
        return {"eps_out": ops.philox_normal(saved["rng"], ops.STREAM_EPS_OUT * 64 + self._layer_id,
                                             B, self.out_features, self.row_offset)}

    _vec_names = ("bias_mu", "bias_rho")

    def _vector_graph(self, vs, cfg, noise, act_mu=None, act_var=None):
        P = dict(zip(self._vec_names, vs))
        return _grad.lrt_vector_graph(P, stochastic=cfg[0], want_kl=cfg[1], priors=self.priors)


class MNFBayesianLinear(_BayesLinearBase):
    """BayesianLinear(in_features, out_features, num_transforms) of LBBNN-GP-MF-MNF.py:133-239."""
    _mnf = True

    def __init__(self, in_features, out_features, num_transforms, *, z_flow_type="RNVP", r_flow_type="RNVP",
                 priors: Optional[Priors] = None):
        super().__init__()
        self._common_init(in_features, out_features, 0.01, priors)         # weight_mu ~ U(+-0.01), :140
        self.q0_mean = nn.Parameter(0.1 * torch.randn(in_features))         # :166
        self.q0_log_var = nn.Parameter(-9 + 0.1 * torch.randn(in_features))  # :167
        self.r0_c = nn.Parameter(0.1 * torch.randn(in_features))            # :170-172
        self.r0_b1 = nn.Parameter(0.1 * torch.randn(in_features))
        self.r0_b2 = nn.Parameter(0.1 * torch.randn(in_features))
        self.z_flow = PropagateFlow(z_flow_type, in_features, num_transforms)   # :175-176
        self.r_flow = PropagateFlow(r_flow_type, in_features, num_transforms)
        self.z = 0

    _names = ("weight_mu", "weight_rho", "lambdal", "bias_mu", "bias_rho", "q0_mean", "q0_log_var",
              "r0_c", "r0_b1", "r0_b2")

    def _param_list(self):
        ps = [getattr(self, n) for n in self._names]
        ps += list(self.z_flow.parameters()) + list(self.r_flow.parameters())
        return ps

    def _check_flows(self):
        """'planar' (K3, fused/batched path), 'chain' (any 1-D flow kinds: lbbnn_flow_chain) or 'dense' (K4: RNVP /
        MNF-type coupling flows); a 1-D flow cannot be paired with a dense one in this build."""
        from .flows import VECTOR_KINDS
        zk, rk = self.z_flow.kind, self.r_flow.kind
        if zk == "Planar" and rk == "Planar":
            return "planar"
        if zk in VECTOR_KINDS and rk in VECTOR_KINDS:
            return "chain"
        if zk in ("RNVP", "MNF") and rk in ("RNVP", "MNF"):
            return "dense"
        raise NotImplementedError("bnn_amd: z_flow=%s with r_flow=%s: 1-D and dense flows cannot be mixed in one layer "
                                  "in this build" % (zk, rk))

    def _chain_flows(self, rng, eps_z, eps_z2, want_kl):
        """z draws + flows of a layer whose flows are 1-D chains: up to three lbbnn_flow_chain launches filling the
        same workspace slots as K3 (z_fwd, z_kl, scal[0..4]).  Row-wise restatement: only the kept row of the
        B-row forward draw is computed (for Radial the reference's norm runs over all B rows -- documented deviation)."""
        ws = self._workspace()
        L, I = self._layer_id, self.in_features
        zs = self.z_flow.chain_steps()
        ops.flow_chain(zs, I=I, q0_mean=self.q0_mean, q0_log_var=self.q0_log_var, eps=eps_z, rng=rng,
                       rng_stream=ops.STREAM_EPS_Z * 64 + L, z_out=ws.z_fwd, logdet=ws.scal[4:5])
        if want_kl:
            ops.flow_chain(zs, I=I, q0_mean=self.q0_mean, q0_log_var=self.q0_log_var, eps=eps_z2, rng=rng,
                           rng_stream=ops.STREAM_EPS_Z2 * 64 + L, z_out=ws.z_kl, logdet=ws.scal[0:1], log_q0=ws.scal[1:2])
            if getattr(ws, "chain_tmp", None) is None:
                ws.chain_tmp = torch.empty(I, dtype=torch.float32, device=self.q0_mean.device)
            ops.flow_chain(self.r_flow.chain_steps(), I=I, z_in=ws.z_kl, z_out=ws.chain_tmp, logdet=ws.scal[2:3],
                           z_last=ws.scal[3:4])

    def _masks(self, cfg, B):
        """Bernoulli(0.5) masks of the dense flows (flows2.py:209,234): from layer.noise or drawn on the device."""
        noise = self.noise or {}
        T = len(self.z_flow.transforms)
        dev, I = self.q0_mean.device, self.in_features

        Tr = len(self.r_flow.transforms)
        pool = []
        explicit = any(k in noise for k in ("zmask", "zmask2", "rmask"))
        # no explicit masks and in-kernel noise: the first launch of the flow kernels draws them from Philox
        # (lbbnn_dense_layer_t::draw_masks) -- the vectors are only allocated here; else ONE torch launch draws all of them
        in_kernel = (not explicit) and self._uses_rng(cfg) and _MASKS_IN_KERNEL

        def draw(n):
            if not pool:
                if self._mask_pool is not None:
                    pool.extend(self._mask_pool)
                    self._mask_pool = None
                elif in_kernel:
                    pool.extend(torch.empty(2 * T + Tr, I, device=dev).unbind(0))
                else:
                    pool.extend(torch.empty(2 * T + Tr, I, device=dev).bernoulli_(0.5).unbind(0))
            return [pool.pop() for _ in range(n)]
        zm = [m[-1] if m.dim() == 2 else m for m in noise["zmask"]] if "zmask" in noise else draw(T)
        out = {"zmask": zm}
        if cfg[1]:
            out["zmask2"] = [m.reshape(-1) for m in noise["zmask2"]] if "zmask2" in noise else draw(T)
            out["rmask"] = [m.reshape(-1) for m in noise["rmask"]] if "rmask" in noise else draw(Tr)
        out["_in_kernel"] = in_kernel
        return out

    as_written = False    # True: the forward's z comes from the B-ROW flow of LBBNN-GP-MF-MNF.py:194 as it is written
                          # (all B rows through z_flow, B-1 of them discarded) instead of the kept row alone -- the
                          # apples-to-apples leg against the CPU reference; no-grad forwards only

    def _sample_z_rows(self, R, rng, eps=None, masks=None):
        """The reference's sample_z arithmetic on R rows (LBBNN-GP-MF-MNF.py:183-186): (zs (R,I), logdet rows (R,), z0)."""
        L = self._layer_id
        z0 = ops.q0_rows(self.q0_mean, self.q0_log_var, R, eps=eps, rng=rng, rng_stream=ops.STREAM_EPS_Z * 64 + L)
        if self._check_flows() == "dense":
            descs, T, keep = self.z_flow.dense_descs(None, None)
            if masks is not None:
                masks = torch.stack([m.reshape(R, self.in_features).float() for m in masks])
            zs, ld, _ = ops.flow_dense_rows(descs, T, z0, masks=masks, rng=rng if masks is None else None,
                                            rng_stream=ops.STREAM_ROW_MASK * 64 + L, row_base=self.row_offset, keep=keep)
        else:
            zs, ld = ops.flow_chain_rows(self.z_flow.chain_steps(), z0)
        return zs, ld, z0

    def sample_z(self, batch_size=1):
        """LBBNN-GP-MF-MNF.py:182-187 as written: z0 (batch_size, I) = q0_mean + q0_std * eps is kept in ``self.z``, all
        rows go through ``z_flow`` and ``(zs[-1], logdet.squeeze())`` is returned -- logdet with the shape the flow kind
        gives it in the reference (RNVP: (batch_size,) -> squeezed; MNF type: 0-d; 1-D kinds: row-wise (batch_size,)).
        (The layer's own forward does not call this: it computes the kept row alone, fused with its other vector work.)"""
        st = ops.RngState.get(self.q0_mean.device)
        noise = self.noise or {}
        eps = noise.get("eps_z")
        if eps is not None and eps.numel() != batch_size * self.in_features:
            eps = None
        with torch.no_grad():
            zs, ld, z0 = self._sample_z_rows(batch_size, st.t, eps=eps, masks=noise.get("zmask") if eps is not None else None)
            st.advance(1)
        self.z = z0                                                   # :185
        logdet = ld.sum() if self.z_flow.kind == "MNF" else ld
        return zs[-1], logdet.squeeze()                               # :187

    def _prep_flows_only(self, rng):
        ws = self._workspace()
        family = self._check_flows()
        if family == "planar":
            ops.mnf_flow_planar(self.q0_mean, self.q0_log_var, self.z_flow.planar_params(), [], rng=rng,
                                layer_id=self._layer_id, z_fwd=ws.z_fwd, scal=ws.scal, want_kl=False)
        elif family == "chain":
            self._chain_flows(rng, None, None, False)
        else:
            from . import _lib
            dl, keep = (_lib.DenseLayer * 1)(), []
            self._dense_layer_desc(dl[0], (True, False, False), keep)
            _lib.check(_lib.lib().lbbnn_layers_dense_flows(dl, 1, rng.data_ptr() if rng is not None else None,
                                                           torch.cuda.current_stream(self.q0_mean.device).cuda_stream),
                       "lbbnn_layers_dense_flows")
            del keep

    def _needed_noise(self, cfg):
        return ["eps_z"] + (["eps_out"] if cfg[0] else []) + (["eps_z2", "eps_act"] if cfg[1] else [])

    def _fusable(self):
        # planar: K3 inside lbbnn_layers_operands; dense / chain: flows first, then flows_done = 1
        return not self.as_written

    def _fill_desc(self, d, cfg, kl_layer):
        ws, noise = super()._fill_desc(d, cfg, kl_layer)
        for name in ("q0_mean", "q0_log_var", "r0_c", "r0_b1", "r0_b2"):
            setattr(d, name, getattr(self, name).data_ptr())
        planar = self._check_flows() == "planar"
        for fd, flow in ((d.z_flow, self.z_flow), (d.r_flow, self.r_flow)):
            fd.T = len(flow.transforms) if planar else 0          # other flow kinds are run by the caller (flows_done)
            for t, tr in enumerate(flow.transforms if planar else ()):
                fd.u[t], fd.w[t], fd.b[t] = tr.u.data_ptr(), tr.w.data_ptr(), tr.bias.data_ptr()
        keep = []                                   # keep reshaped noise views alive until the launch
        eps_z = noise.get("eps_z")
        if eps_z is not None:
            eps_z = (eps_z[-1] if eps_z.dim() == 2 else eps_z).contiguous()
            keep.append(eps_z)
        eps_z2 = noise.get("eps_z2")
        if eps_z2 is not None:
            eps_z2 = eps_z2.reshape(-1).contiguous()
            keep.append(eps_z2)
        eps_act = noise.get("eps_act")
        d.eps_z = eps_z.data_ptr() if eps_z is not None else None
        d.eps_z2 = eps_z2.data_ptr() if eps_z2 is not None else None
        d.eps_act = eps_act.data_ptr() if eps_act is not None else None
        d.z_fwd, d.z_kl, d.scal = ws.z_fwd.data_ptr(), ws.z_kl.data_ptr(), ws.scal.data_ptr()
        d.act_mu, d.act_var = ws.act_mu.data_ptr(), ws.act_var.data_ptr()
        d.flows_done = 0 if self._check_flows() == "planar" else 1
        return keep

    def _dense_layer_desc(self, dl, cfg, keep):
        """Fill one lbbnn_dense_layer_t (K4 batched over layers); `keep` collects what must outlive the launch."""
        ws = self._workspace()
        want_kl = cfg[1]
        masks = self._masks(cfg, 0)
        self._last_masks = masks
        zd, Tz, k1 = self.z_flow.dense_descs(masks["zmask"], masks.get("zmask2"))
        rd, Tr, k2 = self.r_flow.dense_descs(None, masks.get("rmask")) if want_kl else (None, 0, None)
        if getattr(ws, "flow_work", None) is None:
            ws.flow_work = torch.empty(ops.flow_dense_workspace(self.in_features), dtype=torch.float32,
                                       device=self.q0_mean.device)
        noise = self.noise or {}
        eps_z, eps_z2 = noise.get("eps_z"), noise.get("eps_z2")
        if eps_z is not None:
            eps_z = (eps_z[-1] if eps_z.dim() == 2 else eps_z).contiguous()
        if eps_z2 is not None:
            eps_z2 = eps_z2.reshape(-1).contiguous()
        keep.extend([zd, rd, k1, k2, masks, eps_z, eps_z2])
        import ctypes as _ct
        from . import _lib
        dl.q0_mean, dl.q0_log_var = self.q0_mean.data_ptr(), self.q0_log_var.data_ptr()
        dl.zt = _ct.cast(zd, _ct.POINTER(_lib.DenseTransform))
        dl.rt = _ct.cast(rd, _ct.POINTER(_lib.DenseTransform)) if rd is not None else None
        dl.eps_fwd = eps_z.data_ptr() if eps_z is not None else None
        dl.eps_kl = eps_z2.data_ptr() if eps_z2 is not None else None
        dl.z_fwd, dl.z_kl, dl.scal, dl.work = ws.z_fwd.data_ptr(), ws.z_kl.data_ptr(), ws.scal.data_ptr(), ws.flow_work.data_ptr()
        dl.Tz, dl.Tr, dl.I, dl.want_kl, dl.layer_id = Tz, Tr, self.in_features, int(want_kl), self._layer_id
        dl.draw_masks = int(bool(masks.get("_in_kernel")))

    def _uses_rng(self, cfg):
        noise = self.noise or {}
        need = self._needed_noise(cfg)
        have = [k for k in need if k in noise]
        if have and len(have) != len(need):
            raise RuntimeError("bnn_amd: layer.noise must give all of %s or none (got %s)" % (need, have))
        return not have

    def _prep(self, cfg, rng, kl_layer=None, kl_total=None, accumulate=False, finalize=True):
        """x-independent kernels: K3 flows, K1 weight pass, K5 KL finalize (``finalize=False`` defers K5
        to ``_finalize`` so a network can keep it off the critical path)."""
        stochastic, want_kl, _ = cfg
        family = self._check_flows()
        ws = self._workspace()
        noise = self.noise or {}
        eps_z = noise.get("eps_z")
        if eps_z is not None and eps_z.dim() == 2:
            eps_z = eps_z[-1]                      # zs[-1]: only the last of the B rows is used (:187)
        eps_z2 = noise.get("eps_z2")
        if eps_z2 is not None:
            eps_z2 = eps_z2.reshape(-1)
        eps_z = None if eps_z is None else eps_z.contiguous()
        eps_z2 = None if eps_z2 is None else eps_z2.contiguous()
        if family == "planar":
            ops.mnf_flow_planar(self.q0_mean, self.q0_log_var, self.z_flow.planar_params(),
                                self.r_flow.planar_params(), eps_fwd=eps_z, eps_kl=eps_z2,
                                rng=rng, layer_id=self._layer_id, z_fwd=ws.z_fwd, z_kl=ws.z_kl, scal=ws.scal,
                                want_kl=want_kl)
        elif family == "chain":
            self._chain_flows(rng, eps_z, eps_z2, want_kl)
        else:
            from . import _lib
            pre = self._preflow
            if pre is not None and self._keep_dense and pre["cfg"][:2] == tuple(cfg[:2]):
                # z_fwd / z_kl / scal of this call are already in the workspace (one batched launch sequence for the net)
                self._last_masks, self._last_dense_save, self._last_flow_rng = pre["masks"], pre["save"], pre["rng"]
            else:
                dl, keep = (_lib.DenseLayer * 1)(), []
                self._dense_layer_desc(dl[0], cfg, keep)
                self._last_dense_save = None
                if self._keep_dense:
                    # training: the forward keeps every transform's input and the coupling MLPs' hidden activations for
                    # lbbnn_mnf_flow_dense_backward (a fresh buffer per call: it belongs to this call's autograd node)
                    self._last_dense_save = torch.empty(ops.flow_dense_save_size(self.in_features, dl[0].Tz, dl[0].Tr),
                                                        dtype=torch.float32, device=self.q0_mean.device)
                    dl[0].save = self._last_dense_save.data_ptr()
                _lib.check(_lib.lib().lbbnn_layers_dense_flows(dl, 1, rng.data_ptr() if rng is not None else None,
                                                               torch.cuda.current_stream(self.q0_mean.device).cuda_stream),
                           "lbbnn_layers_dense_flows")
                del keep
        if self.as_written:
            # B rows through z_flow, the last one kept (LBBNN-GP-MF-MNF.py:186-187): it replaces the kept-row result above
            B = int(self._cur_B)
            full_eps = noise.get("eps_z")
            if full_eps is not None and full_eps.numel() != B * self.in_features:
                raise RuntimeError("bnn_amd: as_written with explicit noise needs eps_z of shape (B,I)")
            zm = noise.get("zmask") if full_eps is not None else None
            zs, _, _ = self._sample_z_rows(B, rng, eps=full_eps, masks=zm)
            ws.z_fwd.copy_(zs[-1])
        ops.weight_pass(self.weight_mu, self.weight_rho, self.lambdal, z_fwd=ws.z_fwd,
                        z_kl=ws.z_kl if want_kl else None, r0_c=self.r0_c if want_kl else None,
                        bias_rho=self.bias_rho, priors=self.priors, e_w=ws.e_w,
                        var_w=ws.var_w if stochastic else None,
                        kl_rows=ws.kl_rows if want_kl else None,
                        act_mu=ws.act_mu if want_kl else None, act_var=ws.act_var if want_kl else None,
                        bias_var=ws.bias_var, split=self._split_now, e_scale=ws.e_scale, v_scale=ws.v_scale)
        if want_kl and finalize:
            self._finalize(rng, kl_layer, kl_total, accumulate)

    def _finalize(self, rng, kl_layer=None, kl_total=None, accumulate=False):
        ws = self._workspace()
        ops.kl_finalize(ws.kl_rows, self.bias_mu, self.bias_rho, priors=self.priors,
                        act_mu=ws.act_mu, act_var=ws.act_var, eps_act=(self.noise or {}).get("eps_act"),
                        r0_b1=self.r0_b1, r0_b2=self.r0_b2, scal=ws.scal, rng=rng,
                        layer_id=self._layer_id, kl_layer=kl_layer, kl_out=kl_total, accumulate=accumulate)

    def _noise_for_backward(self, saved, B, need_out=True):
        masks = saved.get("masks") or {}
        if saved.get("noise"):
            n = dict(saved["noise"])
            if n["eps_z"].dim() == 2:
                n["eps_z"] = n["eps_z"][-1]
            if "eps_z2" in n:
                n["eps_z2"] = n["eps_z2"].reshape(-1)
            n.update(masks)
            return n
        r, L, I, O = saved["rng"], self._layer_id, self.in_features, self.out_features
        n = {"eps_z": ops.philox_normal(r, ops.STREAM_EPS_Z * 64 + L, 0, I),
             "eps_z2": ops.philox_normal(r, ops.STREAM_EPS_Z2 * 64 + L, 0, I),
             "eps_act": ops.philox_normal(r, ops.STREAM_EPS_ACT * 64 + L, 0, O)}
        if need_out:
            n["eps_out"] = ops.philox_normal(r, ops.STREAM_EPS_OUT * 64 + L, B, O, self.row_offset)
        n.update(masks)
        return n

    def _unpack_params(self, ps):
        n = len(self._names)
        P = dict(zip(self._names, ps[:n]))
        rest = list(ps[n:])
        specs = []
        for flow in (self.z_flow, self.r_flow):
            if flow.kind == "Planar":
                T = len(flow.transforms)
                specs.append(("Planar", [tuple(rest[3 * t:3 * t + 3]) for t in range(T)]))
                rest = rest[3 * T:]
            else:
                trs = []
                for tr in flow.transforms:
                    names = [k for k, _ in tr.named_parameters()]
                    trs.append(dict(zip(names, rest[:len(names)])))
                    rest = rest[len(names):]
                specs.append((flow.kind, trs))
        return P, specs

    _vec_names = _names[3:]

    def _planar_params_from(self, params):
        """(u, w, bias) triples of the z and r flows out of the saved parameter list (order of _param_list)."""
        rest = list(params[len(self._names):])
        Tz, Tr = len(self.z_flow.transforms), len(self.r_flow.transforms)
        zp = [tuple(rest[3 * t:3 * t + 3]) for t in range(Tz)]
        rp = [tuple(rest[3 * (Tz + t):3 * (Tz + t) + 3]) for t in range(Tr)]
        return zp, rp

    def _vector_graph(self, vs, cfg, noise, act_mu=None, act_var=None):
        P, specs = self._unpack_params([None, None, None] + list(vs))
        return _grad.mnf_vector_graph(P, specs[0], specs[1], noise, act_mu, act_var, stochastic=cfg[0],
                                      want_kl=cfg[1], priors=self.priors)


class _NetworkBase(nn.Module):
    """3-layer MLP of Bayesian layers: ReLU, ReLU, log_softmax (LBBNN-GP-MF-LRT.py:206-214).

    Without autograd the forward is ONE stream and 5 launches (``_forward_streams``): the x-independent kernels of
    all layers batched into one launch per kind (flows, weight pass), and the three GEMMs with ReLU / log_softmax fused
    into their epilogues; the KL finalize of all layers rides in the first GEMM's launch and the RNG offset is advanced
    by the weight-pass launch.  The three layers
    share one RNG offset (their Philox streams differ by layer id).  No host synchronisation: HIP-graph capturable.
    With autograd each layer goes through ``_BayesLinearFn`` (HIP forward + HIP backward).
    """
    _kl_total = None
    _train_kl_total = None        # training forward: the network KL as summed on the device by the KL finalize (else None)
    _pre_shared = None

    def _layers(self):
        return [self.l1, self.l2, self.l3]

    def _number_layers(self):
        """Philox stream ids 0..n-1 inside a network (a stand-alone layer takes the next id of a process-wide counter):
        the noise a network draws does not depend on what else the process constructed before it -- every rank of a
        data-parallel job, and a re-run of the same script with one more model in it, see the same streams."""
        for i, l in enumerate(self._layers()):
            l._layer_id = i
        # the batched weight pass of a network is ONE launch: the row-scaled fp16 format needs every row of every layer in
        # one register batch of the vector row kernel there (<= 1280 weights per row, rows of whole float4s: weight_pass.hip
        # launch_weight_pass), so a network with a wider layer, or one whose row length is not a multiple of 4, keeps bf16x3
        # for the layers that qualify for it (found by tools/net_train_fuzz.py: 17-wide head under an fp16 first layer)
        ok = all(ops.operand_ld(l.in_features) <= 1280 and l.in_features % 4 == 0 for l in self._layers())
        for l in self._layers():
            l._f16s_net_ok = ok
        self._plane_cache = {}

    precision = None

    def set_precision(self, name):
        """GEMM arithmetic of THIS network (ops.PRECISIONS; None = follow the process-wide default of ops.set_precision)."""
        if name is not None and name not in ops.PRECISIONS:
            raise ValueError("precision must be one of %s or None" % (ops.PRECISIONS,))
        self.precision = name
        for l in self._layers():
            l.precision = name
        return self

    def _planes(self, key, B, width, dev, plan=None):
        """A (B, plane_ld(width)) zero-initialised fp32-sized buffer for fp16 hi | lo planes: the tail slots past `width` are
        never written afterwards, so they stay zero (the consuming GEMM reads whole 128-B lines)."""
        store = plan if plan is not None else self._plane_cache
        k = ("planes", key, B, width, str(dev))
        buf = store.get(k)
        if buf is None:
            buf = store[k] = torch.zeros((B, ops.plane_ld(width)), dtype=torch.float32, device=dev)
        return buf

    def forward(self, x, sample=False):
        x = x.view(-1, self.dims[0])                                  # …LRT.py:207
        layers = self._layers()
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        self._train_kl_total = None
        if needs_grad or not x.is_cuda or not all(l._fusable() for l in layers):
            self._kl_total = None
            shared = needs_grad and x.is_cuda
            if shared:
                # as in the fused no-grad path the three layers share ONE RNG offset (their Philox streams differ by
                # layer id), advanced once after the last layer
                self._predraw_masks([l for l in layers if l._mnf and l._check_flows() == "dense"])
                self._preflow_dense(layers, sample)
                self._preprep_all(layers, sample, x)
            fused_lsm = False
            try:
                for i, l in enumerate(layers):
                    l._advance_rng = not shared
                    # the head's F.log_softmax (…LRT.py:210) runs in its GEMM's epilogue in the training forward too: the layer's
                    # autograd node then returns log-probabilities and its backward starts with lbbnn_log_softmax_backward
                    l._lsm_now = bool(shared and i == len(layers) - 1 and l.out_features <= 16 and (l.training or sample))
                    fused_lsm = l._lsm_now
                    x = l.forward(x, sample, _relu=(i < 2))           # F.relu fused into the GEMM epilogue
            finally:
                for l in layers:
                    l._advance_rng = True
                    l._lsm_now = False
            if shared:
                sh = self._pre_shared
                if not (sh is not None and sh.get("advanced")):
                    ops.RngState.get(x.device).advance(1)
                self._train_kl_total = sh["kl_total"] if (sh is not None and sh.get("hosted") and sh.get("kl_total") is not None) else None
                self._pre_shared = None
            return x if fused_lsm else F.log_softmax(x, dim=1)       # …LRT.py:210
        return self._forward_streams(x.float(), sample)

    @staticmethod
    def _predraw_masks(dense_layers):
        """One Bernoulli(0.5) launch for every mask the dense flows of these layers need in this forward."""
        need = [l for l in dense_layers if not (l.noise and "zmask" in l.noise) and l._mask_pool is None]
        if not need:
            return
        dev = need[0].q0_mean.device
        rows = [2 * len(l.z_flow.transforms) + len(l.r_flow.transforms) for l in need]
        width = max(l.in_features for l in need)
        allm = torch.empty(sum(rows), width, device=dev)
        if not (_MASKS_IN_KERNEL and all(l._uses_rng((True, l.training, False)) for l in need)):
            allm.bernoulli_(0.5)                 # (else the flow kernels fill them: lbbnn_dense_layer_t::draw_masks)
        r0 = 0
        for l, r in zip(need, rows):
            l._mask_pool = [allm[r0 + k, :l.in_features] for k in range(r)]
            r0 += r

    def _preflow_dense(self, layers, sample):
        """Training forward of a net whose layers all use dense (RNVP / MNF-type) flows: the flows depend on parameters
        only, so all layers' z draws + flows run here in ONE batched launch sequence (lbbnn_layers_dense_flows, 10 launches
        for the net instead of 10 per layer), each keeping its intermediates for lbbnn_mnf_flow_dense_backward; the
        per-layer autograd forward then starts at the weight pass."""
        from . import _lib
        if not _DENSE_HIP_BWD or not all(l._mnf and l._check_flows() == "dense" for l in layers):
            return
        cfgs = [(bool(l.training or sample), bool(l.training), False) for l in layers]
        if len({(len(l.z_flow.transforms), len(l.r_flow.transforms), c[1]) for l, c in zip(layers, cfgs)}) != 1:
            return
        if len(layers) > 4:
            return
        dev = layers[0].q0_mean.device
        rng = ops.RngState.get(dev).t if any(l._uses_rng(c) for l, c in zip(layers, cfgs)) else None
        snap = rng.clone() if rng is not None else None
        self._predraw_masks(layers)
        dls, keep = (_lib.DenseLayer * len(layers))(), []
        for k, (l, c) in enumerate(zip(layers, cfgs)):
            ws = l._workspace()
            for name in ("z_fwd", "z_kl", "scal"):
                setattr(ws, name, torch.empty_like(getattr(ws, name)))
            l._dense_layer_desc(dls[k], c, keep)
            save = torch.empty(ops.flow_dense_save_size(l.in_features, dls[k].Tz, dls[k].Tr), dtype=torch.float32, device=dev)
            dls[k].save = save.data_ptr()
            l._preflow = {"cfg": c, "masks": l._last_masks, "save": save, "rng": snap}
        _lib.check(_lib.lib().lbbnn_layers_dense_flows(dls, len(layers), rng.data_ptr() if rng is not None else None,
                                                       torch.cuda.current_stream(dev).cuda_stream), "lbbnn_layers_dense_flows")
        del keep

    def _preprep_all(self, layers, sample, x):
        """Training forward: K3 (planar flows) and K1 (weight pass) depend on parameters only, so they run here for ALL
        layers -- one launch per kind, lbbnn_layers_operands -- with fresh per-call output vectors; the per-layer autograd
        forwards then consist of the GEMM (carrying the layer's KL tail) alone."""
        from . import _lib
        fams = [(l._check_flows() if l._mnf else "lrt") for l in layers]
        if "chain" in fams or len(layers) > 4:
            return
        if any(f == "dense" and l._preflow is None for f, l in zip(fams, layers)):
            return                                   # dense flows not run yet (torch-backward mode): per-layer path
        cfgs = [(bool(l.training or sample), bool(l.training), False) for l in layers]
        dev = x.device
        xin = x if x.dtype == torch.float32 else x.float()
        rng = ops.RngState.get(dev).t if any(l._uses_rng(c) for l, c in zip(layers, cfgs)) else None
        descs, keep, kls = (_lib.LayerDesc * len(layers))(), [], []
        for i, (l, c, f) in enumerate(zip(layers, cfgs, fams)):
            ws = l._workspace()
            if l._mnf:
                for name in (("act_mu", "act_var") if f == "dense" else ("z_fwd", "z_kl", "scal", "act_mu", "act_var")):
                    setattr(ws, name, torch.empty_like(getattr(ws, name)))
            l._split_now = l._split(xin if i == 0 else None, c) if (i == 0 or layers[i - 1].out_features % 4 == 0) else 0
            kl = torch.empty((), dtype=torch.float32, device=dev) if c[1] else None
            kls.append(kl)
            keep.append(l._fill_desc(descs[i], c, kl))
        # one Philox snapshot per step for every backward re-draw: the dense pre-flow took it already if it ran, else the
        # K1 launch writes it (advance 0: the shared offset is advanced once, after the last layer)
        snap = None
        if rng is not None:
            pf = layers[0]._preflow
            snap = pf["rng"] if (pf is not None and pf.get("rng") is not None) else None
        k1_snap = None
        if rng is not None and snap is None:
            snap = k1_snap = torch.empty(4, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().lbbnn_layers_operands_snap(descs, len(layers), rng.data_ptr() if rng is not None else None,
                                                         k1_snap.data_ptr() if k1_snap is not None else
                                                         (rng.data_ptr() + 16 if rng is not None else None), 0,
                                                         torch.cuda.current_stream(dev).cuda_stream), "lbbnn_layers_operands_snap")
        all_kl = all(c[1] for c in cfgs)
        kl_total = torch.empty((), dtype=torch.float32, device=dev) if all_kl else None
        shared = {"hosted": False, "keep": keep, "kl_total": kl_total, "advanced": False,
                  # every layer draws in-kernel from the shared state and the step has ONE snapshot: the advance may ride early
                  "all_snap": bool(snap is not None and rng is not None and all(l._uses_rng(c) for l, c in zip(layers, cfgs))),
                  "fin_all": (descs, len(layers), rng.data_ptr() if rng is not None else None,
                              kl_total.data_ptr()) if all_kl else None}
        self._pre_shared = shared
        for i, (l, c) in enumerate(zip(layers, cfgs)):
            l._preprep = {"cfg": c, "split": l._split_now, "kl": kls[i], "first": i == 0, "shared": shared, "snap": snap}

    def _forward_streams(self, x, sample):
        """Fused no-grad forward: ONE stream, 2 + 3 launches.

        lbbnn_layers_operands_snap runs the x-independent kernels the GEMMs need for all layers (flows, weight pass: one
        launch per kind); the weight-pass launch also copies the Philox state for the rest of this forward and advances
        the live offset.  Then the three GEMMs run back to back (ReLU / log_softmax in their epilogues); the first one
        carries every layer's KL tail + the network total as one extra workgroup (lbbnn_lrt_gemm_finalize) -- the KL
        depends on parameters only, so it needs neither a launch of its own nor a place on the critical path.
        (A two-stream schedule was measured first: every cross-stream dependency cost 13-15 us on
        the critical path and the side-stream kernels were starved by the GEMM -- see DESIGN.md.)
        """
        from . import _lib
        layers = self._layers()
        dev = x.device
        n = len(layers)
        cfgs = [(bool(l.training or sample), bool(l.training), i < n - 1) for i, l in enumerate(layers)]
        st = None
        if any(l._uses_rng(c) for l, c in zip(layers, cfgs)):
            st = ops.RngState.get(dev)
        rng = st.t if st is not None else None
        want_kl = any(c[1] for c in cfgs)
        # graphs.LaunchPlan records this forward into buffers of its own (kls, the layers' outputs) and keeps `keep` alive
        plan = getattr(self, "_plan_rec", None)
        if plan is not None and want_kl:
            kls = plan.setdefault("kls", torch.empty(n + 1, dtype=torch.float32, device=dev))
        else:
            kls = torch.empty(n + 1, dtype=torch.float32, device=dev) if want_kl else None
        descs = (_lib.LayerDesc * n)()
        keep = []
        if plan is not None:
            plan["keep"] = keep
        for i, (l, c) in enumerate(zip(layers, cfgs)):
            # activations produced by our own GEMMs are dense 16-B aligned rows; the network input is checked
            l._split_now = l._split(x if i == 0 else None, c) if (i == 0 or layers[i - 1].out_features % 4 == 0) else 0
            keep.append(l._fill_desc(descs[i], c, kls[i] if c[1] else None))
        stream = torch.cuda.current_stream(dev).cuda_stream
        # flows that are not planar run first (K4 batched over the layers; 1-D chains per layer), then lbbnn_layers_operands
        # runs K1 only for those layers (flows_done)
        dense = [(l, c) for l, c in zip(layers, cfgs) if l._mnf and l._check_flows() == "dense"]
        # The r flow and the flows' scalars feed the KL only (LBBNN-GP-MF-MNF.py:208-235); the activations need z_k alone (:190-200).
        # With KL wanted and a second GEMM to carry the finalize, that part of the dense flows (2 of the 4 stage launches + the
        # finish for the reference's T = 2: ~42 us) runs on a side stream beside the weight pass and the first GEMM, and the KL
        # finalize + the RNG advance ride in the SECOND GEMM's launch instead of the first.  Same kernels, same arguments, same
        # results bit for bit (test_dense_flows_deferred_r_part_bitwise); LBBNN_DENSE_DEFER=0 keeps everything on one stream.
        # Measured (RNVP, headline sizes, graph replay): 0.237 -> 0.226 ms; the join costs the second GEMM ~11 us of cross-queue
        # latency and the side branch runs 2.5 x slower beside the first GEMM than alone, or the gain would be the whole 42 us.
        defer_ev = fork_ev = deferred = None
        if dense:
            self._predraw_masks([l for l, _ in dense])
            same = len({(len(l.z_flow.transforms), len(l.r_flow.transforms), c[1]) for l, c in dense}) == 1
            groups = [dense] if same else [[lc] for lc in dense]
            # (only while a HIP graph is being captured: launched from Python the two event calls and the stream switch cost the
            # host more than the overlap returns -- RNVP forward 0.405 -> 0.454 ms eager, 0.237 -> 0.226 ms replayed)
            # (not where the second GEMM folds the <= 16-class head into its epilogue -- a three-layer row-scaled fp16 network:
            # the launch that carries the finalize does not fold, and the unfolded head is another arithmetic (fp32 skinny
            # kernel), so a captured forward would no longer be the eager one bit for bit; tools/forward_replay_fuzz.py)
            folds_at_1 = (_HEAD_FOLD and n == 3 and layers[1]._split_now >= 2 and layers[2].out_features <= 16
                          and layers[2]._split_now == 0 and cfgs[2][0] and cfgs[1][2] and layers[1].out_features % 4 == 0)
            defer = (_DENSE_DEFER and (torch.cuda.is_current_stream_capturing() or _DENSE_DEFER == "always") and n >= 2
                     and all(c[1] for _, c in dense) and layers[1].out_features > 16 and not folds_at_1
                     and any(len(l.r_flow.transforms) > 0 for l, _ in dense))
            batches = []
            for grp in groups:
                dls = (_lib.DenseLayer * len(grp))()
                for k, (l, c) in enumerate(grp):
                    l._dense_layer_desc(dls[k], c, keep)
                batches.append((dls, len(grp)))
                _lib.check(_lib.lib().lbbnn_layers_dense_flows_phase(dls, len(grp), rng.data_ptr() if rng is not None else None,
                                                                    1 if defer else 0, stream), "lbbnn_layers_dense_flows_phase")
            if defer:
                fork_ev = torch.cuda.Event()
                fork_ev.record(torch.cuda.current_stream(dev))
                deferred = batches
        for l, c in zip(layers, cfgs):
            if l._mnf and l._check_flows() == "chain":
                noise = l.noise or {}
                eps_z, eps_z2 = noise.get("eps_z"), noise.get("eps_z2")
                if eps_z is not None:
                    eps_z = (eps_z[-1] if eps_z.dim() == 2 else eps_z).contiguous()
                if eps_z2 is not None:
                    eps_z2 = eps_z2.reshape(-1).contiguous()
                keep.extend([eps_z, eps_z2])
                l._chain_flows(rng, eps_z, eps_z2, c[1])
        # K1 of every layer, the planar flows computed inside its own workgroups (no K3 launch: weight_pass.hip); the launch
        # snapshots the Philox state for the rest of this forward.  The live offset is advanced by the extra workgroup of
        # the first GEMM's launch (lbbnn_lrt_gemm_finalize_adv): every workgroup of THIS launch reads it.
        snap = st.t[2:4] if st is not None else None
        # A first layer in the row-scaled fp16 format reads its x as fp16 hi | lo planes: the format pass over the network
        # input rides in the SAME launch as the planar flows (lbbnn_layers_operands_x: extra workgroups on the CUs the two
        # flow workgroups per layer leave idle), so the first GEMM spends no VALU on the split and no launch is added.
        B, x_planes = x.shape[0], False
        if (layers[0]._split_now >= 2 and _F16_FIRST_PLANES and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
                and B > 0):
            xp = self._planes("in", B, layers[0].in_features, dev, plan)
            _lib.check(_lib.lib().lbbnn_layers_operands_x(descs, n, rng.data_ptr() if rng is not None else None,
                                                          snap.data_ptr() if snap is not None else None, 0,
                                                          x.data_ptr(), x.stride(0), xp.data_ptr(), xp.stride(0), B,
                                                          layers[0].in_features, stream), "lbbnn_layers_operands_x")
            x, x_planes = xp, True
        else:
            _lib.check(_lib.lib().lbbnn_layers_operands_snap(descs, n, rng.data_ptr() if rng is not None else None,
                                                             snap.data_ptr() if snap is not None else None, 0, stream),
                       "lbbnn_layers_operands_snap")
        if deferred is not None:
            # enqueued AFTER the weight pass: a captured graph keeps the first-recorded successor of a fork on the parent's queue,
            # and a hop to another queue costs ~11 us -- it must be the side branch that pays it, not the weight pass
            side = _side_stream(dev)
            side.wait_event(fork_ev)
            for dls, cnt in deferred:
                _lib.check(_lib.lib().lbbnn_layers_dense_flows_phase(dls, cnt, rng.data_ptr() if rng is not None else None,
                                                                    2, side.cuda_stream), "lbbnn_layers_dense_flows_phase")
            defer_ev = torch.cuda.Event()
            defer_ev.record(side)
        all_kl = want_kl and all(c[1] for c in cfgs)
        fin_at = 1 if defer_ev is not None else 0
        head_done = False
        for i, (l, c) in enumerate(zip(layers, cfgs)):
            # the KL finalize of all layers (parameters only) rides in the first GEMM's launch as one extra workgroup (in the
            # second's when part of the dense flows was deferred: it waits for the side stream first)
            fin = None
            if i == fin_at and defer_ev is not None:
                torch.cuda.current_stream(dev).wait_event(defer_ev)
            if i == fin_at and (want_kl or st is not None):
                fin = (descs if want_kl else None, n if want_kl else 0, snap.data_ptr() if snap is not None else None,
                       kls[n:].data_ptr() if all_kl else None, rng.data_ptr() if rng is not None else None,
                       1 if rng is not None else 0)
            # row-scaled fp16 layers hand their activations on as fp16 hi | lo PLANES (written by the GEMM epilogue, read by
            # the next GEMM's LDS-DMA as they lie): no fp32 copy of a hidden activation is stored in this no-grad forward
            fmt = l._split_now
            if head_done:
                break                                     # the last layer's GEMM ran inside the previous launch (head fold)
            give_planes = (fmt >= 2 and i + 1 < n and layers[i + 1]._split_now >= 2 and l.out_features % 8 == 0)
            # HEAD FOLD: a row-scaled fp16 layer followed by the <= 16-class head computes the head's two moment products in
            # its own epilogue (lbbnn_gemm_desc_t::head_*): the hidden activation is never stored, the head has no GEMM launch
            head = None
            nxt = layers[i + 1] if i + 1 < n else None
            if (_HEAD_FOLD and fmt >= 2 and nxt is not None and i + 2 == n and nxt.out_features <= 16 and nxt._split_now == 0
                    and cfgs[i + 1][0] and c[2] and l.out_features % 4 == 0 and fin is None):
                wsn = nxt._workspace()
                hout = (plan.setdefault("out%d" % (i + 1), torch.empty(B, nxt.out_features, dtype=torch.float32, device=dev))
                        if plan is not None else torch.empty(B, nxt.out_features, dtype=torch.float32, device=dev))
                key = ("slab", i, B, str(dev))
                store = plan if plan is not None else self._plane_cache
                slab = store.get(key)
                if slab is None:
                    slab = store[key] = torch.empty(ops.head_slab_floats(B, l.out_features), dtype=torch.float32, device=dev)
                head = {"e_w": wsn.e_w, "var_w": wsn.var_w, "bias_mean": nxt.bias_mu, "bias_var": wsn.bias_var,
                        "eps": (nxt.noise or {}).get("eps_out"), "rng_stream": ops.STREAM_EPS_OUT * 64 + nxt._layer_id,
                        "out": hout, "slab": slab, "log_softmax": True}
            obuf = None
            if plan is not None and not give_planes and head is None:
                obuf = plan.setdefault("out%d" % i, torch.empty(B, l.out_features, dtype=torch.float32, device=dev))
            pbuf = self._planes(i, B, l.out_features, dev, plan) if give_planes else None
            y = l._gemm(x, c, snap, log_softmax=(i == n - 1 and l.out_features <= 16), finalize=fin, out=obuf,
                        x_planes=x_planes, out_planes=pbuf, want_out=(not give_planes and head is None), head=head)
            if head is not None:
                x, x_planes, head_done = head["out"], False, True
            else:
                x, x_planes = (pbuf, True) if give_planes else (y, False)
        if layers[-1].out_features > 16:
            x = F.log_softmax(x, dim=1)
        for i, (l, c) in enumerate(zip(layers, cfgs)):
            l.kl = kls[i] if c[1] else 0
        self._kl_total = kls[n] if (want_kl and all(c[1] for c in cfgs)) else None
        return x

    def kl(self):
        if self._kl_total is not None:
            return self._kl_total                                     # summed on the device by K5
        if self._train_kl_total is not None and all(torch.is_tensor(l.kl) and l.kl.grad_fn is not None for l in self._layers()):
            from .losses import _SumKLFn                              # ... also in the training forward: no add kernels
            return _SumKLFn.apply(self._train_kl_total, [getattr(l, "_v1_slot", None) for l in self._layers()],
                                  *[l.kl for l in self._layers()])
        return self.l1.kl + self.l2.kl + self.l3.kl                   # …LRT.py:213-214

    def set_row_offset(self, off: int):
        for l in self._layers():
            l.row_offset = int(off)


class LRTBayesianNetwork(_NetworkBase):
    """BayesianNetwork() of LBBNN-GP-MF-LRT.py:199-214 (reference dims 784-400-600-10; ``dims=`` added)."""

    def __init__(self, dims=(28 * 28, 400, 600, 10), *, priors=None):
        super().__init__()
        self.dims = tuple(dims)
        self.l1 = LRTBayesianLinear(dims[0], dims[1], priors=priors)
        self.l2 = LRTBayesianLinear(dims[1], dims[2], priors=priors)
        self.l3 = LRTBayesianLinear(dims[2], dims[3], priors=priors)
        self._number_layers()


class MNFBayesianNetwork(_NetworkBase):
    """BayesianNetwork() of LBBNN-GP-MF-MNF.py:244-260 (num_transforms=2 there)."""

    def __init__(self, dims=(28 * 28, 400, 600, 10), num_transforms=2, *, z_flow_type="RNVP",
                 r_flow_type="RNVP", priors=None):
        super().__init__()
        self.dims = tuple(dims)
        kw = dict(z_flow_type=z_flow_type, r_flow_type=r_flow_type, priors=priors)
        self.l1 = MNFBayesianLinear(dims[0], dims[1], num_transforms, **kw)
        self.l2 = MNFBayesianLinear(dims[1], dims[2], num_transforms, **kw)
        self.l3 = MNFBayesianLinear(dims[2], dims[3], num_transforms, **kw)
        self._number_layers()
