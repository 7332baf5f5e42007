"""The scalar head of a training step as fused HIP launches (include/lbbnn.h: lbbnn_elbo_loss*).

``elbo_loss(log_probs, target, kl, num_batches)`` is ``F.nll_loss(log_probs, target, reduction='sum') + kl / num_batches`` --
what the reference's ``train()`` writes out (LBBNN-GP-MF-MNF.py:268-271; ...LRT.py:222-225) -- as ONE forward launch and ONE
backward launch instead of torch's nll_loss / mul / add kernels (13 + 4 + 4 us forward, 6 us + fills backward for a
4096 x 10 batch: the reduce kernel of nll_loss runs on a single workgroup).  The reference's own spelling keeps working; this
is the faster one for callers that want it (``bnn_amd.parallel.DataParallelELBO.loss`` uses it on HIP tensors).
"""
import torch
import torch.nn.functional as F

import os as _os

from . import _lib

_FUSE_LSM_BWD = _os.environ.get("LBBNN_FUSE_LSM_BWD", "1") != "0"     # A/B knob


# Hand-over from the loss's backward to the head layer's backward: when the log-probabilities came out of a layer's fused
# log_softmax, the loss backward forms the gradient with respect to the LOGITS in its own launch
# (lbbnn_elbo_loss_backward_logits) and leaves it here under the log-probabilities' address; the layer's backward takes it if
# the gradient it receives is the very tensor the loss returned, instead of launching lbbnn_log_softmax_backward.
_LOGITS_GRAD = {}


class _ElboLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, target, kl, scale):
        B, C = log_probs.shape
        _LOGITS_GRAD.clear()                       # entries live for one backward pass only
        lp = log_probs if log_probs.stride(1) == 1 else log_probs.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=lp.device)
        stream = torch.cuda.current_stream(lp.device).cuda_stream
        _lib.check(_lib.lib().lbbnn_elbo_loss(lp.data_ptr(), lp.stride(0), target.data_ptr(), B, C,
                                              kl.data_ptr() if kl is not None else None, float(scale), loss.data_ptr(), stream),
                   "lbbnn_elbo_loss")
        ctx.save_for_backward(target, lp)
        ctx.meta = (B, C, float(scale), kl is not None)
        return loss

    @staticmethod
    def backward(ctx, g):
        target, lp = ctx.saved_tensors
        B, C, scale, has_kl = ctx.meta
        g = g.contiguous()
        g_logp = torch.empty((B, C), dtype=torch.float32, device=g.device)
        g_kl = torch.empty((), dtype=torch.float32, device=g.device) if has_kl else None
        stream = torch.cuda.current_stream(g.device).cuda_stream
        if _FUSE_LSM_BWD and C <= 64:
            g_logits = torch.empty((B, C), dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().lbbnn_elbo_loss_backward_logits(g.data_ptr(), target.data_ptr(), lp.data_ptr(), lp.stride(0), B, C,
                                                                  scale, g_logp.data_ptr(), g_logits.data_ptr(),
                                                                  g_kl.data_ptr() if has_kl else None, stream),
                       "lbbnn_elbo_loss_backward_logits")
            # the entry HOLDS lp and g_logp: while it exists their storage cannot be freed and handed to another tensor, so
            # "same address" in the head's backward means "same tensor" (an address alone could be a reused block)
            _LOGITS_GRAD[lp.data_ptr()] = (g_logp.data_ptr(), g_logits, lp, g_logp)
        else:
            _lib.check(_lib.lib().lbbnn_elbo_loss_backward(g.data_ptr(), target.data_ptr(), B, C, scale, g_logp.data_ptr(),
                                                           g_kl.data_ptr() if has_kl else None, stream), "lbbnn_elbo_loss_backward")
        return g_logp, None, g_kl, None


def elbo_loss(log_probs, target, kl=None, num_batches=1.0):
    """nll_loss(log_probs, target, reduction='sum') + kl / num_batches.  HIP tensors: the fused launches; anything else (the
    CPU host-logic tests of bnn_amd.parallel run on oracle tensors): the same expression in torch ops."""
    if (log_probs.is_cuda and log_probs.dtype == torch.float32 and log_probs.dim() == 2 and target.dtype == torch.int64
            and target.is_contiguous() and (kl is None or (torch.is_tensor(kl) and kl.is_cuda and kl.dtype == torch.float32
                                                           and kl.numel() == 1))):
        return _ElboLossFn.apply(log_probs, target, kl if kl is None else kl.reshape(()), 1.0 / float(num_batches))
    nll = F.nll_loss(log_probs, target, reduction="sum")
    return nll if kl is None else nll + kl / num_batches


class _SumKLFn(torch.autograd.Function):
    """kl_total of a network whose layers' KL tails ran in ONE finalize (kl_piggy.h): the total was added on the device in
    layer order by that launch; this node only tells autograd that it is the sum of the per-layer values."""

    @staticmethod
    def forward(ctx, total, slots, *kls):
        ctx.n = len(kls)
        ctx.slots = slots
        return total.detach().view_as(total)

    @staticmethod
    def backward(ctx, g):
        # d loss / d kl is the same number for every layer and is known HERE first: the layers' V1 kernels
        # (lbbnn_mnf_aux_backward: by-products of the forward + this number) run as one launch, each layer's backward
        # picks its result up from its slot (layers._BayesLinearFn)
        slots = [s for s in (ctx.slots or []) if s is not None and "out" not in s and "in" in s]
        if slots and g.is_cuda and g.dtype == torch.float32 and g.numel() == 1:
            from . import ops
            gk = g.contiguous()
            for s, o in zip(slots, ops.mnf_aux_backward_batch([s["in"] for s in slots], gk)):
                s["out"] = o + (gk,)
                s.pop("in")
            g = gk
        ctx.slots = None
        return (None, None) + (g,) * ctx.n
