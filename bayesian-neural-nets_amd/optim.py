"""``Adam`` with ``torch.optim.Adam``'s constructor and update rule (the optimizer of the reference's training scripts:
``optim.Adam(net.parameters(), lr=...)``, LBBNN-GP-MF-LRT.py:358, LBBNN-GP-MF-MNF.py:421; per-parameter groups in
LBBNN-GP-MF.py:520-554), executed as ONE multi-tensor HIP launch per parameter group (``lbbnn_adam_step``) instead of
torch's ~115 small kernels for the 66 parameter tensors of the headline net.  The step counter lives on the device, so
``step()`` is HIP-graph capturable as it is (no ``capturable=`` switch needed).

Not supported (raise): ``amsgrad``, ``maximize``, sparse gradients, non-fp32 or CPU parameters.
"""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *,
                 maximize=False, capturable=True):
        if amsgrad or maximize:
            raise NotImplementedError("bnn_amd.optim.Adam: amsgrad / maximize are not implemented")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("bnn_amd.optim.Adam: invalid hyper-parameter")
        # (the extra keys are torch.optim.Adam's own group keys at their defaults: a state_dict of this optimizer then loads
        # into torch.optim.Adam and back)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))

    def _group_state(self, group):
        st = self.state
        for p in group["params"]:
            if p not in st or "exp_avg" not in st[p]:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("bnn_amd.optim.Adam needs contiguous float32 parameters on a HIP device")
                st[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if "step_dev" not in group:
            ref = group["params"][0]
            group["step_dev"] = torch.zeros(1, dtype=torch.float32, device=ref.device)
        return group["step_dev"]

    # torch.optim.Adam keeps one ``state[p]["step"]`` per parameter; this optimizer keeps ONE device-side counter per
    # parameter group (``group["step_dev"]``: every parameter of a group is updated in the same launch, so their counts
    # cannot differ).  state_dict() / load_state_dict() translate between the two, so that checkpoints interchange with
    # torch.optim.Adam and bias correction continues where it stopped.
    def state_dict(self):
        sd = super().state_dict()
        for gi, group in enumerate(self.param_groups):
            step = group.get("step_dev")
            sg = sd["param_groups"][gi]
            sg.pop("step_dev", None)
            if step is not None:
                count = float(step.detach().reshape(-1)[0])          # one device -> host read per group (checkpoint time only)
                for idx in sg["params"]:
                    if idx in sd["state"]:
                        # a tensor of its OWN per parameter, on the CPU, as torch.optim.Adam keeps it when capturable=False: its
                        # foreach path adds 1 to every listed step tensor, so a tensor shared by two parameters would count double
                        sd["state"][idx] = dict(sd["state"][idx], step=torch.tensor(count, dtype=torch.float32))
        return sd

    def load_state_dict(self, state_dict):
        steps = {}
        for gi, sg in enumerate(state_dict["param_groups"]):
            for idx in sg["params"]:
                st = state_dict["state"].get(idx)
                if st is not None and "step" in st:
                    steps[gi] = float(torch.as_tensor(st["step"]).reshape(-1)[0])
                    break
        super().load_state_dict(state_dict)
        self.__dict__.pop("_lists", None)                       # kernel-argument lists point at the old m / v buffers
        for gi, group in enumerate(self.param_groups):
            group.pop("step_dev", None)
            for p in group["params"]:
                self.state.get(p, {}).pop("step", None)
            if gi in steps and group["params"]:
                group["step_dev"] = torch.full((1,), steps[gi], dtype=torch.float32, device=group["params"][0].device)

    @torch.no_grad()
    def step(self, closure=None, grads=None):
        """``grads``: optional (params, tensors) pair -- gradients to use instead of ``p.grad`` (the views of a
        data-parallel flat bucket after its all-reduce, so they are consumed where RCCL left them)."""
        override = {}
        if grads is not None:
            override = {id(p): g for p, g in zip(*grads)}
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if not group["params"]:
                continue
            step = self._group_state(group)
            ps = [p for p in group["params"] if (id(p) in override or p.grad is not None)]
            stream = torch.cuda.current_stream(step.device).cuda_stream
            b1, b2 = group["betas"]
            keep = []
            chunks = [ps[i:i + _lib.ADAM_MAX_TENSORS] for i in range(0, len(ps), _lib.ADAM_MAX_TENSORS)] or [[]]
            cache = self.__dict__.setdefault("_lists", {}).setdefault(id(group), {})    # not in param_groups: state_dict() stays plain
            for ci, chunk in enumerate(chunks):
                gs = []
                for p in chunk:
                    g = override.get(id(p), p.grad)
                    if g.is_sparse:
                        raise RuntimeError("bnn_amd.optim.Adam does not support sparse gradients")
                    if not g.is_contiguous() or g.dtype != torch.float32:
                        g = g.contiguous().float()
                        keep.append(g)
                    gs.append(g)
                # the kernel-argument list is rebuilt only when a pointer changed (gradients living in a flat bucket, or
                # accumulated in place, keep their addresses: 5 ctypes stores per tensor saved on every step)
                # (the m / v addresses are part of the key: optimizer.state may be replaced or cleared between steps --
                # load_state_dict, a fresh state after a checkpoint restore -- and a stale list would update freed buffers)
                key = (tuple(p.data_ptr() for p in chunk), tuple(g.data_ptr() for g in gs),
                       tuple(self.state[p]["exp_avg"].data_ptr() for p in chunk),
                       tuple(self.state[p]["exp_avg_sq"].data_ptr() for p in chunk))
                hit = cache.get(ci)
                if hit is None or hit[0] != key:
                    lst = _lib.AdamList()
                    lst.n = len(chunk)
                    for k, (p, g) in enumerate(zip(chunk, gs)):
                        st = self.state[p]
                        lst.p[k], lst.g[k] = p.data_ptr(), g.data_ptr()
                        lst.m[k], lst.v[k], lst.numel[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                    cache[ci] = (key, lst)
                else:
                    lst = hit[1]
                rc = _lib.lib().lbbnn_adam_step(ctypes.byref(lst), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                                step.data_ptr(), 1 if ci == len(chunks) - 1 else 0, stream)
                _lib.check(rc, "lbbnn_adam_step")
            del keep
        return loss
