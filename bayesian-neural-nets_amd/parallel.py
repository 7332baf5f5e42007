"""Data-parallel ELBO training across the GPUs of one node.

The reference is single-device.  The ELBO minibatch shards over rows with no data-path collective in
the forward: every rank holds the replicated parameters, draws the SAME z vectors (their Philox
streams are keyed by seed / offset / layer only) and rank-distinct activation noise (the epsilon
counter is the GLOBAL row index, set through ``set_row_offset``).  The only exchange is the gradient:
ONE flat fp32 bucket all-reduced over RCCL (``backend="nccl"`` on ROCm) -- xGMI is point-to-point
(7 links/GPU), so one large collective beats many small ones.

Per-rank loss convention: ``loss_r = nll_r + kl / (num_batches * world)``.  Summing the gradients of
loss_r over ranks gives the gradient of  nll(global batch) + kl / num_batches,  i.e. exactly what the
reference's ``train()`` (LBBNN-GP-MF-MNF.py:268-272) computes on the whole batch; the KL (parameters
only, identical on every rank) is counted once.
"""
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of `rank`; the first (global_batch % world) ranks take one extra row."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """One flat fp32 buffer for all gradients: pack -> all_reduce(SUM) -> unpack."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)

    def pack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n

    def unpack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    def all_reduce(self, group=None):
        self.pack()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.unpack()


class DataParallelELBO:
    """Wraps a BayesianNetwork for one-process-per-GPU data parallel training.

        dp = DataParallelELBO(net)                       # after dist.init_process_group("nccl")
        x_r, y_r = dp.shard(x, y)                        # this rank's rows of the global batch
        loss = dp.loss(net(x_r, sample=True), y_r, num_batches)
        loss.backward(); dp.all_reduce_grads(); optimizer.step()
    """

    def __init__(self, net, group=None):
        self.net = net
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.bucket = GradBucket(net.parameters())
        if self.world > 1:
            # replicas must start identical: broadcast rank 0's parameters once
            for p in net.parameters():
                dist.broadcast(p.data, src=0, group=group)

    def shard(self, *tensors):
        B = tensors[0].shape[0]
        lo, hi = shard_bounds(B, self.world, self.rank)
        if hasattr(self.net, "set_row_offset"):
            self.net.set_row_offset(lo)
        out = tuple(t[lo:hi] for t in tensors)
        return out if len(out) > 1 else out[0]

    def loss(self, log_probs, target, num_batches):
        nll = torch.nn.functional.nll_loss(log_probs, target, reduction="sum")
        return nll + self.net.kl() / (num_batches * self.world)

    def all_reduce_grads(self):
        self.bucket.all_reduce(self.group)
