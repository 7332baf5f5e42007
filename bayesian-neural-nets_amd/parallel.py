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
import os
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
        # the buffer is padded to a multiple of 8 ranks x 4 floats so that it also splits evenly (and 16-B aligned) for
        # the reduce-scatter + all-gather form of the exchange at any world size up to one node's 8 GPUs
        self.padded = -(-max(self.numel, 1) // 32) * 32
        self._store = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        self.flat = self._store[:self.numel]
        self.mode = os.environ.get("LBBNN_DP_COLLECTIVE", "all_reduce")
        if self.mode not in ("all_reduce", "rs_ag"):
            raise ValueError("LBBNN_DP_COLLECTIVE must be 'all_reduce' or 'rs_ag'")
        # LBBNN_DP_FORCE_COLLECTIVE=1: issue the collective at world size 1 too (a SUM over one rank is the identity).  It
        # exists so that the RCCL calls of this module run on a 1-GPU box: tests/test_parity_gpu.py holds the forced step
        # to the unforced one bit for bit
        self.force = os.environ.get("LBBNN_DP_FORCE_COLLECTIVE", "0") == "1"
        self.last = None

    def _slices(self):
        off = 0
        for p in self.params:
            n = p.numel()
            yield p, self.flat[off:off + n]
            off += n

    def views(self):
        """Per-parameter views of the flat buffer (what ``bnn_amd.optim.Adam.step(grads=...)`` consumes, so the reduced
        gradients never have to be copied back into ``p.grad``)."""
        return [s.view_as(p) for p, s in self._slices()]

    def _multi_copy(self, pairs):
        """pairs: (dst, src-or-None) fp32 HIP tensors -> lbbnn_multi_copy, <= 80 tensors per launch."""
        import ctypes
        from . import _lib
        stream = torch.cuda.current_stream(self.flat.device).cuda_stream
        for i in range(0, len(pairs), _lib.ADAM_MAX_TENSORS):
            chunk = pairs[i:i + _lib.ADAM_MAX_TENSORS]
            lst = _lib.CopyList()
            lst.n = len(chunk)
            for k, (d, s) in enumerate(chunk):
                lst.dst[k], lst.src[k], lst.numel[k] = d.data_ptr(), (s.data_ptr() if s is not None else None), d.numel()
            _lib.check(_lib.lib().lbbnn_multi_copy(ctypes.byref(lst), stream), "lbbnn_multi_copy")

    def _hip_ok(self, grads):
        return self.flat.is_cuda and all(g is None or (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()) for g in grads)

    def pack(self):
        grads = [p.grad for p in self.params]
        if self._hip_ok(grads):                          # one launch for all parameters
            self._multi_copy([(s, g) for (p, s), g in zip(self._slices(), grads)])
            return
        for (p, s), g in zip(self._slices(), grads):
            if g is None:
                s.zero_()
            else:
                s.copy_(g.reshape(-1))

    def unpack(self):
        if self.flat.is_cuda and all(p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32
                                     for p in self.params):
            self._multi_copy([(p.grad, s) for p, s in self._slices()])
            return
        for p, s in self._slices():
            g = s.view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)

    def wants_collective(self, group=None) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(group) > 1 or self.force

    def collective(self, group=None):
        """The exchange itself, on the flat buffer as it stands (no pack / unpack): SUM over the ranks of `group`."""
        if self.wants_collective(group):
            world = dist.get_world_size(group)
            if self.mode == "rs_ag" and self.padded % world == 0:
                # xGMI is a full point-to-point mesh: reduce-scatter + all-gather moves 1/world of the bucket per peer
                # over every link at once, where a ring all-reduce pushes 2(world-1)/world of it through one link
                # (SURVEY.md section 5); which of the two RCCL runs faster is measured, not assumed: bench.py --train
                shard = self._store.view(world, -1)[dist.get_rank(group)]
                dist.reduce_scatter_tensor(shard, self._store, op=dist.ReduceOp.SUM, group=group)
                dist.all_gather_into_tensor(self._store, shard, group=group)
                self.last = "reduce_scatter_tensor + all_gather_into_tensor (%d B per rank shard)" % (shard.numel() * 4)
            else:
                dist.all_reduce(self._store, op=dist.ReduceOp.SUM, group=group)
                self.last = "all_reduce (%d B)" % (self._store.numel() * 4)
            return True
        return False

    def all_reduce(self, group=None, unpack: bool = True):
        """pack -> all_reduce(SUM) -> unpack.  ``unpack=False`` leaves the reduced gradients in the flat buffer only
        (use ``views()`` with ``bnn_amd.optim.Adam.step(grads=...)``)."""
        self.pack()
        self.collective(group)
        if unpack:
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
        self._exchange = on and (self.world > 1 or self.bucket.force)
        if self._exchange:
            # replicas must start identical: broadcast rank 0's parameters once
            for p in net.parameters():
                dist.broadcast(p.data, src=0, group=group)
        self.sync_rng()

    def sync_rng(self):
        """The contract of the sharded forward -- the SAME z on every rank, a KL identical on every rank, rank-distinct
        eps through ``row_offset`` -- holds only while every rank's Philox {seed, offset} pair is the same: rank 0's pair
        is broadcast here (construction) and may be re-broadcast by the caller after rank-local work that drew noise on
        some ranks only (an evaluation pass on rank 0)."""
        p0 = next(iter(self.net.parameters()), None)
        if self._exchange and p0 is not None and p0.is_cuda:
            from . import ops
            st = ops.RngState.get(p0.device)
            live = st.t[:2].clone()
            dist.broadcast(live, src=0, group=self.group)
            st.t[:2].copy_(live)

    def bucket_numel(self) -> int:
        return self.bucket.numel

    def make_graphed_step(self, optimizer, example_x, example_y, num_batches, warmup: int = 3):
        """The data-parallel training step as TWO HIP graphs around ONE collective:

            graph A   zero_grad -> forward -> nll + kl / (num_batches * world) -> backward (HIP kernels, vector chains
                      deferred and batched) -> all gradients packed into the flat bucket (one launch)
            eager     all-reduce of the bucket over RCCL (world > 1; nothing at world 1)
            graph B   bnn_amd.optim.Adam on the reduced bucket (one launch)

        Three host calls per step instead of ~90 launches: the eager step of the headline net is host-bound at ~2 ms, the
        graphs run at GPU speed.  The collective stays OUTSIDE the graphs on purpose: capturing RCCL inside a graph works
        in principle, but a capture that misbehaves would hang all N ranks, and an N-GPU node is not available to the
        build to test it on -- between two graphs the collective is the plain, well-trodden ``dist.all_reduce``.
        ``optimizer`` must be capture-safe (``bnn_amd.optim.Adam``).  Returns step(x, y) -> loss (a static tensor).
        Raises RuntimeError while an autograd graph of an earlier forward through the network is still alive (a loss or
        an output of an eager step that has not been dropped): the parameters' AccumulateGrad nodes remember the stream
        they were created on, and a node created on the default stream makes autograd synchronise the capture stream
        with the default stream -- which a HIP stream capture does not survive (round 2: a segmentation fault in
        capture_end).  ``graphs.assert_no_live_graph`` is the check."""
        from . import graphs, layers
        graphs.assert_no_live_graph(self.net, "parallel.DataParallelELBO.make_graphed_step")
        net, dev = self.net, example_x.device
        ov = layers.vector_backward_overlap
        static_x, static_y = example_x.clone(), example_y.clone()
        one = torch.ones((), dtype=torch.float32, device=dev)       # the root gradient, made once (graphs.make_graphed_train_step)

        def fwd_bwd():
            optimizer.zero_grad(set_to_none=True)
            loss = self.loss(net(static_x, sample=True), static_y, num_batches)
            with ov():
                loss.backward(one if (loss.dim() == 0 and loss.dtype == torch.float32) else None)
            self.bucket.pack()
            return loss

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fwd_bwd()
                self.bucket.collective(self.group)
                optimizer.step(grads=self.reduced_grads())
        torch.cuda.current_stream(dev).wait_stream(side)
        _quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if _quiet is not None:
            _quiet(False)
        g_a, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with graphs.capture(g_a):
            static_loss = fwd_bwd()
        with graphs.capture(g_b, pool=g_a.pool()):
            optimizer.step(grads=self.reduced_grads())
        if _quiet is not None:
            _quiet(True)
        if self.bucket.last:
            self.bucket.last += " between two HIP graphs"

        def step(x, y):
            if x.data_ptr() != static_x.data_ptr():       # (a batch written straight into step.inputs needs no copy)
                static_x.copy_(x)
            if y.data_ptr() != static_y.data_ptr():
                static_y.copy_(y)
            g_a.replay()
            self.bucket.collective(self.group)
            g_b.replay()
            return static_loss

        step.graphs = (g_a, g_b)
        step.inputs = (static_x, static_y)
        step._root_grad = one                          # read by every replay of graph A: lives as long as the step
        return step

    def describe_collective(self) -> str:
        backend = dist.get_backend(self.group) if (dist.is_available() and dist.is_initialized()) else "none"
        return "%s over %d rank(s), backend %s" % (self.bucket.last or ("no exchange (world 1)" if self.world == 1
                                                                       else self.bucket.mode), self.world, backend)

    def shard(self, *tensors):
        B = tensors[0].shape[0]
        lo, hi = shard_bounds(B, self.world, self.rank)
        if hasattr(self.net, "set_row_offset"):
            self.net.set_row_offset(lo)
        out = tuple(t[lo:hi] for t in tensors)
        return out if len(out) > 1 else out[0]

    def loss(self, log_probs, target, num_batches):
        """nll_r + kl / (num_batches * world) (module docstring); on HIP tensors one fused launch (bnn_amd.losses.elbo_loss)."""
        from .losses import elbo_loss
        return elbo_loss(log_probs, target, self.net.kl(), num_batches * self.world)

    def all_reduce_grads(self, unpack: bool = True):
        self.bucket.all_reduce(self.group, unpack=unpack)

    def reduced_grads(self):
        """(parameters, gradient views into the reduced flat bucket) for ``optim.Adam.step(grads=...)``."""
        return self.bucket.params, self.bucket.views()
