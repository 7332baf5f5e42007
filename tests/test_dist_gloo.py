"""CPU tier: the N>1 path (batch sharding + flat-bucket gradient all-reduce) with world_size-2 gloo.

The HIP layers have no CPU path, so the sharded ELBO is exercised here with the CPU ORACLE as the
differentiable layer arithmetic: what is under test is bnn_amd.parallel (shard bounds, KL scaling,
bucket pack/all-reduce/unpack), which is device-agnostic host logic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lbbnn_oracle as orc


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _OracleNet(torch.nn.Module):
    """Tiny LRT network whose arithmetic is the oracle's (CPU); noise indexed by GLOBAL row."""

    def __init__(self, dims, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.layers = torch.nn.ModuleList()
        for i in range(3):
            p = orc.init_lrt_params(dims[i], dims[i + 1], g)
            self.layers.append(torch.nn.ParameterDict({k: torch.nn.Parameter(v) for k, v in p.items()}))
        self.row_offset = 0
        self._kl = None

    def set_row_offset(self, off):
        self.row_offset = off

    def forward(self, x, eps_global):
        B = x.shape[0]
        eps = [e[self.row_offset:self.row_offset + B] for e in eps_global]
        out, kl = orc.lrt_network_forward(x, [dict(l) for l in self.layers], eps)
        self._kl = kl
        return out

    def kl(self):
        return self._kl


def _worker(rank, world, port, dims, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bnn_amd  # noqa: F401  (package import must work without a GPU)
    from bnn_amd.parallel import DataParallelELBO, shard_bounds
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(123)
    x = torch.rand(B, dims[0], generator=g)
    y = torch.randint(0, dims[3], (B,), generator=g)
    eps = [torch.randn(B, dims[i + 1], generator=g) for i in range(3)]
    net = _OracleNet(dims, seed=7 + rank)          # deliberately different on each rank: broadcast must fix it
    dp = DataParallelELBO(net)
    xr, yr = dp.shard(x, y)
    lo, hi = shard_bounds(B, world, rank)
    assert xr.shape[0] == hi - lo and net.row_offset == lo
    loss = dp.loss(net(xr, eps), yr, num_batches=10)
    loss.backward()
    dp.all_reduce_grads()
    grads = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    q.put((rank, grads.tolist(), params.tolist()))   # plain lists: no fd passing that needs the sender alive
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process():
    dims, B, world = (12, 10, 8, 5), 9, 2            # odd batch: uneven shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, torch.tensor(gr), torch.tensor(pr)) for r, gr, pr in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # replicas identical after broadcast, gradients identical after all-reduce
    assert torch.equal(res[0][2], res[1][2])
    assert torch.allclose(res[0][1], res[1][1], rtol=0, atol=0)
    # single-process reference on the whole batch with rank 0's (broadcast) parameters
    g = torch.Generator().manual_seed(123)
    x = torch.rand(B, dims[0], generator=g)
    y = torch.randint(0, dims[3], (B,), generator=g)
    eps = [torch.randn(B, dims[i + 1], generator=g) for i in range(3)]
    net = _OracleNet(dims, seed=7)
    out = net(x, eps)
    loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + net.kl() / 10
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.allclose(res[0][1], ref, rtol=1e-5, atol=1e-6)


def test_shard_bounds_cover_batch():
    from bnn_amd.parallel import shard_bounds
    for B in (1, 7, 8, 4096, 32768 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_grad_bucket_roundtrip_single_process():
    from bnn_amd.parallel import GradBucket
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2))]
    for p in ps[:2]:
        p.grad = torch.randn_like(p)
    want = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in ps]
    b = GradBucket(ps)
    assert b.numel == 12 + 5 + 4
    b.all_reduce()                       # no process group: identity
    for p, w in zip(ps, want):
        assert torch.equal(p.grad, w)
