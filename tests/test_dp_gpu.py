"""GPU tier: the N > 1 machinery on ONE MI355X.

  * the RCCL calls of bnn_amd.parallel actually execute: a single-rank `nccl` process group + LBBNN_DP_FORCE_COLLECTIVE=1
    makes GradBucket.collective issue all_reduce / reduce_scatter + all_gather on the device bucket (a SUM over one rank is
    the identity), in the eager step and between the two HIP graphs of make_graphed_step: parameters after 3 steps are
    BITWISE those of the step without a collective;
  * `python bench.py --gpus 2` with no launcher starts its own two ranks (gloo, the ranks share this box's one card) and
    prints ONE JSON line that proves them ("ranks": world / backend / devices);
  * building a graphed step while an eager autograd graph is still alive raises RuntimeError (round 2: a segmentation
    fault in capture_end) -- the capture is never started.
Every case runs in a process of its own (a process group, or a deliberately dirty autograd state)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keep(name, r):
    """Full stdout / stderr of a child under gpurun_out/ (merged back from the GPU box), for the post-mortem of a failure."""
    try:
        d = os.path.join(ROOT, "gpurun_out", "test_logs")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name + ".log"), "w") as f:
            f.write("rc %s\n---- stdout\n%s\n---- stderr\n%s\n" % (r.returncode, r.stdout, r.stderr))
    except OSError:
        pass


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


_DP_CODE = r"""
import os, sys, copy, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bnn_amd
from bnn_amd import layers
from bnn_amd.parallel import DataParallelELBO
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 128, 64, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
init = copy.deepcopy(net.state_dict())
x = torch.rand(256, 1, 28, 28, device=dev); y = torch.randint(0, 10, (256,), device=dev)

def run(force, mode, graphed):
    os.environ["LBBNN_DP_FORCE_COLLECTIVE"] = "1" if force else "0"
    os.environ["LBBNN_DP_COLLECTIVE"] = mode
    net.load_state_dict(init)
    opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
    dp = DataParallelELBO(net)
    assert dp.bucket.force == force and dp.bucket.mode == mode
    step = None
    if graphed:
        step = dp.make_graphed_step(opt, x, y, 100, warmup=2)
        net.load_state_dict(init)
        for st in opt.state.values():
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        for g in opt.param_groups:
            g["step_dev"].zero_()
    losses = []
    for it in range(3):
        bnn_amd.manual_seed(50 + it)
        if graphed:
            loss = step(x, y)
        else:
            opt.zero_grad(set_to_none=True)
            loss = dp.loss(net(x, sample=True), y, 100)
            with layers.vector_backward_overlap():
                loss.backward()
            dp.all_reduce_grads(unpack=False)
            opt.step(grads=dp.reduced_grads())
        losses.append(float(loss.detach()))
        del loss
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in net.named_parameters()}, losses, dp.describe_collective()

# graphed variants first (a capture wants no live eager graph; every eager loss above is dropped as well)
results = {}
for graphed in (True, False):
    for force, mode in ((False, "all_reduce"), (True, "all_reduce"), (True, "rs_ag")):
        results[(graphed, force, mode)] = run(force, mode, graphed)
base_p, base_l, base_d = results[(False, False, "all_reduce")]
assert "no exchange" in base_d, base_d
for key, (p, l, d) in results.items():
    assert l == base_l, (key, l, base_l)
    for k in base_p:
        assert torch.equal(p[k], base_p[k]), (key, k)
    if key[1]:
        assert "nccl" in d and ("all_reduce" in d or "reduce_scatter" in d), d
        assert ("reduce_scatter" in d) == (key[2] == "rs_ag"), d
assert base_l[-1] < base_l[0]
print("DPNCCL_OK", dist.get_backend(), dist.get_world_size(), sorted({v[2] for v in results.values()}))
dist.destroy_process_group()
"""


def test_single_rank_nccl_forced_collective_is_bitwise_the_plain_step():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _DP_CODE % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    _keep("dp_nccl_single_rank", r)
    assert r.returncode == 0 and "DPNCCL_OK nccl 1" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-3000:])


_GUARD_CODE = r"""
import sys, torch
sys.path.insert(0, %(root)r)
import bnn_amd
from bnn_amd import graphs
from bnn_amd.parallel import DataParallelELBO
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 64, 32, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
x = torch.rand(64, 1, 28, 28, device=dev); y = torch.randint(0, 10, (64,), device=dev)
opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
dp = DataParallelELBO(net)
assert graphs.live_autograd_nodes(net) == []
# one EAGER training step whose loss is kept: the autograd graph (and the parameters' AccumulateGrad nodes, created on the
# default stream) stay alive
opt.zero_grad(set_to_none=True)
loss = dp.loss(net(x, sample=True), y, 100)
loss.backward()
opt.step()
torch.cuda.synchronize()
live = graphs.live_autograd_nodes(net)
assert len(live) == 3 and all(s == 0 for _, s in live), live
assert net.l1.kl.grad_fn is not None
kl_before = float(net.l1.kl)
hits = 0
for make in (lambda: dp.make_graphed_step(opt, x, y, 100, warmup=1),
             lambda: graphs.make_graphed_train_step(net, opt, lambda n, a, b: dp.loss(n(a, sample=True), b, 100), x, y, warmup=1)):
    try:
        make()
    except RuntimeError as e:
        assert "still alive" in str(e) and "del loss" in str(e), str(e)
        hits += 1
assert hits == 2
assert not torch.cuda.is_current_stream_capturing()
# the refused factories have detached the network's OWN references into that graph (layer.kl is a tensor of the last graph,
# as in the reference); what is left is the caller's loss -- dropping it is what the message asks for
assert net.l1.kl.grad_fn is None and float(net.l1.kl) == kl_before
assert len(graphs.live_autograd_nodes(net)) == 3
del loss
assert graphs.live_autograd_nodes(net) == []
graphs.assert_no_live_graph(net, "test")
print("GUARD_OK")
"""


def test_graphed_step_refuses_to_capture_over_a_live_eager_graph():
    r = subprocess.run([sys.executable, "-c", _GUARD_CODE % {"root": ROOT}], capture_output=True, text=True, timeout=300)
    _keep("capture_guard", r)
    assert r.returncode == 0 and "GUARD_OK" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-3000:])


def test_bench_gpus2_self_spawned_ranks_report_themselves():
    """The driver's command shape at N = 2 with NO launcher and no WORLD_SIZE: bench.py starts its own ranks.  This box has
    one card, so the ranks share it over gloo (bench.pick_backend); what is checked is the start-up path, the rank-agreed
    control flow and the "ranks" record -- not a rate."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-secondary", "--no-kernel-events"],
                       capture_output=True, text=True, timeout=900, env=env)
    _keep("bench_gpus2_selfspawn", r)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    rk = d["ranks"]
    assert rk["world"] == 2 and rk["launcher"] == "self-spawned" and len(rk["devices"]) == 2
    assert {dv["rank"] for dv in rk["devices"]} == {0, 1} and len({dv["pid"] for dv in rk["devices"]}) == 2
    assert rk["all_reduce_of_ones"] == 2.0
    assert rk["backend"] in ("gloo", "nccl")
    assert "secondary_train" in d and d["secondary_train"]["value"] > 0        # the training leg (collective included) ran


@pytest.mark.gpu
def test_bench_training_leg_watchdog_keeps_the_headline():
    """The training leg (the only part of a bench run with a data-path collective) runs last, behind a watchdog: with
    --train-timeout 0 the watchdog fires at once -- rank 0 still prints exactly one line, the forward headline is in it,
    the leg's place holds the reason, and every process leaves with 0 (two self-spawned ranks: the other rank's own
    watchdog ends it)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-secondary", "--no-kernel-events", "--train-timeout", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    _keep("bench_train_watchdog", r)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["ranks"]["world"] == 2
    assert "did not finish" in d["secondary_train"]["error"]
