#!/usr/bin/env python3
"""ELBO-forward throughput of the MNF Bayesian MLP on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one training-mode forward of the 784-1200-1200-10 MNF network (2 planar flows per
layer) over one synthetic MNIST-shaped batch of 4096 rows PER GPU: the three layers' sampled
activations, log_softmax, and net.kl() -- what the reference's train() runs before .backward()
(LBBNN-GP-MF-MNF.py:268-270).  Inputs are resident in HBM before the timed region; noise is drawn
in-kernel (Philox).  Data parallel: every rank holds the (replicated) parameters and its own
4096 rows (weak scaling); the forward has no collective.

Headline arithmetic since round 3: the row-scaled fp16 hi + lo operand format with 3 mean + 1 variance MFMA products
(--precision fp16x3f; 1.4-1.8e-5 of max|out| against fp64, contract 1e-4); the strict 3 + 3 form (fp16x3, 3e-8), the exact
fp32 MFMA path and the reduced single-product bf16 mode are timed in the same process as labelled secondary legs.
`python bench.py --gpus N` with no launcher starts its own N ranks (spawn_ranks) and reports them ("ranks").

How the number is taken (round 2; the round-1 line did not reproduce under the driver's command):
  1. W untimed warm-up steps (--warmup, honoured and reported), then warm-up CONTINUES to steady
     state: blocks of 25 steps run back to back (no synchronisation between them) until the last
     4 block times agree within 2 % or 0.5 s has passed ("settle" in the JSON) -- a fresh GPU needs
     tens of milliseconds to reach its clocks, far longer than 5 warm-up steps.
  2. the timed region: barrier + synchronize, EXACTLY K steps enqueued back to back with nothing
     between them, barrier + synchronize; MAX over ranks; ms_per_step = elapsed / K.  Nothing slow
     sits between the settle phase and the region (events come from a pool made beforehand, the
     collector is off): an idle GPU drops its clocks within milliseconds.  The per-step distribution
     (ms_per_step_median/min/max: an outlier step is visible) comes from a SEPARATE pass of the same
     K steps with one event per step boundary (an event record is a barrier packet: ~3 us per step
     when it sat inside the region, as it did until late in round 3).  A region whose mean step is
     > 2x the settled step (a single ~40 ms stall hits a run on this pool now and then, whatever is
     running) is timed again, at most twice; every attempt is listed in "timed_attempts" and the
     reported one is the first without a stall.
  3. AFTER the timed region, a separate pass brackets every GEMM launch with HIP events on the
     launch stream ("roofline", sampled_in = "separate pass after the timed region").  A roofline
     that contradicts the timed region (share of step > 1, launch longer than a step) is not
     printed: "roofline_invalid" carries the reason instead.
     A step is one call of a RECORDED LAUNCH PLAN (bnn_amd.graphs.LaunchPlan: the forward's C calls
     -- 3 calls, 5 kernels in the fp16 forward -- recorded once after the warm-up and made again from a list; --graph:
     one HIP-graph replay; --eager: the launches issued from Python each step); the noise is fresh
     on every call -- the Philox offset lives on the device and is advanced by the forward's kernels.
  4. the same three stages again with the exact-fp32 MFMA GEMM ("secondary": reference precision) and with the strict
     3 + 3 product fp16 form ("secondary_strict_fp16x3").
  4b. at N > 1: the strong-scaling form (the same 4096 rows split N ways) as "secondary_strong".
  5. with --train (default at N > 1): the full data-parallel training step (forward, backward,
     flat-bucket gradient all-reduce over RCCL, Adam) as "secondary_train".

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  roofline      dominant kernel (the 80x128-tile dual-moment GEMM) vs the MFMA peak of its dtype; frac_rocprof = the same
                fraction from the committed rocprofv3 summary of this command (profiles/r03_kernel_stats_bench_default.csv);
  cpu_baseline  the CPU oracle (port of the reference op sequence, as-written B-row z flow) timed
                on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

DIMS = (784, 1200, 1200, 10)
T_FLOWS = 2
BATCH = 4096
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md, dense BF16 MFMA
HBM_PEAK_GBS = 8000.0
SETTLE_BLOCK = 25        # steps per block of the steady-state warm-up
SETTLE_WINDOW = 4        # ... whose last this-many block times must agree
SETTLE_TOL = 0.02        # ... within this
SETTLE_MAX_S = 0.5


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="rows per GPU (headline: 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel roofline pass")
    ap.add_argument("--no-step-events", action="store_true", help="no per-step event marks inside the timed region")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fp32 (reference-precision) leg")
    ap.add_argument("--no-reduced", action="store_true",
                    help="skip the reduced-precision leg (one bf16 product per moment: OUTSIDE the 1e-4 contract)")
    ap.add_argument("--train", dest="train", action="store_true", default=None,
                    help="also time the data-parallel training step (default: only when N > 1)")
    ap.add_argument("--train-timeout", type=int, default=240,
                    help="seconds after which a training leg that has not finished is abandoned (the line is printed without it)")
    ap.add_argument("--no-train", dest="train", action="store_false")
    ap.add_argument("--plan", dest="launch", action="store_const", const="plan", default="plan",
                    help="(default) a step = bnn_amd.graphs.LaunchPlan: the forward's C calls (3 calls = 5 kernels in the fp16 forward) recorded once after "
                         "the warm-up and replayed from a list -- the eager launches without the Python between them (~30 us of "
                         "host time per step instead of ~100), fresh Philox noise on every call (the offset lives on the device)")
    ap.add_argument("--graph", dest="launch", action="store_const", const="graph",
                    help="a step = one replay of the forward captured in a HIP graph (~6 us per replay slower than the plan on "
                         "this stack: the gap between two graph launches)")
    ap.add_argument("--eager", dest="launch", action="store_const", const="eager",
                    help="launch every step from Python (~100 us of host time against ~165 us of GPU time: as fast as the plan in "
                         "a long run, 0-8 %% slower in a 20-step region that starts from an idle queue)")
    ap.add_argument("--precision", choices=("fp16x3f", "fp16x3", "bf16x3", "fp32"), default="fp16x3f",
                    help="GEMM arithmetic of the headline leg (bnn_amd.ops.PRECISIONS): fp16x3f = row-scaled fp16 hi + lo operands, "
                         "3 mean + 1 variance products on the fp16 matrix cores, fp32 accumulate (1.4-1.8e-5 of max|out| against "
                         "fp64 on the headline layers; contract 1e-4); fp16x3 = the same operands, 3 + 3 products (3e-8: tighter "
                         "than an fp32-accumulate torch.mm); bf16x3 = round 2's headline format (2.7e-6); fp32 = exact fp32 MFMA")
    return ap.parse_args()


PROFILE_TAG = "r03"


def pmc_traffic(precision):
    """HBM-side bytes per launch of the dominant GEMM from the committed rocprofv3 PMC pass (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE; counters cannot be read from inside the process) -- None if no pass is committed for
    this precision."""
    for name in ("%s_pmc_gemm_traffic.json" % PROFILE_TAG, "r02_pmc_gemm_traffic.json", "r01_e_pmc_gemm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            for rec in (d if isinstance(d, list) else [d]):
                if rec.get("precision") == precision:
                    return rec["hbm_bytes_per_launch"], rec["source"]
        except (OSError, ValueError, KeyError):
            pass
    return None, None


# the dominant GEMM instantiation of each precision as rocprofv3 prints it (demangled; both x sources of the fp16 kernel)
KERNEL_OF = {"fp16x3f": "lrt_gemm_f16s_kernel<5, 2, 4, 1,", "fp16x3": "lrt_gemm_f16s_kernel<5, 2, 4, 3,",
             "bf16x3": "lrt_gemm_bf16x3_kernel<5, 2, 4, false, 3,", "bf16": "lrt_gemm_bf16x3_kernel<5, 2, 4, false, 1,",
             "fp32": "lrt_gemm_f32_dma_kernel<5, 2, 4,"}


def rocprof_avg_us(precision):
    """Average dispatch duration (us) of this precision's dominant GEMM kernel in the committed `rocprofv3 --kernel-trace
    --stats` summary of the bench command (profiles/<tag>_kernel_stats_bench_default.csv), or (None, None)."""
    import csv
    path = os.path.join(ROOT, "profiles", "%s_kernel_stats_bench_default.csv" % PROFILE_TAG)
    try:
        with open(path, newline="") as f:
            tot = cnt = 0.0
            for row in csv.DictReader(f):
                if KERNEL_OF.get(precision, "?") in row.get("Name", ""):
                    tot += float(row["TotalDurationNs"]); cnt += float(row["Calls"])
        if cnt:
            return tot / cnt / 1e3, os.path.relpath(path, ROOT)
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def host_cores():
    """Host cores this job may actually use: the cgroup CPU quota when there is one (the GPU box
    gives a 1-GPU job a share of the host), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("BNN_CPU_THREADS", "16")))   # gpurun: 16-core share per GPU


def cpu_baseline(batch, seconds):
    """The oracle's restatement of the reference op sequence on the host cores (kind 'port').

    As written in the reference: the z flow runs on all B rows (of which only the last is kept),
    randn draws included.  Returns samples/s (median of the timed iterations)."""
    from oracle import lbbnn_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    layers, zf, rf = [], [], []
    for i in range(3):
        I, O = DIMS[i], DIMS[i + 1]
        layers.append(orc.init_mnf_params(I, O, g))
        zf.append(orc.init_planar_flow(I, T_FLOWS, g))
        rf.append(orc.init_planar_flow(I, T_FLOWS, g))
    x = torch.rand(batch, DIMS[0], generator=g)

    def one():
        noise = []
        for i in range(3):
            I, O = DIMS[i], DIMS[i + 1]
            noise.append({"eps_z": torch.randn(batch, I), "eps_out": torch.randn(batch, O),
                          "eps_z2": torch.randn(1, I), "eps_act": torch.randn(O)})
        out, kl = orc.mnf_network_forward(x, layers, zf, rf, noise)
        return out, kl

    with torch.no_grad():
        t0 = time.perf_counter()
        one()
        first = time.perf_counter() - t0
        one()
        iters = max(3, min(50, int(seconds / max(first, 1e-3)) - 2))
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            one()
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    res = {"value": batch / med, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "%d ELBO forwards of the same 784-1200-1200-10 MNF/planar net at batch %d "
                     "(torch-CPU fp32 oracle, as-written B-row z flow, randn draws included), median %.1f ms"
                     % (iters, batch, med * 1e3)}
    # the same forward with autograd recording, as train() runs it before .backward() (both); a few iterations
    for pd in layers + [t for f in zf + rf for t in f.transforms]:
        for v in pd.values():
            if isinstance(v, torch.Tensor) and v.is_floating_point():
                v.requires_grad_(True)
    tg = []
    for _ in range(max(3, min(10, iters // 4))):
        t0 = time.perf_counter()
        one()
        tg.append(time.perf_counter() - t0)
    tg.sort()
    res["value_grad_enabled"] = batch / tg[len(tg) // 2]
    return res


def _recorded_events(n):
    """n timing events, each recorded once already: torch creates the HIP event at the first record()."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    return evs


_EVENT_POOL = []


def _event_pool(n):
    """n recorded timing events from a pool that only grows: no event is created (and no device synchronisation made)
    between a settle phase and its timed region after the first region of the process."""
    if len(_EVENT_POOL) < n:
        _EVENT_POOL.extend(_recorded_events(n - len(_EVENT_POOL)))
    return _EVENT_POOL[:n]


def settle(run_step, world=1):
    """Warm-up to steady state.  Blocks of SETTLE_BLOCK steps are enqueued back to back with one event between blocks;
    the host only ever waits for the block BEFORE the one it has just enqueued, so the GPU queue never drains (a
    synchronisation between blocks would put the ~0.1 ms pipeline refill into every measurement -- the first settle
    logic of this round did, and called a GPU that was still ramping its clocks converged).  Done when the last
    SETTLE_WINDOW block times agree within SETTLE_TOL, or after SETTLE_MAX_S.  Returns the report for the JSON line."""
    max_blocks = 256
    evs = _recorded_events(max_blocks + 2)
    times, t_start = [], time.perf_counter()
    evs[0].record()
    for _ in range(SETTLE_BLOCK):
        run_step()
    evs[1].record()
    k, ok = 1, False
    while True:
        for _ in range(SETTLE_BLOCK):
            run_step()
        evs[k + 1].record()
        evs[k].synchronize()                                   # block k-1 is done; block k is running
        times.append(evs[k - 1].elapsed_time(evs[k]) / SETTLE_BLOCK)
        k += 1
        w = times[-SETTLE_WINDOW:]
        ok = len(w) == SETTLE_WINDOW and max(w) <= (1.0 + SETTLE_TOL) * min(w)
        stop = ok or time.perf_counter() - t_start > SETTLE_MAX_S or k >= max_blocks
        if world > 1:
            # every rank must run the same number of steps (the training step contains a collective): stop together,
            # when the LAST rank is ready
            import torch.distributed as dist
            flag = torch.tensor([0.0 if stop else 1.0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            stop = bool(flag.item() == 0.0) or k >= max_blocks
        if stop:
            break
    return {"steps": (k + 1) * SETTLE_BLOCK, "ms": (time.perf_counter() - t_start) * 1e3, "converged": bool(ok),
            "first_block_ms_per_step": times[0], "last_blocks_ms_per_step": times[-SETTLE_WINDOW:]}


STALL_FACTOR = 10.0     # a step this many times longer than the region's median step is an external stall
MAX_ATTEMPTS = 3


STALL_REGION = 2.0      # a timed region whose mean step is this many times the settled step contains an external stall


def timed_region(run_step, steps, sync, step_events, attempts=None, world=1, steady_ms=None):
    """The timed region: barrier + synchronize, EXACTLY `steps` steps enqueued back to back with NOTHING between them,
    barrier + synchronize.  (Until late in round 3 one HIP event marked every step boundary inside the region: an event
    record is a barrier packet on this stack, and the ~3 us bubble behind each of them was 2-3 % of a 0.125 ms step.  The
    per-step distribution now comes from a SEPARATE instrumented pass after the region: "ms_per_step_median/min/max".)
    The region is repeated (at most MAX_ATTEMPTS times) while its mean step is more than STALL_REGION x the settled step of
    the settle phase, or -- in the instrumented pass -- a step took more than STALL_FACTOR x the median.  Why: on this pool a
    single ~40 ms stall hits a run now and then (round 1's driver line: 37 ms inside one GEMM launch's bracket; a round-2 run
    of the fp32 leg: 48 ms for 20 steps that take 7.7), unrelated to the step being timed.  Every attempt is listed in the
    JSON line ("timed_attempts"); the reported value is the first attempt without a stall (the last one if all have one).
    With N > 1 all ranks must agree on repeating: the decision is all-reduced (MAX)."""
    while True:
        elapsed, _ = _timed_region_once(run_step, steps, sync, False)
        rec = {"ms_per_step": elapsed / steps * 1e3}
        stalled = bool(steady_ms and rec["ms_per_step"] > STALL_REGION * steady_ms)
        per_step = None
        if step_events:
            _, per_step = _timed_region_once(run_step, steps, sync, True)         # the separate, instrumented pass
            srt = sorted(per_step)
            rec["max_step_ms"], rec["median_step_ms"] = srt[-1], srt[len(srt) // 2]
            rec["step_stats_from"] = "a separate pass of the same %d steps with one event per step boundary" % steps
        if world > 1:
            import torch.distributed as dist
            flag = torch.tensor([1.0 if stalled else 0.0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            stalled = bool(flag.item() > 0)
        rec["stall"] = stalled
        if attempts is not None:
            attempts.append(rec)
        if not stalled or (attempts is not None and len(attempts) >= MAX_ATTEMPTS) or attempts is None:
            return elapsed, per_step


def _timed_region_once(run_step, steps, sync, step_events):
    """barrier + synchronize, EXACTLY ``steps`` steps, barrier + synchronize.  Returns (elapsed seconds, per-step ms)."""
    # Nothing slow may sit between the end of the settle phase and the timed steps: a GPU left idle for milliseconds drops its
    # clocks and 20 steps (2.5 ms) are over before it has them back (measured: a gc.collect() placed here cost the line 10 %).
    # The events are made once, ahead of the first timed region of the process; the collector is off inside the region.
    marks = _event_pool(steps + 1) if step_events else None
    import gc
    gc.disable()
    sync()
    t0 = time.perf_counter()
    if marks is not None:
        marks[0].record()
        for i in range(steps):
            run_step()
            marks[i + 1].record()
    else:
        for _ in range(steps):
            run_step()
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)] if marks is not None else None
    return elapsed, per_step


def step_stats(res, per_step):
    if per_step:
        s = sorted(per_step)
        res["ms_per_step_median"] = s[len(s) // 2]
        res["ms_per_step_min"] = s[0]
        res["ms_per_step_max"] = s[-1]


def roofline_pass(ops, run_step, sync, n_steps, launches_per_step):
    """The separate pass that brackets every GEMM launch (events pre-created AND pre-recorded).  Returns the log."""
    log = ops.GemmEventLog(launches_per_step * (n_steps + 2), group=launches_per_step, every=1)
    run_step()                      # queue depth: the first bracketed launch is not the first thing on an idle GPU
    ops.GEMM_EVENTS = log
    try:
        for _ in range(n_steps):
            run_step()
        sync()
    finally:
        ops.GEMM_EVENTS = None
    # what a bracket itself costs: the same two events around a ~1 us kernel (the RNG advance by 0), in the same kind of loop.
    # An event is a queue packet of its own, so a bracketed launch also pays the dispatch latency of an isolated kernel; rocprofv3
    # times the dispatch alone.  Reported beside the roofline, not subtracted from it (`achieved` stays the conservative number).
    st = ops.RngState.get(torch.device("cuda", torch.cuda.current_device()))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(24)]
    for s, e in evs:
        s.record(); e.record()
    sync()
    for s, e in evs:
        run_step()
        s.record(); st.advance(0); e.record()
    sync()
    fl = sorted(s.elapsed_time(e) for s, e in evs)
    log.bracket_floor_us = fl[len(fl) // 2] * 1e3
    log.n_steps = n_steps          # (2 GEMM launches per step when the 10-class head is folded into the second one, else 3)
    return log


def roofline_object(events, precision, ms_per_step, sampled_in):
    floor_us = getattr(events, "bracket_floor_us", None)
    """(roofline, None) or (None, reason).  Dominant kernel = the <5,2,4> instantiation (80x128 tile) = the layer-1 and
    layer-2 GEMMs; achieved = ALGORITHMIC 4*B*I*O flop per launch / mean HIP-event time of those launches."""
    big = [(b, i, o, s.elapsed_time(e)) for (b, i, o, s, e) in events if o > 16]
    if not big:
        return None, "no GEMM launch was bracketed"
    n_steps = getattr(events, "n_steps", None) or max(len(events) // 3, 1)
    flops = sum(4.0 * b * i * o for (b, i, o, _) in big) / len(big)
    avg_ms = sum(ms for (_, _, _, ms) in big) / len(big)
    med_ms = sorted(ms for (_, _, _, ms) in big)[len(big) // 2]
    share = (sum(s.elapsed_time(e) for (_, _, _, s, e) in events) / n_steps) / ms_per_step
    if avg_ms > ms_per_step:
        return None, "average bracketed GEMM launch %.1f us is longer than a whole step of the timed region (%.1f us)" % (
            avg_ms * 1e3, ms_per_step * 1e3)
    if share > 1.0:
        return None, "bracketed GEMM time per step is %.2f x the timed region's step time" % share
    # products executed per algorithmic product on the 16-bit matrix cores: bf16x3 / fp16x3 3 + 3 of 2, fp16x3f 3 + 1 of 2
    executed = {"bf16x3": 3.0, "fp16x3": 3.0, "fp16x3f": 2.0, "bf16": 1.0, "fp32": 1.0}[precision]
    peak = FP32_MFMA_PEAK_TFLOPS if precision == "fp32" else BF16_MFMA_PEAK_TFLOPS      # fp16 and bf16 MFMA: the same rate
    kernel = {"fp16x3f": "lrt_gemm_f16s_kernel<5,2,4,NPV=1>", "fp16x3": "lrt_gemm_f16s_kernel<5,2,4,NPV=3>",
              "bf16x3": "lrt_gemm_bf16x3_kernel<5,2,4>", "bf16": "lrt_gemm_bf16x3_kernel<5,2,4,NP=1>",
              "fp32": "lrt_gemm_f32_dma_kernel<5,2,4>"}[precision]
    ach = flops / (avg_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(precision)
    prof_us, prof_src = rocprof_avg_us(precision)
    roof = {"bound": "mfma", "kernel": kernel + " (dual-moment GEMM, 80x128 tile)",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "traffic": traffic, "traffic_unit": "HBM-side bytes per launch", "traffic_source": traffic_src,
            "traffic_note": "FETCH_SIZE x2 + WRITE_SIZE = what the 8 private L2s request from the fabric (Infinity-Cache hits "
                            "included); DESIGN.md 7.4",
            "executed_mfma_tflops": ach * executed,
            "note": "achieved = ALGORITHMIC 4*B*I*O flop per launch / HIP-event time; this format executes %g 16-bit MFMA "
                    "products per algorithmic product" % executed if executed != 1.0 else
                    "achieved = ALGORITHMIC 4*B*I*O flop per launch / HIP-event time",
            "avg_launch_us": avg_ms * 1e3, "median_launch_us": med_ms * 1e3, "launches": len(big),
            "event_bracket_floor_us": floor_us,
            "event_bracket_note": "the same two HIP events around a ~1 us kernel read event_bracket_floor_us: a bracket includes "
                                  "the dispatch latency of an isolated launch, which rocprofv3's per-dispatch duration does not",
            "gemm_share_of_step": share, "sampled_steps": n_steps, "sampled_in": sampled_in}
    if prof_us:
        # the same fraction from the COMMITTED rocprofv3 summary of this command: line and profiles/ agree
        roof["rocprof_avg_launch_us"] = prof_us
        roof["frac_rocprof"] = flops / (prof_us * 1e-6) / 1e12 / peak
        roof["rocprof_source"] = prof_src
    return roof, None


def forward_leg(args, bnn_amd, ops, net, x, sync, precision, world):
    """warm-up -> settle -> timed region -> roofline pass for one GEMM precision.  Returns a dict of raw results."""
    net.set_precision(precision)

    def step():
        out = net(x, sample=True)
        return out, net.kl()

    leg = {}
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        sync()
        run = step
        leg["launch"], leg["launch_fallback_reason"] = "eager", None
        try:
            if args.launch == "plan":
                from bnn_amd import graphs
                plan = graphs.LaunchPlan(net, x, sample=True)
                out, kl = plan.out, plan.kl
                run = plan
                leg["launch"] = "plan"
            elif args.launch == "graph":
                from bnn_amd import graphs
                graph = torch.cuda.CUDAGraph()
                with graphs.capture(graph):
                    out, kl = step()
                run = graph.replay
                leg["launch"] = "graph"
        except Exception as exc:           # a recording / capture that fails (e.g. a communicator's watchdog touching the device
            sync()                         # during a capture) must not cost the run: the leg is launched from Python, and says so
            run = step
            leg["launch_fallback_reason"] = "%s: %s" % (type(exc).__name__, str(exc)[:200])
        if not args.no_step_events:
            _event_pool(args.steps + 1)                      # (made before the settle phase, not between it and the timed region)
        leg["settle"] = settle(run, world)
        leg["attempts"] = []
        elapsed, per_step = timed_region(run, args.steps, sync, not args.no_step_events, leg["attempts"], world,
                                         steady_ms=leg["settle"]["last_blocks_ms_per_step"][-1])
        if leg["launch"] == "eager":
            out, kl = step()
        sync()
        assert torch.isfinite(out).all() and torch.isfinite(kl)
        t = torch.tensor([elapsed], device=x.device, dtype=torch.float64)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        leg["elapsed"] = float(t.item())
        leg["per_step"] = per_step
        leg["events"] = None
        if not args.no_kernel_events:
            # kernels inside a replayed graph cannot be bracketed; the pass always runs the eager launches
            leg["events"] = roofline_pass(ops, step, sync, max(8, min(args.steps, 40)), 3)
    return leg


def train_leg(args, bnn_amd, net, x, sync, world, rank):
    """The data-parallel training step of BASELINE configs[3]: forward, nll + kl/(num_batches*world), backward (HIP),
    ONE flat-bucket gradient all-reduce (RCCL when world > 1), bnn_amd.optim.Adam on the reduced bucket."""
    from bnn_amd import parallel, optim
    dev = x.device
    net.set_precision(args.precision)
    dp = parallel.DataParallelELBO(net)
    opt = optim.Adam(net.parameters(), lr=1e-3)
    y = torch.randint(0, DIMS[-1], (x.shape[0],), device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank))

    def eager_step():
        opt.zero_grad(set_to_none=True)
        loss = dp.loss(net(x, sample=True), y, 600)
        loss.backward()
        dp.all_reduce_grads(unpack=False)
        opt.step(grads=dp.reduced_grads())
        return loss

    # two HIP graphs around the eager collective (parallel.DataParallelELBO.make_graphed_step); the plain eager step if
    # the capture fails for any reason (recorded in the JSON)
    mode, why = "two HIP graphs around the collective", None
    try:
        gstep = dp.make_graphed_step(opt, x, y, 600)
        gx, gy = gstep.inputs                    # the batch is resident in the graph's own input buffers (as x is for the forward legs)
        gx.copy_(x.view_as(gx)); gy.copy_(y)
        step = lambda: gstep(gx, gy)
    except Exception as e:                                   # noqa: BLE001 -- report, fall back, keep the bench alive
        mode, why = "eager", "%s: %s" % (type(e).__name__, str(e)[:200])
        step = eager_step
    for _ in range(max(3, min(args.warmup, 10))):
        step()
    sync()
    steps = max(5, min(args.steps, 50))
    if not args.no_step_events:
        _event_pool(steps + 1)
    st = settle(step, world)
    attempts = []
    elapsed, per_step = timed_region(step, steps, sync, not args.no_step_events, attempts, world,
                                     steady_ms=st["last_blocks_ms_per_step"][-1])
    loss = step()
    sync()
    assert torch.isfinite(loss)
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    res = {"what": "data-parallel training step (%s): ELBO forward, HIP backward, one flat fp32 gradient bucket "
                   "all-reduced over RCCL (%d elements), bnn_amd.optim.Adam" % (mode, dp.bucket_numel()),
           "graph_fallback_reason": why,
           "value": x.shape[0] * world * steps / elapsed, "unit": "samples/s", "steps": steps,
           "ms_per_step": elapsed / steps * 1e3, "settle": st, "timed_attempts": attempts,
           "bucket_bytes": dp.bucket_numel() * 4, "collective": dp.describe_collective()}
    step_stats(res, per_step)
    # whole-step roofline: forward 4 B sum(IO) + backward 8 B sum(IO) algorithmic flop (dX and dW of both moment products; the
    # first layer has no dX: 12 B sum(IO) - 4 B I1 O1) against the 16-bit MFMA peak -- the step also contains ~0.35 ms of
    # HBM-bound passes (output gradients, weight-pass backward, operand transposes, Adam), so this is a floor on how far the
    # step is from the matrix peak, not a kernel efficiency
    sum_io = sum(DIMS[i] * DIMS[i + 1] for i in range(3))
    flop = (12.0 * sum_io - 4.0 * DIMS[0] * DIMS[1]) * x.shape[0]
    ach = flop / (elapsed / steps) / 1e12
    res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK_TFLOPS,
                       "gflop_per_step_algorithmic": flop / 1e9,
                       "note": "ALGORITHMIC flop of forward + backward of the three layers per rank / step time; forward GEMMs in the "
                               "row-scaled fp16 format (2 executed products per algorithmic one), backward products bf16x3 (3 per one)",
                       "kernel_summary": "profiles/r03_train_step_graph_planar_kernel_summary.txt"}
    return res


DTYPE_OF = {"fp16x3f": "f16 (row-scaled fp16 hi + lo split of the f32 operands: 3 mean + 1 variance MFMA products, f32 accumulate)",
            "fp16x3": "f16 (row-scaled fp16 hi + lo split of the f32 operands: 3 + 3 MFMA products, f32 accumulate)",
            "bf16x3": "bf16 (hi + lo split of the f32 operands: 3 + 3 MFMA products, f32 accumulate)", "fp32": "f32"}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(base_env, rank, world, port):
    """Environment of child rank ``rank``: what torch.distributed.run would have set (one rank per GPU of ONE node)."""
    env = dict(base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LBBNN_BENCH_CHILD": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    return env


def spawn_ranks(n, argv=None, script=None, env=None, timeout=None, popen=None):
    """`python bench.py --gpus N` WITHOUT a launcher: this process starts the N ranks itself -- N fresh children of the same
    script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set -- and never touches a GPU (no HIP call has been made when this
    runs, and none is made here: a process that has initialised the GPU must not be replaced or forked from on this pool).
    Rank 0's stdout (the ONE JSON line) is relayed to this process's stdout as it comes; every child's stderr goes to this
    process's stderr.  Returns the worst child return code (a rank killed by a signal counts as 128 + signal); when one rank
    fails the others are terminated -- they would otherwise wait in a collective until its timeout."""
    import subprocess
    popen = popen or subprocess.Popen
    script = script or os.path.abspath(__file__)
    argv = list(sys.argv[1:] if argv is None else argv)
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    base = dict(os.environ if env is None else env)
    procs = []
    for r in range(n):
        procs.append(popen([sys.executable, script] + argv, env=rank_env(base, r, n, port),
                           stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    t_end = None if timeout is None else time.monotonic() + timeout
    worst, done = 0, [False] * n
    out0 = procs[0].stdout

    def _rc(p):
        rc = p.returncode
        return 128 - rc if rc < 0 else rc

    import threading
    lines = []

    def _relay():
        # the JSON line goes to stdout; anything else a library wrote to rank 0's stdout (gloo announces its connections
        # there) goes to stderr, so that this process's stdout is the ONE line of the contract
        for raw in iter(out0.readline, b""):
            line = raw.decode(errors="replace")
            lines.append(line)
            dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            dst.write(line)
            dst.flush()

    th = threading.Thread(target=_relay, daemon=True)
    if out0 is not None:
        th.start()
    failed = False
    while not all(done):
        for i, p in enumerate(procs):
            if not done[i] and p.poll() is not None:
                done[i] = True
                worst = max(worst, _rc(p))
                failed = failed or p.returncode != 0
        if failed or (t_end is not None and time.monotonic() > t_end):
            for i, p in enumerate(procs):               # exact PIDs this process started, nothing by pattern
                if not done[i]:
                    p.terminate()
            for i, p in enumerate(procs):
                if not done[i]:
                    try:
                        p.wait(timeout=20)
                    except Exception:                    # noqa: BLE001
                        p.kill()
                        p.wait()
                    done[i] = True
                    worst = max(worst, _rc(p) or 1)
            if not failed:
                worst = max(worst, 124)
            break
        time.sleep(0.05)
    if out0 is not None:
        th.join(timeout=10)
    return worst


def ranks_report(world, rank, backend, dev_index):
    """Who actually ran: world size and backend as torch.distributed reports them, and every rank's device (index, name,
    gcn arch, PCI bus id, uuid) all-gathered -- N DISTINCT bus ids / uuids are what shows that N GPUs took part."""
    pr = torch.cuda.get_device_properties(dev_index)
    mine = {"rank": rank, "pid": os.getpid(), "device_index": dev_index, "name": pr.name,
            "arch": getattr(pr, "gcnArchName", None),
            "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
            "uuid": str(getattr(pr, "uuid", ""))}
    if world > 1:
        import torch.distributed as dist
        devs = [None] * world
        dist.all_gather_object(devs, mine)
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)                       # one collective on the data-path backend: counts the ranks it reached
        return {"world": dist.get_world_size(), "backend": dist.get_backend(), "devices": devs,
                "distinct_devices": len({d["uuid"] or d["pci"] for d in devs}),
                "all_reduce_of_ones": float(probe.item()), "launcher": os.environ.get("LBBNN_BENCH_CHILD") and "self-spawned" or "external"}
    return {"world": 1, "backend": None, "devices": [mine], "distinct_devices": 1, "launcher": "single process"}


def pick_backend(world, ndev):
    """nccl (= RCCL) with one rank per GPU is the form the metric is defined on.  LBBNN_BENCH_BACKEND overrides; without
    it a box with FEWER devices than ranks falls back to gloo with ranks sharing cards (a rehearsal of the N-rank flow --
    RCCL refuses two ranks on one device); the JSON line says which (`ranks.backend`, `ranks.devices`)."""
    forced = os.environ.get("LBBNN_BENCH_BACKEND")
    if forced:
        return forced
    return "nccl" if (world <= 1 or ndev >= world) else "gloo"


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves (before anything in this process touches the GPU)
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: one rank per GPU (run it without a launcher and "
                         "it starts its own ranks)" % (args.gpus, world))
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    backend = pick_backend(world, ndev)
    dev_index = local_rank if (backend == "nccl" or local_rank < ndev) else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    want_train = args.train if args.train is not None else world > 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            if want_train and rank == 0:
                # RCCL reports the algorithm / protocol it picks per collective at INFO level: kept in a file, parsed below
                os.environ.setdefault("NCCL_DEBUG", "INFO")
                os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,COLL")
                os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/lbbnn_rccl_%d.log" % os.getpid())
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    ranks = ranks_report(world, rank, backend, dev_index)

    import bnn_amd
    from bnn_amd import ops

    torch.manual_seed(0)          # same parameters and same z-noise stream on every rank
    net = bnn_amd.mnf.BayesianNetwork(DIMS, T_FLOWS, z_flow_type="Planar", r_flow_type="Planar").to(dev)
    net.train()
    net.set_row_offset(rank * args.batch)     # eps counters are global row indices: rank-distinct draws
    B = args.batch
    x = torch.rand(B, 1, 28, 28, device=dev, generator=torch.Generator(device=dev).manual_seed(1 + rank))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    legs = {args.precision: forward_leg(args, bnn_amd, ops, net, x, sync, args.precision, world)}
    if args.precision != "fp32" and not args.no_secondary:
        legs["fp32"] = forward_leg(args, bnn_amd, ops, net, x, sync, "fp32", world)
        if args.precision != "fp16x3" and world == 1:
            legs["fp16x3"] = forward_leg(args, bnn_amd, ops, net, x, sync, "fp16x3", world)
    if args.precision != "fp32" and not args.no_reduced and world == 1:
        legs["bf16"] = forward_leg(args, bnn_amd, ops, net, x, sync, "bf16", world)
    net.set_precision(args.precision)
    strong = None
    if world > 1 and B % world == 0 and not args.no_secondary:
        # SURVEY.md 8(d) asks for both scalings: the SAME 4096 rows split N ways (rank r takes rows [r B/N, (r+1) B/N)), no
        # collective either -- reported beside the weak-scaling headline, never instead of it
        Bs = B // world
        net.set_row_offset(rank * Bs)
        strong = forward_leg(args, bnn_amd, ops, net, x[:Bs], sync, args.precision, world)
        net.set_row_offset(rank * B)
    def emit(train):
        """Rank 0: build and print the ONE JSON line (everything but `train` was measured before this is called)."""

        if rank == 0:
            sum_io = sum(DIMS[i] * DIMS[i + 1] for i in range(3))
            main_leg = legs[args.precision]
            elapsed = main_leg["elapsed"]
            total = B * world * args.steps
            sampled_in = "separate eager pass after the timed region (every GEMM launch bracketed by HIP events)"
            res = {
                "metric": "ELBO forward samples/sec, 784-1200^2-10 MNF MLP, batch 4096 per GPU",
                "value": total / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": DTYPE_OF[args.precision], "precision": args.precision,
                "data": "synthetic", "hip_graph": main_leg["launch"] == "graph",
                "launch": {"plan": "recorded launch plan (bnn_amd.graphs.LaunchPlan: the forward's C calls -- flows + weight pass, GEMM 1 with the KL "
                                   "finalize riding, GEMM 2 with the head folded in + its finalize: 5 kernels -- replayed from a list)",
                           "graph": "one HIP-graph replay per step", "eager": "5 launches per step from Python"}[main_leg["launch"]],
                "launch_fallback_reason": main_leg["launch_fallback_reason"],
                "config": {"workload": "LBBNN-GP-MF-MNF 784-1200-1200-10, 2 planar flows/layer, batch %d per GPU, "
                                       "training-mode ELBO forward (activations + log_softmax + kl), in-kernel Philox noise" % B,
                           "global_batch": B * world, "parallelism": "dp%d (replicated parameters, no forward collective)" % world},
                "gflop_per_step_algorithmic": 4.0 * B * sum_io / 1e9,
                "settle": main_leg["settle"], "timed_attempts": main_leg["attempts"],
                "ranks": ranks,
            }
            step_stats(res, main_leg["per_step"])
            if main_leg["events"]:
                roof, why = roofline_object(main_leg["events"], args.precision, res["ms_per_step"], sampled_in)
                if roof is not None:
                    res["roofline"] = roof
                else:
                    res["roofline_invalid"] = why
            if "fp32" in legs and args.precision != "fp32":
                leg = legs["fp32"]
                sec = {"dtype": "f32", "what": "the same step with the exact-fp32 MFMA GEMM (reference precision), same process",
                       "value": total / leg["elapsed"], "unit": "samples/s", "steps": args.steps,
                       "ms_per_step": leg["elapsed"] / args.steps * 1e3, "settle": leg["settle"],
                       "timed_attempts": leg["attempts"]}
                step_stats(sec, leg["per_step"])
                if leg["events"]:
                    roof, why = roofline_object(leg["events"], "fp32", sec["ms_per_step"], sampled_in)
                    if roof is not None:
                        sec["roofline"] = roof
                    else:
                        sec["roofline_invalid"] = why
                res["secondary"] = sec
            if "fp16x3" in legs and args.precision != "fp16x3":
                leg = legs["fp16x3"]
                st3 = {"dtype": DTYPE_OF["fp16x3"], "precision": "fp16x3",
                       "what": "the same step with 3 + 3 products (2-3e-8 of max|out| against fp64: tighter than an fp32-accumulate "
                               "torch.mm), same process",
                       "value": total / leg["elapsed"], "unit": "samples/s", "steps": args.steps,
                       "ms_per_step": leg["elapsed"] / args.steps * 1e3, "settle": leg["settle"], "timed_attempts": leg["attempts"]}
                step_stats(st3, leg["per_step"])
                if leg["events"]:
                    roof, why = roofline_object(leg["events"], "fp16x3", st3["ms_per_step"], sampled_in)
                    if roof is not None:
                        st3["roofline"] = roof
                    else:
                        st3["roofline_invalid"] = why
                res["secondary_strict_fp16x3"] = st3
            if "bf16" in legs:
                leg = legs["bf16"]
                red = {"dtype": "bf16 (ONE product per moment)",
                       "what": "REDUCED PRECISION, outside the 1e-4 contract (2e-3 relative on the mean GEMM): the plain bf16 MFMA "
                               "arithmetic BASELINE configs[1] names, same step, same process; not comparable with `value`",
                       "value": total / leg["elapsed"], "unit": "samples/s", "steps": args.steps,
                       "ms_per_step": leg["elapsed"] / args.steps * 1e3, "settle": leg["settle"],
                       "timed_attempts": leg["attempts"]}
                step_stats(red, leg["per_step"])
                if leg["events"]:
                    roof, why = roofline_object(leg["events"], "bf16", red["ms_per_step"], sampled_in)
                    if roof is not None:
                        roof["executed_mfma_tflops"] = roof["achieved"]
                        roof["note"] = "achieved = ALGORITHMIC 4*B*I*O flop per launch / HIP-event time; one bf16 product per algorithmic product"
                        roof["traffic"], roof["traffic_source"] = None, None
                        red["roofline"] = roof
                    else:
                        red["roofline_invalid"] = why
                res["secondary_reduced_bf16"] = red
            if strong is not None:
                st = {"what": "STRONG scaling: the headline's global batch of %d rows split over the %d ranks (%d rows each), same step"
                              % (B, world, B // world),
                      "scaling": "strong", "global_batch": B, "value": B * args.steps / strong["elapsed"], "unit": "samples/s",
                      "steps": args.steps, "ms_per_step": strong["elapsed"] / args.steps * 1e3, "settle": strong["settle"],
                      "timed_attempts": strong["attempts"]}
                step_stats(st, strong["per_step"])
                res["secondary_strong"] = st
            if train is not None:
                if world > 1 and backend == "nccl":
                    train["rccl"] = rccl_log_summary(os.environ.get("NCCL_DEBUG_FILE"))
                res["secondary_train"] = train
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(B, args.cpu_seconds)
                res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            print(json.dumps(res), flush=True)

    # The training leg is the only part of the run with a data-path collective (the gradient all-reduce): it runs LAST, behind a
    # watchdog, so that neither an exception on one rank nor a collective that never returns can take the headline with it --
    # after --train-timeout seconds rank 0 prints the line with the reason in place of the leg and every rank leaves with 0.
    train = None
    if want_train:
        import threading

        def on_timeout():
            if rank == 0:
                emit({"error": "the training leg did not finish within %d s (rank 0 gave up; headline legs unaffected)" % args.train_timeout})
            os._exit(0)

        guard = threading.Timer(args.train_timeout, on_timeout)
        guard.daemon = True
        guard.start()
        try:
            train = train_leg(args, bnn_amd, net, x, sync, world, rank)
        except Exception as e:                                   # noqa: BLE001 -- reported in the line; the other ranks' watchdogs end them
            guard.cancel()
            if rank == 0:
                emit({"error": "%s: %s" % (type(e).__name__, str(e)[:300])})
            os._exit(0)
        guard.cancel()
    emit(train)
    if world > 1:
        dist.destroy_process_group()


def rccl_log_summary(path):
    """What RCCL said about itself on rank 0 (NCCL_DEBUG=INFO): version, transport lines, and the algorithm / protocol
    of the AllReduce calls it logged.  Best effort: None when the log is absent."""
    if not path:
        return None
    try:
        lines = open(path, errors="replace").read().splitlines()
    except OSError:
        return None
    out = {"ranks_seen": None, "version": None, "allreduce": [], "transport": []}
    import re
    for ln in lines:
        if "RCCL version" in ln or "NCCL version" in ln:
            out["version"] = ln.split("INFO")[-1].strip()[:120]
        m = re.search(r"nranks (\d+)", ln)
        if m:
            out["ranks_seen"] = int(m.group(1))
        if "AllReduce" in ln and ("algo" in ln.lower() or "proto" in ln.lower()) and len(out["allreduce"]) < 4:
            out["allreduce"].append(ln.split("INFO")[-1].strip()[:200])
        if (" via " in ln or "Connected all" in ln) and len(out["transport"]) < 4:
            out["transport"].append(ln.split("INFO")[-1].strip()[:160])
    return out


if __name__ == "__main__":
    main()
