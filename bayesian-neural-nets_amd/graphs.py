"""HIP-graph capture of launch-bound loops (one process per GPU; graphs instead of a tracing compiler).

``make_graphed_train_step`` captures ONE full training step -- forward, backward (both HIP kernels) and the
optimizer update -- into a HIP graph and returns a callable that replays it on new data.  The eager step of the
headline net issues ~85 launches and is host-bound at ~2.3 ms; replayed from a graph it runs at GPU speed (~1.1 ms;
DESIGN.md section 9).  Noise stays fresh across replays because the Philox {seed, offset} pair lives in device memory and
is advanced by a kernel inside the graph.
"""
import weakref

import torch


# ---- capture guard -------------------------------------------------------------------------------------------------
# Every autograd node this package creates (layers._BayesLinearFn, base._BaseFn, vd._VDFn) carries a _NodeMark; the
# mark lives exactly as long as the node, i.e. as long as something (a loss, an output) still holds the graph the
# node is part of -- and with it the AccumulateGrad nodes of the layer's parameters, which remember the stream they
# were created on.  A stream capture that runs a backward pass through such an old AccumulateGrad node makes autograd
# synchronise the capture stream with that node's stream; HIP does not survive it (measured in round 2: a segmentation
# fault inside capture_end, gpurun_out/r02/dpgraph.log).  The factories below therefore refuse to start while a graph
# through the network is alive, instead of crashing the process.
_MARKS = weakref.WeakSet()


class _NodeMark:
    __slots__ = ("owner", "stream", "capturing", "__weakref__")


def mark_autograd_node(ctx, owner):
    """Called from the forward of every autograd.Function of this package: ``ctx`` is the graph node being built."""
    m = _NodeMark()
    m.owner = weakref.ref(owner)
    m.capturing = torch.cuda.is_current_stream_capturing()
    m.stream = torch.cuda.current_stream(owner_device(owner)).cuda_stream
    ctx._lbbnn_mark = m
    _MARKS.add(m)


def owner_device(mod):
    p = next(iter(mod.parameters()), None)
    return p.device if (p is not None and p.is_cuda) else None


def live_autograd_nodes(net):
    """[(module class name, stream handle)] for the autograd nodes through ``net``'s layers that are still alive."""
    mods = {id(m) for m in net.modules()}
    out = []
    for m in list(_MARKS):
        o = m.owner()
        if o is not None and id(o) in mods and not m.capturing:
            out.append((type(o).__name__, m.stream))
    return out


def release_module_graph_refs(net):
    """The modules themselves hold results of the last forward as attributes, as the reference's do (``layer.kl``,
    ``layer.log_prior`` ...: LBBNN-GP-MF-MNF.py:237, LBBNN-GP-MF.py:246-251) -- tensors with a grad_fn, i.e. references into
    the last autograd graph.  They are replaced by detached tensors (same values): nothing can call backward through an
    attribute of a finished step, and they must not keep an eager graph alive under a capture.  Returns how many."""
    n = 0
    for mod in net.modules():
        for name, val in list(vars(mod).items()):
            if isinstance(val, torch.Tensor) and val.grad_fn is not None:
                setattr(mod, name, val.detach())
                n += 1
    return n


def assert_no_live_graph(net, who):
    """Raise (instead of letting the capture crash the process) when an autograd graph through ``net`` is still alive.
    The network's own references into its last graph (``layer.kl`` ...) are detached first; what is left is held by the
    caller (a loss, an output)."""
    import gc
    if live_autograd_nodes(net):
        release_module_graph_refs(net)
    if live_autograd_nodes(net):
        gc.collect()                                   # a graph kept alive by a reference cycle only is not the caller's fault
    live = live_autograd_nodes(net)
    if live:
        on_default = sum(1 for _, s in live if s == 0)
        raise RuntimeError(
            "bnn_amd.%s: %d autograd node(s) of an earlier forward through this network are still alive (%d created on the "
            "default stream; layers: %s).  Their AccumulateGrad nodes would be synchronised inside the stream capture, which "
            "HIP does not survive (segmentation fault in capture_end).  Drop the tensors that hold that graph (`del loss, "
            "out`) or build the graphed step before the first eager training step."
            % (who, len(live), on_default, ", ".join(sorted({n for n, _ in live}))))


def capture(graph, **kw):
    """``torch.cuda.graph(graph, ...)`` with a capture mode that survives a live communicator.  A process that has
    initialised RCCL runs torch's ProcessGroupNCCL watchdog thread, which polls HIP events of outstanding collectives; under
    the default GLOBAL capture mode a HIP call from ANY thread that is not legal inside a capture invalidates the capture
    and raises in that thread -- in the watchdog that is an uncaught exception, the process aborts (seen once in round 3 on
    a single-rank nccl group: rc -6 before the first replay; a race, most runs pass).  THREAD_LOCAL restricts the check to
    the capturing thread, which is the only one that issues work into the capture."""
    if "capture_error_mode" not in kw:
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                kw["capture_error_mode"] = "thread_local"
        except Exception:                                # noqa: BLE001
            pass
    return torch.cuda.graph(graph, **kw)


def make_graphed_train_step(net, optimizer, loss_fn, example_x, example_y, warmup: int = 3, overlap_vector_backward=None):
    """loss_fn(net, x, y) -> scalar loss.  The optimizer must be capture-safe: ``bnn_amd.optim.Adam`` (device-side
    step counter) or ``torch.optim.Adam(capturable=True)``.  The warm-up steps run eagerly first, so optimizer state is
    allocated outside the capture.
    ``overlap_vector_backward``: the vector-sized backward chains of the MNF layers are deferred
    (``layers.vector_backward_overlap``): each layer's backward only files its chain, and all layers' chains are issued
    in the same launches after the backward pass, before the optimizer step (planar: one launch instead of three; RNVP /
    MNF type: 14 launches instead of 42).  Default (None): on, unless env LBBNN_BWD_OVERLAP=0.  Measured on the headline
    net: planar 0.95 -> 0.88 ms, RNVP 1.21 -> 1.08 ms.
    Returns step(x, y) -> loss tensor (a static buffer, overwritten by the next replay)."""
    import contextlib
    import os
    from . import layers
    if overlap_vector_backward is None:
        env = os.environ.get("LBBNN_BWD_OVERLAP")
        overlap_vector_backward = env != "0"
    ov = layers.vector_backward_overlap if overlap_vector_backward else contextlib.nullcontext
    assert_no_live_graph(net, "graphs.make_graphed_train_step")
    dev = example_x.device
    static_x, static_y = example_x.clone(), example_y.clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            optimizer.zero_grad(set_to_none=True)
            with ov():
                loss_fn(net, static_x, static_y).backward()
            optimizer.step()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    optimizer.zero_grad(set_to_none=True)
    one = torch.ones((), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    # the warm-up ran on a side stream, the capture runs on torch's capture stream: the AccumulateGrad nodes created during
    # warm-up therefore sit on another (non-default) stream than the captured backward -- intended, as in torch's own
    # whole-network capture recipe; newer torch versions warn about it
    _quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if _quiet is not None:
        _quiet(False)
    with capture(graph):
        static_loss = loss_fn(net, static_x, static_y)
        with ov():
            # the root gradient as a tensor made once, outside the capture: loss.backward() alone fills a fresh ones_like(loss)
            # on every replay (a ~4.7 us launch between the loss kernel and its backward)
            static_loss.backward(one if (static_loss.dim() == 0 and static_loss.dtype == torch.float32) else None)
        optimizer.step()

    if _quiet is not None:
        _quiet(True)

    def step(x, y):
        # a batch that already lies in the graph's input buffers (step.inputs: a loader that writes its host-to-device copy
        # straight into them) costs no device-to-device copy: 2 x ~4.7 us ahead of every replay otherwise
        if x.data_ptr() != static_x.data_ptr():
            static_x.copy_(x)
        if y.data_ptr() != static_y.data_ptr():
            static_y.copy_(y)
        graph.replay()
        return static_loss

    step.graph = graph
    step.inputs = (static_x, static_y)
    step._root_grad = one                          # read by every replay: lives as long as the step
    return step


class LaunchPlan:
    """The no-grad forward of a network as a RECORDED LIST OF C CALLS, replayed without the Python in between.

    An eager forward of the headline net costs the host ~100 us (descriptor structs filled from ~120 attribute reads, three
    wrapper layers per launch) for 4 C calls that enqueue 5 kernels; the GPU needs ~165 us for them.  ``LaunchPlan(net, x,
    sample=True)`` runs ONE forward with ``_lib.RECORD`` on -- every call that goes through the C ABI is kept with the
    arguments ctypes was given -- into buffers of its own, and ``plan()`` makes the same calls again: same kernels, same
    arguments, same stream, fresh noise (the Philox offset lives on the device).  Unlike a HIP-graph replay there is nothing
    between two forwards but the stream's own order (a replayed graph costs ~6 us per launch on this stack), and unlike the
    eager call the host needs ~30 us.

    Like a captured graph the plan is frozen: same input BUFFER (copy new data into ``plan.x``), same shapes, same mode
    (train / eval, sample), same precision, same explicit noise settings, same stream; parameters may change IN PLACE
    (optimizer steps), not be re-assigned.  Outputs are the plan's static buffers, overwritten by the next call:
    ``plan()`` -> (log-probabilities (B, classes), kl or None); ``net.l1.kl`` ... and ``net.kl()`` read the same buffers.
    """

    def __init__(self, net, x, sample=True):
        from . import _lib
        if not x.is_cuda:
            raise RuntimeError("bnn_amd.graphs.LaunchPlan needs a HIP tensor")
        if not hasattr(net, "_forward_streams") or not all(l._fusable() for l in net._layers()):
            raise RuntimeError("bnn_amd.graphs.LaunchPlan: this network has no fused no-grad forward to record")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("bnn_amd.graphs.LaunchPlan cannot be recorded inside a graph capture")
        self.net, self.x, self.sample = net, x, bool(sample)
        self.stream = torch.cuda.current_stream(x.device).cuda_stream
        x2 = x.view(-1, net.dims[0]).float()
        if x2.data_ptr() != x.data_ptr():
            raise RuntimeError("bnn_amd.graphs.LaunchPlan: the input must be a contiguous float32 buffer (it is re-read on every call)")
        with torch.no_grad():
            net(x, sample=sample)                       # workspaces, kernels and the RNG state exist before the recording
            rec = {}
            net._plan_rec = rec
            _lib.RECORD = calls = []
            try:
                self.out = net._forward_streams(x2, sample)
            finally:
                _lib.RECORD = None
                net._plan_rec = None
        self._rec = rec                                 # kls, the layers' outputs, and whatever the forward had to keep alive
        self._calls = [(name, fn, args) for (name, fn, args) in calls]
        self.kl = net._kl_total if net._kl_total is not None else None
        self._check = _lib.check

    def __call__(self):
        if torch.cuda.current_stream(self.x.device).cuda_stream != self.stream:
            raise RuntimeError("bnn_amd.graphs.LaunchPlan: recorded on another stream")
        check = self._check
        for name, fn, args in self._calls:
            rc = fn(*args)
            if rc:
                check(rc, name)
        return self.out, self.kl

    def __len__(self):
        return len(self._calls)
