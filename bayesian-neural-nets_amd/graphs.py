"""HIP-graph capture of launch-bound loops (one process per GPU; graphs instead of a tracing compiler).

``make_graphed_train_step`` captures ONE full training step -- forward, backward (both HIP kernels) and the
optimizer update -- into a HIP graph and returns a callable that replays it on new data.  The eager step of the
headline net issues ~85 launches and is host-bound at ~2.3 ms; replayed from a graph it runs at GPU speed (~1.1 ms;
DESIGN.md section 9).  Noise stays fresh across replays because the Philox {seed, offset} pair lives in device memory and
is advanced by a kernel inside the graph.
"""
import torch


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
    # the warm-up ran on a side stream, the capture runs on torch's capture stream: the AccumulateGrad nodes created during
    # warm-up therefore sit on another (non-default) stream than the captured backward -- intended, as in torch's own
    # whole-network capture recipe; newer torch versions warn about it
    _quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if _quiet is not None:
        _quiet(False)
    with torch.cuda.graph(graph):
        static_loss = loss_fn(net, static_x, static_y)
        with ov():
            static_loss.backward()
        optimizer.step()

    if _quiet is not None:
        _quiet(True)

    def step(x, y):
        static_x.copy_(x)
        static_y.copy_(y)
        graph.replay()
        return static_loss

    step.graph = graph
    return step
