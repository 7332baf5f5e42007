"""GPU parity tests proper: the HIP path (through the C ABI) vs the CPU oracle and the golden
vectors.  Tolerance: 1e-4 relative fp32 (BASELINE.json north_star), written per assertion;
most checks hold far tighter and say so."""
import math

import numpy as np
import pytest
import torch

from conftest import elementwise_violation, rel_err, sub
from oracle import lbbnn_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4          # the contract
TIGHT = 5e-6        # what the fp32-exact MFMA path actually achieves on these sizes


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def bnn():
    import bnn_amd
    return bnn_amd


def _load_layer(layer, p, dev):
    layer.load_state_dict({k: v for k, v in p.items()})
    return layer.to(dev)


# --------------------------------------------------------------------------- K1 weight pass
@pytest.mark.parametrize("O,I", [(4, 6), (17, 33), (10, 784), (1200, 784), (10, 1200), (33, 1201)])
def test_weight_pass_vs_oracle(bnn, dev, O, I):
    ops = bnn.ops
    g = torch.Generator().manual_seed(O * 1000 + I)
    p = orc.init_mnf_params(I, O, g)
    zf = 1 + 0.1 * torch.randn(I, generator=g)
    zk = 1 + 0.1 * torch.randn(I, generator=g)
    d = {k: v.to(dev) for k, v in p.items()}
    ld = ops.operand_ld(I)
    e_w = torch.full((O, ld), float("nan"), device=dev)
    var_w = torch.full((O, ld), float("nan"), device=dev)
    kl_rows, act_mu, act_var, bias_var = (torch.empty(O, device=dev) for _ in range(4))
    ops.weight_pass(d["weight_mu"], d["weight_rho"], d["lambdal"], z_fwd=zf.to(dev), z_kl=zk.to(dev),
                    r0_c=d["r0_c"], bias_rho=d["bias_rho"], priors=bnn.Priors(), e_w=e_w, var_w=var_w,
                    kl_rows=kl_rows, act_mu=act_mu, act_var=act_var, bias_var=bias_var)
    alpha = orc.alpha_of(p["lambdal"]); sigma = orc.sigma_of(p["weight_rho"])
    assert rel_err(e_w[:, :I], p["weight_mu"] * alpha * zf) < TIGHT
    assert rel_err(var_w[:, :I], sigma ** 2 * alpha ** 2) < TIGHT
    assert (e_w[:, I:] == 0).all() and (var_w[:, I:] == 0).all()          # zero-filled operand tail
    ref_rows = orc.kl_weight_elem(p["weight_mu"] * zk, sigma, alpha, orc.Priors()).sum(1)
    assert rel_err(kl_rows, ref_rows) < TIGHT
    assert rel_err(act_mu, p["r0_c"] @ (zk * p["weight_mu"] * alpha).T) < 2e-5
    assert rel_err(act_var, p["r0_c"] ** 2 @ (sigma ** 2 * alpha ** 2).T) < TIGHT
    assert rel_err(bias_var, orc.sigma_of(p["bias_rho"]) ** 2) < TIGHT


@pytest.mark.parametrize("O,I", [(64, 1200), (9, 33)])
def test_weight_pass_wide_parameter_ranges(bnn, dev, O, I):
    """K1's arithmetic over the ranges TRAINED parameters reach, not only the reference's initial ones (rho in [-5, -4],
    lambdal in [0, 1]): rho in [-14, 4] crosses both softplus branches (series below exp(rho) = 0.04, log1p above), lambdal
    in [-14, 14] takes alpha from 1e-6 to 1 - 1e-6, |mu| up to 3.  Every operand ELEMENT within 3e-6 relative of the fp64
    oracle (the raw hardware exp2 / rcp forms of round 3 are held to the bar of the library forms they replaced), the KL row
    sums within 1e-5; row kernel (I % 4 == 0) and generic kernel."""
    ops = bnn.ops
    g = torch.Generator().manual_seed(O + I)
    mu = (3 * (2 * torch.rand(O, I, generator=g) - 1)).double()
    rho = (-14 + 18 * torch.rand(O, I, generator=g)).double()
    lam = (-14 + 28 * torch.rand(O, I, generator=g)).double()
    zf = (1 + 0.3 * torch.randn(I, generator=g)).double()
    zk = (1 + 0.3 * torch.randn(I, generator=g)).double()
    rc = torch.randn(I, generator=g).double()
    f = lambda t: t.float().to(dev)
    ld = ops.operand_ld(I)
    e_w, var_w = torch.empty(O, ld, device=dev), torch.empty(O, ld, device=dev)
    kl_rows, act_mu, act_var = (torch.empty(O, device=dev) for _ in range(3))
    ops.weight_pass(f(mu), f(rho), f(lam), z_fwd=f(zf), z_kl=f(zk), r0_c=f(rc), priors=bnn.Priors(), e_w=e_w, var_w=var_w,
                    kl_rows=kl_rows, act_mu=act_mu, act_var=act_var)
    mu, rho, lam, zf, zk, rc = (t.float().double() for t in (mu, rho, lam, zf, zk, rc))      # the fp32 values the kernel saw
    alpha, sigma = torch.sigmoid(lam), torch.nn.functional.softplus(rho)
    ref_e, ref_v = mu * alpha * zf, sigma ** 2 * alpha ** 2
    assert float(((e_w[:, :I].cpu().double() - ref_e).abs() / ref_e.abs().clamp_min(1e-30)).max()) < 3e-6
    assert float(((var_w[:, :I].cpu().double() - ref_v).abs() / ref_v.abs().clamp_min(1e-30)).max()) < 3e-6
    ref_rows = orc.kl_weight_elem(mu * zk, sigma, alpha, orc.Priors()).sum(1)
    assert torch.isfinite(kl_rows).all() and rel_err(kl_rows, ref_rows) < 1e-5
    assert rel_err(act_mu, rc @ (zk * mu * alpha).T) < 2e-5
    assert rel_err(act_var, rc ** 2 @ ref_v.T) < 1e-5


def test_weight_pass_deterministic(bnn, dev):
    ops = bnn.ops
    g = torch.Generator().manual_seed(5)
    p = {k: v.to(dev) for k, v in orc.init_lrt_params(784, 1200, g).items()}
    outs = []
    for _ in range(2):
        kl_rows = torch.empty(1200, device=dev)
        ops.weight_pass(p["weight_mu"], p["weight_rho"], p["lambdal"], priors=bnn.Priors(), kl_rows=kl_rows)
        outs.append(kl_rows.clone())
    assert torch.equal(outs[0], outs[1])


# --------------------------------------------------------------------------- K2 GEMM
def _gemm_case(bnn, dev, B, I, O, relu=False, mean_only=False, seed=0):
    ops = bnn.ops
    g = torch.Generator().manual_seed(seed + B + 7 * I + 13 * O)
    x = torch.rand(B, I, generator=g)
    ew = 0.2 * (torch.rand(O, I, generator=g) - 0.5)
    vw = 1e-4 * torch.rand(O, I, generator=g)
    bm = 0.2 * (torch.rand(O, generator=g) - 0.5)
    bv = 1e-4 * torch.rand(O, generator=g)
    eps = torch.randn(B, O, generator=g)
    ld = ops.operand_ld(I)
    ewp = torch.zeros(O, ld); ewp[:, :I] = ew
    vwp = torch.zeros(O, ld); vwp[:, :I] = vw
    out = ops.lrt_gemm(x.to(dev), ewp.to(dev), vwp.to(dev), I=I, O=O, bias_mean=bm.to(dev), bias_var=bv.to(dev),
                       eps=eps.to(dev), relu=relu, mean_only=mean_only)
    x64, ew64, vw64 = x.double(), ew.double(), vw.double()
    ref = x64 @ ew64.T + bm.double()
    if not mean_only:
        ref = ref + torch.sqrt((x64 ** 2) @ vw64.T + bv.double()) * eps.double()
    if relu:
        ref = torch.relu(ref)
    return out, ref


@pytest.mark.parametrize("B,I,O", [(3, 6, 4), (5, 33, 17), (8, 784, 10), (100, 784, 400), (128, 400, 400),
                                   (1024, 784, 400), (257, 1200, 1200), (64, 1200, 10), (130, 50, 81), (1, 7, 1)])
def test_gemm_vs_fp64(bnn, dev, B, I, O):
    out, ref = _gemm_case(bnn, dev, B, I, O)
    assert rel_err(out, ref) < TIGHT
    out, ref = _gemm_case(bnn, dev, B, I, O, relu=True)
    assert rel_err(out, ref) < TIGHT
    out, ref = _gemm_case(bnn, dev, B, I, O, mean_only=True)
    assert rel_err(out, ref) < TIGHT


def test_gemm_full_size_headline(bnn, dev):
    """BASELINE configs[2] layer shapes at B=4096, against fp64 on the host."""
    for (I, O) in [(784, 1200), (1200, 1200), (1200, 10)]:
        out, ref = _gemm_case(bnn, dev, 4096, I, O)
        assert rel_err(out, ref) < TIGHT


def test_gemm_identity_asymmetric(bnn, dev):
    """A = I with an asymmetric B catches a transposed accumulator map (cdna guide section 3)."""
    ops = bnn.ops
    I = O = 32
    B = 48
    x = torch.zeros(B, I); x[:32] = torch.eye(32)
    x[32:, 3] = 2.0
    ew = (torch.arange(O)[:, None] * 100.0 + torch.arange(I)[None, :])       # asymmetric, exact in fp32
    out = ops.lrt_gemm(x.to(dev), ew.contiguous().to(dev), ew.contiguous().to(dev), I=I, O=O, mean_only=True)
    ref = x @ ew.T
    assert torch.equal(out.cpu(), ref)


def test_gemm_linearity_full_size(bnn, dev):
    """Size-independent property at the full headline size: the mean path is linear in x."""
    ops = bnn.ops
    B, I, O = 4096, 784, 1200
    g = torch.Generator().manual_seed(3)
    ld = ops.operand_ld(I)
    ew = torch.zeros(O, ld); ew[:, :I] = 0.02 * (torch.rand(O, I, generator=g) - 0.5)
    ew = ew.to(dev)
    x1 = torch.rand(B, I, generator=g).to(dev); x2 = torch.rand(B, I, generator=g).to(dev)
    f = lambda x: ops.lrt_gemm(x, ew, ew, I=I, O=O, mean_only=True)
    lhs = f(x1 + 2 * x2)
    rhs = f(x1) + 2 * f(x2)
    assert rel_err(lhs, rhs) < TIGHT


# --------------------------------------------------------------------------- K3 planar flow
@pytest.mark.parametrize("case", ["planar0", "planar1", "planar2", "planar3"])
def test_planar_flow_vs_golden(bnn, dev, golden, case):
    c = golden("flows.npz").case(case)
    I, T = [int(v) for v in c["shape"]]
    flow = bnn.flows.PropagateFlow("Planar", I, T)
    flow.load_state_dict(sub(c, "p."))
    flow = flow.to(dev)
    z, ld = flow(c["z"].to(dev))
    assert rel_err(z, c["z_out"]) < TIGHT
    assert abs(float(ld) - float(c["logdet"])) < 1e-6


# --------------------------------------------------------------------------- layers vs golden
@pytest.mark.parametrize("case", ["c0", "c1", "c2", "c3"])
def test_lrt_layer_vs_golden(bnn, dev, golden, case):
    c = golden("lrt.npz").case(case)
    B, I, O = [int(v) for v in c["shape"]]
    layer = _load_layer(bnn.lrt.BayesianLinear(I, O), sub(c, "p."), dev)
    x = c["x"].to(dev)
    with torch.no_grad():
        layer.train()
        layer.noise = {"eps_out": c["eps"].to(dev)}
        out = layer(x, sample=True)
        assert rel_err(out, c["out_train"]) < TIGHT
        assert rel_err(layer.kl, c["kl"]) < TIGHT
        layer.eval()
        layer.noise = {"eps_out": c["eps_eval"].to(dev)}
        out = layer(x, sample=True)
        assert rel_err(out, c["out_eval_sample"]) < TIGHT
        assert layer.kl == 0
        layer.noise = None
        out = layer(x, sample=False, calculate_log_probs=True)
        assert rel_err(out, c["out_mean"]) < TIGHT
        assert rel_err(layer.kl, c["kl_logprobs"]) < TIGHT


@pytest.mark.parametrize("case", ["c0", "c1", "c2", "c3"])
def test_mnf_layer_vs_golden(bnn, dev, golden, case):
    c = golden("mnf.npz").case(case)
    B, I, O, T = [int(v) for v in c["shape"]]
    assert str(c["kind"]) == "Planar"
    layer = _load_layer(bnn.mnf.BayesianLinear(I, O, T, z_flow_type="Planar", r_flow_type="Planar"),
                        sub(c, "p."), dev)
    x = c["x"].to(dev)
    with torch.no_grad():
        layer.train()
        layer.noise = {k: c[k].to(dev) for k in ("eps_z", "eps_out", "eps_z2", "eps_act")}
        out = layer(x, sample=True)
        assert rel_err(out, c["out_train"]) < TIGHT
        assert rel_err(layer.kl, c["kl"]) < TIGHT
        layer.eval()
        layer.noise = {"eps_z": c["eps_z_eval"].to(dev)}
        out = layer(x, sample=False)
        assert rel_err(out, c["out_eval_mean"]) < TIGHT
        assert layer.kl == 0


def test_smallnets_vs_golden(bnn, dev, golden):
    c = golden("lrt.npz").case("smallnet")
    dims = [int(v) for v in c["dims"]]
    net = bnn.lrt.BayesianNetwork(dims)
    for i, l in enumerate((net.l1, net.l2, net.l3)):
        l.load_state_dict(sub(c, "l%d." % i))
    net = net.to(dev).train()
    for i, l in enumerate((net.l1, net.l2, net.l3)):
        l.noise = {"eps_out": c["eps%d" % i].to(dev)}
    with torch.no_grad():
        out = net(c["x"].to(dev), sample=True)
    assert rel_err(out, c["out"]) < TIGHT and rel_err(net.kl(), c["kl"]) < TIGHT

    c = golden("mnf.npz").case("smallnet")
    dims = [int(v) for v in c["dims"]]
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar")
    for i, l in enumerate((net.l1, net.l2, net.l3)):
        l.load_state_dict(sub(c, "l%d.p." % i))
    net = net.to(dev).train()
    for i, l in enumerate((net.l1, net.l2, net.l3)):
        l.noise = {k: c["l%d.%s" % (i, k)].to(dev) for k in ("eps_z", "eps_out", "eps_z2", "eps_act")}
    with torch.no_grad():
        out = net(c["x"].to(dev), sample=True)
    assert rel_err(out, c["out"]) < TIGHT and rel_err(net.kl(), c["kl"]) < TIGHT


# --------------------------------------------------------------------------- full-size network vs oracle
def _oracle_mnf_net(x, P, zf, rf, noises):
    """orc.mnf_network_forward, also returning the per-layer KLs."""
    h, kls = x.reshape(x.shape[0], -1), []
    for i, p in enumerate(P):
        h, k, _ = orc.mnf_forward(h, p, zf[i], rf[i], noises[i])
        kls.append(k)
        if i < len(P) - 1:
            h = torch.relu(h)
    return torch.log_softmax(h, dim=1), kls


def _to64(x, P, zf, rf, noises):
    """The oracle is dtype-preserving: its fp64 run is the truth both precisions are held against."""
    d = lambda t: ([d(u) for u in t] if isinstance(t, (list, tuple)) else t.double())
    f64 = lambda fl: orc.Flow(fl.kind, [{k: v.double() for k, v in tr.items()} for tr in fl.transforms])
    return (x.double(), [{k: v.double() for k, v in p.items()} for p in P], [f64(f) for f in zf], [f64(f) for f in rf],
            [{k: d(v) for k, v in n.items()} for n in noises])


def test_headline_network_vs_oracle(bnn, dev):
    """BASELINE configs[2]: MNF 784-1200-1200-10, 2 planar flows/layer, B=4096, injected noise; exact-fp32 MFMA path.
    Bars: global-norm relative error < 1e-4 (contract), every layer's KL < 1e-4, and ELEMENT-WISE
    |out - ref| <= 1e-6 max|ref| + 1e-4 |ref| (torch.allclose form)."""
    dims, B, T = (784, 1200, 1200, 10), 4096, 2
    torch.manual_seed(11)
    net = bnn.mnf.BayesianNetwork(dims, T, z_flow_type="Planar", r_flow_type="Planar")
    layers = [net.l1, net.l2, net.l3]
    g = torch.Generator().manual_seed(12)
    x = torch.rand(B, 784, generator=g)
    noises, P, zf, rf = [], [], [], []
    for l in layers:
        sd = {k: v.detach().clone() for k, v in l.state_dict().items()}
        P.append(sd)
        zf.append(orc.flow_from_state("z_flow", "Planar", sd, T))
        rf.append(orc.flow_from_state("r_flow", "Planar", sd, T))
        noises.append({"eps_z": torch.randn(1, l.in_features, generator=g),
                       "eps_out": torch.randn(B, l.out_features, generator=g),
                       "eps_z2": torch.randn(1, l.in_features, generator=g),
                       "eps_act": torch.randn(l.out_features, generator=g)})
    ref_out, ref_kls = _oracle_mnf_net(*_to64(x, P, zf, rf, noises))
    net = net.to(dev).train()
    for l, n in zip(layers, noises):
        l.noise = {k: v.to(dev) for k, v in n.items()}
    with torch.no_grad():
        out = net(x.to(dev), sample=True)
    assert rel_err(out, ref_out) < TOL
    assert rel_err(net.kl(), sum(ref_kls)) < TOL
    for l, k_ref in zip(layers, ref_kls):                       # per-layer KL too
        assert rel_err(l.kl, k_ref) < TOL
    v = elementwise_violation(out, ref_out, rtol=1e-4, atol_frac=1e-6)
    assert v <= 1.0, v
    assert torch.isfinite(out).all()


# --------------------------------------------------------------------------- in-kernel noise
def test_philox_moments_and_reproducibility(bnn, dev):
    ops = bnn.ops
    bnn.manual_seed(1234)
    st = ops.RngState.get(dev)
    n = ops.philox_normal(st.t, 5, 4096, 1200)
    assert abs(float(n.mean())) < 3e-3 and abs(float(n.var()) - 1) < 5e-3
    k = float(((n ** 4).mean()))                       # kurtosis 3
    assert abs(k - 3) < 0.05
    assert float(n.abs().max()) < 7.0
    n2 = ops.philox_normal(st.t, 5, 4096, 1200)
    assert torch.equal(n, n2)
    n3 = ops.philox_normal(st.t, 6, 4096, 1200)
    assert not torch.equal(n, n3)
    # row offset shifts the counter: rows [100, 200) of a 0-based draw == rows [0,100) drawn at base 100
    n4 = ops.philox_normal(st.t, 5, 100, 1200, row_base=100)
    assert torch.equal(n[100:200], n4)
    # correlation between neighbouring columns / rows
    assert abs(float((n[:, :-1] * n[:, 1:]).mean())) < 3e-3
    assert abs(float((n[:-1] * n[1:]).mean())) < 3e-3


def test_inkernel_noise_equals_injected(bnn, dev):
    """The GEMM's in-kernel Philox draw is independent of tiling: feeding the same values as an
    explicit eps gives a bit-identical output."""
    ops = bnn.ops
    bnn.manual_seed(77)
    st = ops.RngState.get(dev)
    for (B, I, O) in [(130, 64, 80), (512, 784, 1200), (64, 128, 10)]:
        g = torch.Generator().manual_seed(B)
        x = torch.rand(B, I, generator=g).to(dev)
        ld = ops.operand_ld(I)
        ew = torch.zeros(O, ld); ew[:, :I] = 0.1 * torch.randn(O, I, generator=g)
        vw = torch.zeros(O, ld); vw[:, :I] = 1e-3 * torch.rand(O, I, generator=g)
        ew, vw = ew.to(dev), vw.to(dev)
        a = ops.lrt_gemm(x, ew, vw, I=I, O=O, rng=st.t, rng_stream=9, row_offset=1000)
        eps = ops.philox_normal(st.t, 9, B, O, row_base=1000)
        b = ops.lrt_gemm(x, ew, vw, I=I, O=O, eps=eps)
        assert torch.equal(a, b)


def test_layer_noise_advances_and_reseeds(bnn, dev):
    torch.manual_seed(5)
    layer = bnn.lrt.BayesianLinear(64, 32).to(dev).train()
    x = torch.rand(16, 64, device=dev)
    with torch.no_grad():
        bnn.manual_seed(42)
        a1 = layer(x, sample=True); a2 = layer(x, sample=True)
        bnn.manual_seed(42)
        b1 = layer(x, sample=True)
    assert not torch.equal(a1, a2)
    assert torch.equal(a1, b1)


# --------------------------------------------------------------------------- backward (interim torch-op recompute)
def test_backward_matches_oracle_autograd(bnn, dev, golden):
    c = golden("mnf.npz").case("c1")
    B, I, O, T = [int(v) for v in c["shape"]]
    p = sub(c, "p.")
    layer = _load_layer(bnn.mnf.BayesianLinear(I, O, T, z_flow_type="Planar", r_flow_type="Planar"), p, dev).train()
    noise = {k: c[k] for k in ("eps_z", "eps_out", "eps_z2", "eps_act")}
    layer.noise = {k: v.to(dev) for k, v in noise.items()}
    x = c["x"].to(dev).requires_grad_(True)
    out = layer(x, sample=True)
    loss = (out ** 2).sum() + layer.kl / 600
    loss.backward()
    # oracle under CPU autograd
    pc = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xc = c["x"].clone().requires_grad_(True)
    zf = orc.flow_from_state("z_flow", "Planar", pc, T)
    rf = orc.flow_from_state("r_flow", "Planar", pc, T)
    o, kl, _ = orc.mnf_forward(xc, pc, zf, rf, noise)
    ((o ** 2).sum() + kl / 600).backward()
    assert rel_err(x.grad, xc.grad) < TOL
    for name, prm in layer.named_parameters():
        assert rel_err(prm.grad, pc[name].grad) < TOL, name


def test_training_step_runs(bnn, dev):
    """The reference's train() body (LBBNN-GP-MF-MNF.py:265-272) runs unchanged on the modules."""
    torch.manual_seed(0)
    net = bnn.mnf.BayesianNetwork((784, 64, 48, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    data = torch.rand(32, 1, 28, 28, device=dev)
    target = torch.randint(0, 10, (32,), device=dev)
    net.train()
    losses = []
    for _ in range(5):
        net.zero_grad()
        outputs = net(data, sample=True)
        nll = torch.nn.functional.nll_loss(outputs, target, reduction="sum")
        loss = nll + net.kl() / 600
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(math.isfinite(v) for v in losses)
    assert losses[-1] < losses[0]


def test_error_paths(bnn, dev):
    ops = bnn.ops
    with pytest.raises(RuntimeError):
        ops.lrt_gemm(torch.rand(4, 8), torch.rand(4, 32), torch.rand(4, 32), I=8, O=4)          # CPU tensors
    x = torch.rand(4, 8, device=dev)
    w = torch.rand(4, 32, device=dev)
    with pytest.raises(RuntimeError, match="noise"):
        ops.lrt_gemm(x, w, w, I=8, O=4)                                                          # no eps, no rng
    with pytest.raises(NotImplementedError):
        l = bnn.mnf.BayesianLinear(8, 4, 2, z_flow_type="Planar", r_flow_type="RNVP").to(dev)   # mixed families
        l(x)
    with pytest.raises(NotImplementedError):
        bnn.flows.PropagateFlow("NoSuchFlow", 8, 2)


# --------------------------------------------------------------------------- scheduling variants
def test_stream_schedule_equals_sequential(bnn, dev, golden):
    """The fused no-grad schedule (batched K3/K1, ReLU/log_softmax in the GEMM epilogues, K5 of all layers at the
    end) and the per-layer autograd path compute the same numbers."""
    c = golden("mnf.npz").case("smallnet")
    dims = [int(v) for v in c["dims"]]
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar")
    for i, l in enumerate((net.l1, net.l2, net.l3)):
        l.load_state_dict(sub(c, "l%d.p." % i))
    net = net.to(dev).train()
    for i, l in enumerate((net.l1, net.l2, net.l3)):
        l.noise = {k: c["l%d.%s" % (i, k)].to(dev) for k in ("eps_z", "eps_out", "eps_z2", "eps_act")}
    x = c["x"].to(dev)
    with torch.no_grad():
        a = net(x, sample=True); kla = net.kl().clone()
    b = net(x, sample=True); klb = net.kl()
    assert b.requires_grad and klb.requires_grad
    assert rel_err(a, b) < 1e-6 and rel_err(kla, klb) < 1e-6
    assert rel_err(net.l1.kl + net.l2.kl + net.l3.kl, kla) < 1e-6


@pytest.mark.parametrize("B,I,O", [(4096, 1200, 10), (37, 100, 16), (16, 33, 3)])
def test_skinny_gemm_fused_log_softmax(bnn, dev, B, I, O):
    ops = bnn.ops
    g = torch.Generator().manual_seed(B + O)
    x = torch.rand(B, I, generator=g)
    ld = ops.operand_ld(I)
    ew = torch.zeros(O, ld); ew[:, :I] = 0.2 * (torch.rand(O, I, generator=g) - 0.5)
    vw = torch.zeros(O, ld); vw[:, :I] = 1e-3 * torch.rand(O, I, generator=g)
    bm = torch.rand(O, generator=g); bv = 1e-3 * torch.rand(O, generator=g)
    eps = torch.randn(B, O, generator=g)
    out = ops.lrt_gemm(x.to(dev), ew.to(dev), vw.to(dev), I=I, O=O, bias_mean=bm.to(dev), bias_var=bv.to(dev),
                       eps=eps.to(dev), log_softmax=True)
    x64 = x.double()
    pre = x64 @ ew[:, :I].double().T + bm.double() + torch.sqrt((x64 ** 2) @ vw[:, :I].double().T + bv.double()) * eps.double()
    assert rel_err(out, torch.log_softmax(pre, 1)) < TIGHT


def test_hip_graph_capture_replays_with_fresh_noise(bnn, dev):
    torch.manual_seed(3)
    net = bnn.mnf.BayesianNetwork((784, 256, 128, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    x = torch.rand(64, 784, device=dev)
    with torch.no_grad():
        for _ in range(3):
            net(x, sample=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = net(x, sample=True)
            kl = net.kl()
        g.replay(); torch.cuda.synchronize()
        o1, k1 = out.clone(), kl.clone()
        g.replay(); torch.cuda.synchronize()
        o2, k2 = out.clone(), kl.clone()
    assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
    assert not torch.equal(o1, o2)              # the device-side RNG offset advanced inside the graph
    assert abs(float(k1) - float(k2)) / abs(float(k1)) < 1e-2 and float(k1) != float(k2)


@pytest.mark.parametrize("kind,prec", [("Planar", "bf16x3"), ("Planar", "fp32"), ("RNVP", "bf16x3"), ("lrt", "bf16x3")])
def test_launch_plan_equals_the_eager_sequence_bitwise(bnn, dev, kind, prec):
    """graphs.LaunchPlan (the forward's C calls recorded once, replayed without the Python in between): from the same Philox
    {seed, offset}, call k of the plan gives the k-th eager forward's output, per-layer KL and total bit for bit; parameters
    changed in place between calls are picked up; a different stream is refused."""
    from bnn_amd import graphs, ops
    bnn.set_precision(prec)
    try:
        torch.manual_seed(13)
        if kind == "lrt":
            net = bnn.lrt.BayesianNetwork((784, 256, 128, 10)).to(dev).train()
        else:
            net = bnn.mnf.BayesianNetwork((784, 256, 128, 10), 2, z_flow_type=kind, r_flow_type=kind).to(dev).train()
        x = torch.rand(192, 1, 28, 28, device=dev)
        st = ops.RngState.get(dev)
        with torch.no_grad():
            net(x, sample=True); torch.cuda.synchronize()
            start = st.t[:2].clone()
            orig = net.l2.bias_mu.detach().clone()
            eager = []
            for k in range(3):
                if k == 2:
                    net.l2.bias_mu.add_(0.25)                       # an in-place parameter update between forwards
                o = net(x, sample=True)
                eager.append((o.clone(), net.kl().clone(), [l.kl.clone() for l in (net.l1, net.l2, net.l3)]))
            net.l2.bias_mu.copy_(orig)                              # (a + 0.25) - 0.25 is not a bit for bit
            torch.cuda.synchronize()
            plan = graphs.LaunchPlan(net, x, sample=True)
            assert 3 <= len(plan) <= 16
            st.t[:2].copy_(start)
            for k in range(3):
                if k == 2:
                    net.l2.bias_mu.add_(0.25)
                out, kl = plan()
                torch.cuda.synchronize()
                assert torch.equal(out, eager[k][0]) and torch.equal(kl, eager[k][1]) and torch.equal(net.kl(), eager[k][1]), k
                assert all(torch.equal(l.kl, e) for l, e in zip((net.l1, net.l2, net.l3), eager[k][2])), k
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                with pytest.raises(RuntimeError):
                    plan()
    finally:
        bnn.set_precision("fp32")


@pytest.mark.parametrize("prec", [None, "fp16x3f"])
@pytest.mark.parametrize("flow", ["RNVP", "MNF"])
def test_dense_flows_deferred_r_part_bitwise(bnn, dev, flow, prec, monkeypatch):
    """The fused no-grad forward of a net with dense flows runs the r flow + the flows' scalars on a side stream beside the weight
    pass and the first GEMM, and the KL finalize in the second GEMM's launch (layers._DENSE_DEFER): same outputs, same per-layer
    KL, same total, bit for bit, as with everything on one stream -- eager and under HIP-graph replay."""
    from bnn_amd import layers, ops
    torch.manual_seed(5)
    net = bnn.mnf.BayesianNetwork((784, 320, 256, 10), 2, z_flow_type=flow, r_flow_type=flow).to(dev).train()
    # (row-scaled fp16: the second GEMM folds the 10-class head into its epilogue, and a launch that carries the KL finalize
    # does not -- the deferral stands back there, so that the captured forward stays the eager one bit for bit)
    net.set_precision(prec)
    x = torch.rand(160, 784, device=dev)
    st = ops.RngState.get(dev)
    res = {}
    with torch.no_grad():
        net(x, sample=True); torch.cuda.synchronize()
        start = st.t[:2].clone()
        for defer in (False, "always"):
            monkeypatch.setattr(layers, "_DENSE_DEFER", defer)
            st.t[:2].copy_(start)
            outs = []
            for _ in range(2):
                o = net(x, sample=True)
                outs.append((o.clone(), net.kl().clone(), [l.kl.clone() for l in (net.l1, net.l2, net.l3)]))
            torch.cuda.synchronize()
            res[defer] = outs
        for a, b in zip(res[False], res["always"]):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            assert all(torch.equal(u, v) for u, v in zip(a[2], b[2]))
        # the deferred schedule inside a captured graph
        monkeypatch.setattr(layers, "_DENSE_DEFER", True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = net(x, sample=True)
            kl = net.kl()
        st.t[:2].copy_(start)
        for k in range(2):
            g.replay(); torch.cuda.synchronize()
            assert torch.equal(out, res[False][k][0]) and torch.equal(kl, res[False][k][1]), k


@pytest.mark.parametrize("flow", ["Planar", "RNVP"])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_hip_graph_replays_equal_the_eager_sequence_bitwise(bnn, dev, flow, prec):
    """What bench.py times by default -- ONE replay of the captured forward per step -- is the same computation as the eager
    launches: from the same Philox {seed, offset}, replay k gives the k-th eager forward's output and KL bit for bit (the
    offset lives on the device and is advanced by the forward's own kernels, inside the graph too)."""
    from bnn_amd import ops
    bnn.set_precision(prec)
    try:
        torch.manual_seed(11)
        net = bnn.mnf.BayesianNetwork((784, 256, 128, 10), 2, z_flow_type=flow, r_flow_type=flow).to(dev).train()
        x = torch.rand(192, 784, device=dev)
        st = ops.RngState.get(dev)
        with torch.no_grad():
            net(x, sample=True); torch.cuda.synchronize()
            start = st.t[:2].clone()
            eager = []
            for _ in range(3):
                o = net(x, sample=True)
                eager.append((o.clone(), net.kl().clone()))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = net(x, sample=True)
                kl = net.kl()
            st.t[:2].copy_(start)
            for k in range(3):
                g.replay(); torch.cuda.synchronize()
                assert torch.equal(out, eager[k][0]) and torch.equal(kl, eager[k][1]), k
            assert not torch.equal(eager[0][0], eager[1][0])
    finally:
        bnn.set_precision("fp32")


# --------------------------------------------------------------------------- split-precision (bf16x3) path
@pytest.mark.parametrize("B,I,O", [(128, 64, 80), (100, 784, 400), (257, 1200, 1200), (1024, 784, 400), (64, 40, 17),
                                   (4000, 784, 1200), (3333, 1200, 1190), (2100, 96, 1200)])
def test_split_gemm_vs_fp64(bnn, dev, B, I, O):
    """bf16x3 mean and variance products on the bf16 matrix cores: inside the 1e-4 contract (measured ~5e-6)."""
    ops = bnn.ops
    g = torch.Generator().manual_seed(B + I + O)
    x = torch.rand(B, I, generator=g)
    p = orc.init_mnf_params(I, O, g)
    z = 1 + 0.1 * torch.randn(I, generator=g)
    d = {k: v.to(dev) for k, v in p.items()}
    ld = ops.operand_ld(I)
    e_w = torch.empty(O, ld, device=dev); var_w = torch.empty(O, ld, device=dev); bias_var = torch.empty(O, device=dev)
    ops.weight_pass(d["weight_mu"], d["weight_rho"], d["lambdal"], z_fwd=z.to(dev), bias_rho=d["bias_rho"],
                    priors=bnn.Priors(), e_w=e_w, var_w=var_w, bias_var=bias_var, split=True)
    eps = torch.randn(B, O, generator=g)
    out = ops.lrt_gemm(x.to(dev), e_w, var_w, I=I, O=O, bias_mean=d["bias_mu"], bias_var=bias_var,
                       eps=eps.to(dev), relu=False, split=True)
    alpha = orc.alpha_of(p["lambdal"].double()); sigma = orc.sigma_of(p["weight_rho"].double())
    ew = p["weight_mu"].double() * alpha * z.double()
    vw = sigma ** 2 * alpha ** 2
    x64 = x.double()
    ref = x64 @ ew.T + p["bias_mu"].double() + torch.sqrt((x64 ** 2) @ vw.T + orc.sigma_of(p["bias_rho"].double()) ** 2) * eps.double()
    err = rel_err(out, ref)
    assert err < 2e-5, err
    # mean-only path too
    out = ops.lrt_gemm(x.to(dev), e_w, var_w, I=I, O=O, bias_mean=d["bias_mu"], mean_only=True, split=True)
    assert rel_err(out, x64 @ ew.T + p["bias_mu"].double()) < 5e-5


def test_split_headline_network_vs_oracle(bnn, dev):
    """Whole 784-1200-1200-10 MNF/planar net at B=4096 under set_precision('bf16x3') vs the fp32 oracle."""
    dims, B, T = (784, 1200, 1200, 10), 4096, 2
    torch.manual_seed(21)
    net = bnn.mnf.BayesianNetwork(dims, T, z_flow_type="Planar", r_flow_type="Planar")
    layers = [net.l1, net.l2, net.l3]
    g = torch.Generator().manual_seed(22)
    x = torch.rand(B, 784, generator=g)
    noises, P, zf, rf = [], [], [], []
    for l in layers:
        sd = {k: v.detach().clone() for k, v in l.state_dict().items()}
        P.append(sd)
        zf.append(orc.flow_from_state("z_flow", "Planar", sd, T))
        rf.append(orc.flow_from_state("r_flow", "Planar", sd, T))
        noises.append({"eps_z": torch.randn(1, l.in_features, generator=g),
                       "eps_out": torch.randn(B, l.out_features, generator=g),
                       "eps_z2": torch.randn(1, l.in_features, generator=g),
                       "eps_act": torch.randn(l.out_features, generator=g)})
    ref_out, ref_kl = orc.mnf_network_forward(x, P, zf, rf, noises)
    net = net.to(dev).train()
    for l, n in zip(layers, noises):
        l.noise = {k: v.to(dev) for k, v in n.items()}
    bnn.set_precision("bf16x3")
    try:
        with torch.no_grad():
            out = net(x.to(dev), sample=True)
            kl = net.kl()
            assert net.l1._split_now and net.l2._split_now and not net.l3._split_now
        # per-layer path (autograd Function) uses the same kernels
        out2 = net.l1(x.to(dev), sample=True)
    finally:
        bnn.set_precision("fp32")
    e = rel_err(out, ref_out)
    assert e < TOL, e
    assert rel_err(kl, ref_kl) < TIGHT          # KL never touches the reduced-precision operands
    assert torch.isfinite(out2).all()
    # against the fp64 truth, element-wise: the split path carries 16 mantissa bits per operand, so its absolute error
    # floor is ~3e-6 max|ref| (measured) where the fp32 path's is ~3e-7: atol = 1e-5 max|ref| here, 1e-6 there
    ref64, kls64 = _oracle_mnf_net(*_to64(x, P, zf, rf, noises))
    v = elementwise_violation(out, ref64, rtol=1e-4, atol_frac=1e-5)
    assert v <= 1.0, v
    for l, k_ref in zip(layers, kls64):
        assert rel_err(l.kl, k_ref) < TOL


# --------------------------------------------------------------------------- dense coupling flows (RNVP / MNF type)
@pytest.mark.parametrize("case", ["c4", "c5", "c6", "c7"])
def test_mnf_layer_dense_flows_vs_golden(bnn, dev, golden, case):
    """The reference's DEFAULT flow type (RNVP, LBBNN-GP-MF-MNF.py:46-47) and the MNF-type flow."""
    c = golden("mnf.npz").case(case)
    B, I, O, T = [int(v) for v in c["shape"]]
    kind = str(c["kind"])
    assert kind in ("RNVP", "MNF")
    layer = _load_layer(bnn.mnf.BayesianLinear(I, O, T, z_flow_type=kind, r_flow_type=kind), sub(c, "p."), dev)
    x = c["x"].to(dev)
    with torch.no_grad():
        layer.train()
        n = {k: c[k].to(dev) for k in ("eps_z", "eps_out", "eps_z2", "eps_act")}
        n["zmask"] = [c["zmask%d" % t].to(dev) for t in range(T)]
        n["zmask2"] = [c["zmask2_%d" % t].to(dev) for t in range(T)]
        n["rmask"] = [c["rmask%d" % t].to(dev) for t in range(T)]
        layer.noise = n
        out = layer(x, sample=True)
        assert rel_err(out, c["out_train"]) < TIGHT
        assert rel_err(layer.kl, c["kl"]) < TIGHT
        layer.eval()
        layer.noise = {"eps_z": c["eps_z_eval"].to(dev), "zmask": [c["zmask_eval%d" % t].to(dev) for t in range(T)]}
        out = layer(x, sample=False)
        assert rel_err(out, c["out_eval_mean"]) < TIGHT
        # default construction (RNVP, in-kernel noise, device-drawn masks) runs and is finite
        layer.noise = None
        layer.train()
        out = layer(x, sample=True)
        assert torch.isfinite(out).all() and torch.isfinite(layer.kl)
        z, ld = layer.sample_z(B)
        assert z.shape == (I,) and torch.isfinite(z).all() and torch.isfinite(ld).all()
        assert ld.shape == ((B,) if kind == "RNVP" else ()) and layer.z.shape == (B, I)     # …MNF.py:185-187, flows2.py:219,241


@pytest.mark.parametrize("case", ["c0", "c1", "c4", "c5", "c6", "c7"])
def test_sample_z_and_forward_as_written_vs_golden(bnn, dev, golden, case):
    """``sample_z(batch_size)`` AS WRITTEN (LBBNN-GP-MF-MNF.py:182-187): all batch_size rows through z_flow, z0 kept in
    ``self.z``, (zs[-1], logdet.squeeze()) returned with the reference's logdet shape -- against the oracle's restatement on
    the reference layer's recorded (B,I) draws and masks; then ``as_written = True``: the layer forward whose z comes from
    that B-row flow must give the reference layer's recorded output like the default kept-row forward does."""
    c = golden("mnf.npz").case(case)
    B, I, O, T = [int(v) for v in c["shape"]]
    kind = str(c["kind"]) if "kind" in c else "Planar"
    p = sub(c, "p.")
    layer = _load_layer(bnn.mnf.BayesianLinear(I, O, T, z_flow_type=kind, r_flow_type=kind), p, dev).train()
    dense = kind in ("RNVP", "MNF")
    eps_z = c["eps_z"]
    assert tuple(eps_z.shape) == (B, I)
    masks = [c["zmask%d" % t] for t in range(T)] if dense else None
    layer.noise = {"eps_z": eps_z.to(dev)}
    if dense:
        layer.noise["zmask"] = [m.to(dev) for m in masks]
    z, ld = layer.sample_z(B)
    zr, ldr, z0r = orc.mnf_sample_z(p, eps_z, orc.flow_from_state("z_flow", kind, p, T), masks)
    assert z.shape == zr.shape and rel_err(z, zr) < TIGHT
    assert ld.shape == ldr.shape and float((ld.cpu() - ldr).abs().max()) < 1e-6 + 2e-5 * float(ldr.abs().max())
    assert layer.z.shape == (B, I) and rel_err(layer.z, z0r) < TIGHT
    # the forward, both ways
    n = {k: c[k].to(dev) for k in ("eps_z", "eps_out", "eps_z2", "eps_act")}
    if dense:
        n["zmask"] = [c["zmask%d" % t].to(dev) for t in range(T)]
        n["zmask2"] = [c["zmask2_%d" % t].to(dev) for t in range(T)]
        n["rmask"] = [c["rmask%d" % t].to(dev) for t in range(T)]
    layer.noise = n
    with torch.no_grad():
        ref_path = layer(c["x"].to(dev), sample=True)
        kl_kept = layer.kl.clone()
        layer.as_written = True
        out = layer(c["x"].to(dev), sample=True)
    assert rel_err(out, c["out_train"]) < TIGHT and rel_err(layer.kl, c["kl"]) < TIGHT
    assert rel_err(out, ref_path) < TIGHT and torch.equal(layer.kl, kl_kept)
    with pytest.raises(RuntimeError):
        layer(c["x"].to(dev).requires_grad_(True), sample=True)            # as_written is an evaluation / timing mode


def test_as_written_layers_in_kernel_noise_match_kept_row_path(bnn, dev):
    """Planar flows, in-kernel Philox noise, the headline layer sizes at B = 512: the as-written B-row z flow draws the
    kept row's eps from the same counters as the fused kept-row kernel, so both forwards of a layer agree (to rounding:
    the row kernel sums in another order) -- the discarded B-1 rows change nothing, as in the reference."""
    torch.manual_seed(13)
    net = bnn.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    x = torch.rand(512, 784, device=dev)
    with torch.no_grad():
        for l in (net.l1, net.l2, net.l3):
            l.as_written = False
            bnn.manual_seed(7, 0)
            a = l(x, sample=True).clone()
            kl_a = l.kl.clone()
            l.as_written = True
            bnn.manual_seed(7, 0)
            b = l(x, sample=True)
            assert rel_err(b, a) < 2e-5 and rel_err(l.kl, kl_a) < 1e-6
            x = torch.relu(a)
        # and the network runs with every layer in that mode
        out = net(torch.rand(512, 784, device=dev), sample=True)
        assert torch.isfinite(out).all() and torch.isfinite(net.kl())


def test_dense_flow_backward_matches_oracle_autograd(bnn, dev, golden):
    c = golden("mnf.npz").case("c5")           # RNVP, (5,33,17), T=2
    B, I, O, T = [int(v) for v in c["shape"]]
    p = sub(c, "p.")
    layer = _load_layer(bnn.mnf.BayesianLinear(I, O, T), p, dev).train()
    noise = {k: c[k] for k in ("eps_z", "eps_out", "eps_z2", "eps_act")}
    noise["zmask"] = [c["zmask%d" % t] for t in range(T)]
    noise["zmask2"] = [c["zmask2_%d" % t] for t in range(T)]
    noise["rmask"] = [c["rmask%d" % t] for t in range(T)]
    layer.noise = {k: ([m.to(dev) for m in v] if isinstance(v, list) else v.to(dev)) for k, v in noise.items()}
    x = c["x"].to(dev).requires_grad_(True)
    out = layer(x, sample=True)
    ((out ** 2).sum() + layer.kl / 600).backward()
    pc = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xc = c["x"].clone().requires_grad_(True)
    zf = orc.flow_from_state("z_flow", "RNVP", pc, T)
    rf = orc.flow_from_state("r_flow", "RNVP", pc, T)
    o, kl, _ = orc.mnf_forward(xc, pc, zf, rf, noise)
    ((o ** 2).sum() + kl / 600).backward()
    assert rel_err(x.grad, xc.grad) < TOL
    for name, prm in layer.named_parameters():
        assert rel_err(prm.grad, pc[name].grad) < 5e-4, name     # tiny gradients through the 4-layer MLP: looser


def test_reference_default_network_trains(bnn, dev):
    """BayesianNetwork() exactly as the reference constructs it (RNVP flows, 784-400-600-10) takes SGD steps."""
    torch.manual_seed(0)
    net = bnn.mnf.BayesianNetwork().to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    data = torch.rand(16, 1, 28, 28, device=dev)
    target = torch.randint(0, 10, (16,), device=dev)
    net.train()
    vals = []
    for _ in range(3):
        net.zero_grad()
        loss = torch.nn.functional.nll_loss(net(data, sample=True), target, reduction="sum") + net.kl() / 600
        loss.backward()
        opt.step()
        vals.append(float(loss.detach()))
    assert all(math.isfinite(v) for v in vals)
    with torch.no_grad():
        net.eval()
        out = net(data, sample=False)
    assert out.shape == (16, 10) and torch.isfinite(out).all()


# --------------------------------------------------------------------------- baseline LBBNN (gate x Gaussian sampling)
@pytest.mark.parametrize("case", ["c0", "c1", "c2"])
def test_base_layer_vs_golden(bnn, dev, golden, case):
    c = golden("base.npz").case(case)
    B, I, O = [int(v) for v in c["shape"]]
    layer = _load_layer(bnn.base.BayesianLinear(I, O, 1), sub(c, "p."), dev)
    x = c["x"].to(dev)
    cg = c["cgamma"].to(dev)
    with torch.no_grad():
        layer.train()
        layer.alpha = 1 / (1 + torch.exp(-layer.lambdal))            # as sample_elbo does (LBBNN-GP-MF.py:292-293)
        layer.gamma.alpha = layer.alpha
        layer.noise = {k: c[k].to(dev) for k in ("eps_w", "eps_b", "tau_w", "tau_b")}
        out = layer(x, cg, sample=True)
        assert rel_err(out, c["out_train"]) < TIGHT
        assert rel_err(layer.log_prior, c["log_prior"]) < 2e-5
        assert rel_err(layer.log_variational_posterior, c["log_q"]) < 2e-5
        # exact=True variant (end of training, :612-627): hard gate
        for o in (layer.weight_prior, layer.bias_prior, layer.gamma_prior, layer.gamma):
            o.exact = True
        hard = torch.round(cg)
        layer.noise = {k: c["x_" + k].to(dev) for k in ("eps_w", "eps_b", "tau_w", "tau_b")}
        out = layer(x, hard, sample=True)
        assert rel_err(out, c["x_out_train"]) < TIGHT
        assert rel_err(layer.log_prior, c["x_log_prior"]) < 2e-5
        assert rel_err(layer.log_variational_posterior, c["x_log_q"]) < 2e-5
        layer.eval()
        layer.noise = None
        assert rel_err(layer(x, hard, sample=False, medimean=True), c["out_medimean"]) < TIGHT
        layer.alpha = c["alpha_attr"].to(dev)
        assert rel_err(layer(x, hard, sample=False, medimean=False), c["out_mean"]) < TIGHT
        assert layer.log_prior == 0


def test_base_backward_and_sample_elbo(bnn, dev, golden):
    c = golden("base.npz").case("c1")
    B, I, O = [int(v) for v in c["shape"]]
    p = sub(c, "p.")
    layer = _load_layer(bnn.base.BayesianLinear(I, O, 1), p, dev).train()
    noise = {k: c[k] for k in ("eps_w", "eps_b", "tau_w", "tau_b")}
    layer.noise = {k: v.to(dev) for k, v in noise.items()}
    with torch.no_grad():
        layer.gamma.alpha = 1 / (1 + torch.exp(-layer.lambdal))
    x = c["x"].to(dev).requires_grad_(True)
    out = layer(x, c["cgamma"].to(dev), sample=True)
    ((out ** 2).sum() + (layer.log_variational_posterior - layer.log_prior) / 600).backward()
    pc = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xc = c["x"].clone().requires_grad_(True)
    o, lp, lq = orc.base_forward(xc, pc, c["cgamma"], noise, mode="sample",
                                 gamma_alpha=orc.alpha_of(pc["lambdal"]).detach())
    ((o ** 2).sum() + (lq - lp) / 600).backward()
    assert rel_err(x.grad, xc.grad) < TOL
    for name, prm in layer.named_parameters():
        if pc[name].grad is None:
            continue
        assert rel_err(prm.grad, pc[name].grad) < 2e-4, name
    # whole-network ELBO sample as the reference's train() calls it (:331-337)
    torch.manual_seed(1)
    net = bnn.base.BayesianNetwork((784, 64, 48, 10)).to(dev).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    data = torch.rand(32, 1, 28, 28, device=dev); target = torch.randint(0, 10, (32,), device=dev)
    vals = []
    for _ in range(3):
        net.zero_grad()
        loss, lp, lq, nll = net.sample_elbo(data, target)
        loss.backward(); opt.step()
        vals.append(float(loss.detach()))
    assert all(math.isfinite(v) for v in vals)


@pytest.mark.parametrize("case", ["c0", "c1", "c2"])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_base_backward_all_hip_full_graph_vs_oracle_autograd(bnn, dev, golden, case, prec):
    """loss.backward() of the baseline layer as sample_elbo builds the graph (LBBNN-GP-MF.py:292-318): alpha = sigmoid(lambdal)
    feeds BOTH the relaxed gate draw and Bernoulli.log_prob, taus are Gamma rsamples of (a, b).  Every gradient --
    x, weight_mu/rho, lambdal (through the gate AND through alpha), weight_a/b, pa, pb, bias_mu/rho/a/b -- from the HIP
    backward (lbbnn_output_grad, GEMMs, lbbnn_gate_backward) against fp64 autograd of the oracle on the same draws.  The
    gate uses temperature 0.5 instead of the reference's 0.001 so that it is not saturated and its gradient is informative."""
    c = golden("base.npz").case(case)
    B, I, O = [int(v) for v in c["shape"]]
    p = sub(c, "p.")
    layer = _load_layer(bnn.base.BayesianLinear(I, O, 1), p, dev).train()
    g = torch.Generator().manual_seed(77)
    logistic = torch.log(torch.rand(O, I, generator=g).clamp(1e-6, 1 - 1e-6))
    logistic = logistic - torch.log1p(-torch.exp(logistic))
    noise = {k: c[k] for k in ("eps_w", "eps_b")}
    gam_w = torch.rand(1, generator=g) + 0.5          # stand-ins for the Gamma draws' dependence on (a, b): tau = u * a / b
    gam_b = torch.rand(O, generator=g) + 0.5

    def build(P, x, lib):
        alpha = 1 / (1 + torch.exp(-P["lambdal"]))
        cg = torch.sigmoid((torch.log(alpha) - torch.log1p(-alpha) + lib(logistic)) / 0.5)
        tau_w = lib(gam_w) * P["weight_a"] / P["weight_b"]
        tau_b = lib(gam_b) * P["bias_a"] / P["bias_b"]
        return alpha, cg, tau_w, tau_b

    # HIP path
    P = dict(layer.named_parameters())
    x = c["x"].to(dev).requires_grad_(True)
    alpha, cg, tau_w, tau_b = build(P, x, lambda t: t.to(dev))
    layer.gamma.alpha = alpha
    layer.noise = {"eps_w": noise["eps_w"].to(dev), "eps_b": noise["eps_b"].to(dev), "tau_w": tau_w, "tau_b": tau_b}
    bnn.set_precision(prec)
    try:
        out = layer(x, cg, sample=True)
        loss = (out ** 2).sum() + (layer.log_variational_posterior - layer.log_prior) / 600
        loss.backward()
    finally:
        bnn.set_precision("fp32")
    # oracle, fp64
    P64 = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
    x64 = c["x"].double().requires_grad_(True)
    a64, cg64, tw64, tb64 = build(P64, x64, lambda t: t.double())
    n64 = {"eps_w": noise["eps_w"].double(), "eps_b": noise["eps_b"].double(), "tau_w": tw64, "tau_b": tb64}
    o, lp, lq = orc.base_forward(x64, P64, cg64, n64, mode="sample", gamma_alpha=a64)
    ((o ** 2).sum() + (lq - lp) / 600).backward()
    assert rel_err(out, o) < (TIGHT if prec == "fp32" else 2e-5)
    assert rel_err(layer.log_prior, lp) < 2e-5 and rel_err(layer.log_variational_posterior, lq) < 2e-5
    tol = 2e-4 if prec == "fp32" else 5e-4
    assert rel_err(x.grad, x64.grad) < tol
    for name, prm in layer.named_parameters():
        assert P64[name].grad is not None, name
        assert rel_err(prm.grad, P64[name].grad) < tol, name


def test_base_network_sample_elbo_vs_reference_anchor(bnn, dev, golden):
    """net.sample_elbo(input, target) -- the reference's call, LBBNN-GP-MF.py:331 -- on the HIP path against the
    reference's own numbers at SURVEY.md 8c's anchor (tests/golden/base_elbo.npz: seed 0, 784-400-600-10, B = 100;
    loss 2736.66748, log_prior -689876.94, log_q 707226.31, nll 408.1622), with the recorded draws injected."""
    c = golden("base_elbo.npz").case("elbo")
    torch.manual_seed(0)
    net = bnn.base.BayesianNetwork().to(dev).train()
    for li, l in enumerate((net.l1, net.l2, net.l3)):
        l.noise = {k: c["l%d.%s" % (li + 1, k)].to(dev) for k in ("eps_w", "eps_b", "tau_w", "tau_b")}
        g = c["l%d.cgamma" % (li + 1)].to(dev)
        l.gamma.rsample = (lambda g=g: g)                      # the relaxed-Bernoulli draw of :300-302, as recorded
    for prec in ("fp32", "bf16x3"):
        bnn.set_precision(prec)
        try:
            loss, lp, lq, nll = net.sample_elbo(c["x"].to(dev), c["target"].to(dev))
        finally:
            bnn.set_precision("fp32")
        assert rel_err(lp, c["log_prior"]) < 2e-5, prec
        assert rel_err(lq, c["log_q"]) < 2e-5, prec
        assert rel_err(nll, c["nll"]) < TOL, prec
        assert rel_err(loss, c["loss"]) < 2e-5, prec
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


# --------------------------------------------------------------------------- variational dropout
@pytest.mark.parametrize("case", ["c0", "c1", "c2"])
def test_vd_layer_vs_golden(bnn, dev, golden, case):
    c = golden("vd.npz").case(case)
    B, I, O = [int(v) for v in c["shape"]]
    layer = bnn.vd.BayesianLayer(I, O)
    with torch.no_grad():
        layer.theta.copy_(c["theta"])
    layer = layer.to(dev)
    layer.noise = {"zeta": c["zeta"].to(dev)}
    with torch.no_grad():
        out = layer(c["x"].to(dev))
    assert rel_err(out, c["out"]) < TIGHT
    pred = torch.log_softmax(out, 1)
    loss = bnn.vd.loss_fn(pred, c["target"].to(dev), torch.nn.Sequential(layer), num_batches=float(c["num_batches"]))
    assert rel_err(loss, c["loss"]) < 1e-5
    loss = bnn.vd.loss_fn(pred, c["target"].to(dev), torch.nn.Sequential(layer))     # the reference's 3-argument call
    assert rel_err(loss, c["loss"]) < 1e-5


@pytest.mark.parametrize("B,n,m", [(37, 50, 24), (64, 96, 10), (128, 256, 160), (33, 16, 7)])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_vd_backward_all_hip_vs_oracle_autograd(bnn, dev, B, n, m, prec):
    """variational_dropout.py:170 ``loss.backward()`` through BayesianLayer: dX and dtheta from the HIP kernels
    (lbbnn_output_grad with gv_scale = alpha, lbbnn_format_operand, lbbnn_vd_operands, lbbnn_transpose_operand, the GEMM
    with the combine epilogue) against fp64 autograd of the oracle; injected zeta and the in-kernel draw."""
    torch.manual_seed(14)
    layer = bnn.vd.BayesianLayer(n, m).to(dev)
    g = torch.Generator().manual_seed(15)
    x = torch.randn(B, n, generator=g)
    zeta = torch.randn(B, m, generator=g)
    c = torch.randn(B, m, generator=g)
    layer.noise = {"zeta": zeta.to(dev)}
    xd = x.to(dev).requires_grad_(True)
    bnn.set_precision(prec)
    try:
        out = layer(xd)
        (out * c.to(dev)).sum().backward()
    finally:
        bnn.set_precision("fp32")
    x64 = x.double().requires_grad_(True)
    th64 = layer.theta.detach().cpu().double().requires_grad_(True)
    ref = orc.vd_forward(x64, th64, layer.alpha.cpu().double(), zeta.double())
    (ref * c.double()).sum().backward()
    tol = 2e-4
    assert rel_err(out, ref) < (TIGHT if prec == "fp32" else 2e-5)
    assert rel_err(xd.grad, x64.grad) < tol and rel_err(layer.theta.grad, th64.grad) < tol
    # in-kernel zeta: the backward re-creates the forward's draw from the saved Philox state
    layer.noise = None
    layer.theta.grad = None
    bnn.manual_seed(3, 0)
    st = bnn.ops.RngState.get(dev)
    z_used = bnn.ops.philox_normal(st.t, bnn.ops.STREAM_EPS_OUT * 64 + layer._layer_id, B, m, 0)
    xd2 = x.to(dev).requires_grad_(True)
    out2 = layer(xd2)
    (out2 * c.to(dev)).sum().backward()
    x64b = x.double().requires_grad_(True)
    th64b = layer.theta.detach().cpu().double().requires_grad_(True)
    (orc.vd_forward(x64b, th64b, layer.alpha.cpu().double(), z_used.cpu().double()) * c.double()).sum().backward()
    assert rel_err(xd2.grad, x64b.grad) < tol and rel_err(layer.theta.grad, th64b.grad) < tol


def test_vd_network_full_size_and_training(bnn, dev):
    """BASELINE configs[4] shape (3072-4096-4096-10 'CIFAR-flat') forward vs fp64, in both precisions; then
    the reference's BNN (784-1200-1200-1200-10) takes optimizer steps."""
    torch.manual_seed(2)
    I, O, B = 3072, 4096, 512
    layer = bnn.vd.BayesianLayer(I, O).to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, I, generator=g); zeta = torch.randn(B, O, generator=g)
    layer.noise = {"zeta": zeta.to(dev)}
    th = layer.theta.detach().cpu().double()
    ref = x.double() @ th + torch.sqrt((x.double() ** 2) @ th ** 2 * 0.2) * zeta.double()
    for prec in ("fp32", "bf16x3"):
        bnn.set_precision(prec)
        try:
            with torch.no_grad():
                out = layer(x.to(dev))
        finally:
            bnn.set_precision("fp32")
        assert rel_err(out, ref) < (TIGHT if prec == "fp32" else 2e-5), prec
    net = bnn.vd.BNN().to(dev)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4)
    data = torch.rand(64, 1, 28, 28, device=dev); target = torch.randint(0, 10, (64,), device=dev)
    vals = []
    for _ in range(3):
        net.zero_grad()
        loss = bnn.vd.loss_fn(net(data), target, net)
        loss.backward(); opt.step()
        vals.append(float(loss.detach()))
    assert all(math.isfinite(v) for v in vals) and vals[-1] < vals[0]


def test_graphed_training_step_subprocess():
    """Whole training step (HIP forward + hybrid backward + Adam) captured in a HIP graph and replayed.
    Runs in its own process: capture wants a clean autograd state."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, torch
sys.path.insert(0, %r)
import bnn_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 128, 64, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
x = torch.rand(256, 1, 28, 28, device=dev); y = torch.randint(0, 10, (256,), device=dev)
lf = lambda n, a, b: torch.nn.functional.nll_loss(n(a, sample=True), b, reduction="sum") + n.kl() / 100
step = bnn_amd.graphs.make_graphed_train_step(net, opt, lf, x, y)
vals = [float(step(x, y).detach().clone()) for _ in range(30)]
torch.cuda.synchronize()
assert all(v == v for v in vals), vals
assert vals[-1] < vals[0], (vals[0], vals[-1])
assert len(set(vals)) > 20
print("GRAPH_OK", vals[0], vals[-1])
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_graphed_data_parallel_step_equals_eager_subprocess():
    """parallel.DataParallelELBO.make_graphed_step (graph A: forward + backward + bucket pack | eager all-reduce | graph B:
    Adam) at world size 1: after the same number of steps from the same state and the same Philox offsets the parameters
    are bitwise those of the eager bucket step.  Own process (capture wants a clean autograd state)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, copy, torch
sys.path.insert(0, %r)
import bnn_amd
from bnn_amd.parallel import DataParallelELBO
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 128, 64, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
init = copy.deepcopy(net.state_dict())
x = torch.rand(256, 1, 28, 28, device=dev); y = torch.randint(0, 10, (256,), device=dev)
res = []
for mode in ("graph", "eager"):       # graph first: an eager autograd graph alive on the default stream breaks a later capture
    net.load_state_dict(init)
    opt = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)
    dp = DataParallelELBO(net)
    if mode == "graph":
        step = dp.make_graphed_step(opt, x, y, 100, warmup=2)
        net.load_state_dict(init)                         # the warm-up steps moved the parameters: start over
        opt2 = None
        for st in opt.state.values():
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        for g in opt.param_groups:
            g["step_dev"].zero_()
    losses = []
    for it in range(4):
        bnn_amd.manual_seed(50 + it)
        if mode == "eager":
            opt.zero_grad(set_to_none=True)
            loss = dp.loss(net(x, sample=True), y, 100)
            from bnn_amd import layers
            with layers.vector_backward_overlap():
                loss.backward()
            dp.all_reduce_grads(unpack=False)
            opt.step(grads=dp.reduced_grads())
        else:
            loss = step(x, y)
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    res.append(({k: v.detach().clone() for k, v in net.named_parameters()}, losses))
assert res[0][1] == res[1][1], (res[0][1], res[1][1])
for k in res[0][0]:
    assert torch.equal(res[0][0][k], res[1][0][k]), k
assert res[0][1][-1] < res[0][1][0]
print("DPGRAPH_OK", res[0][1])
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DPGRAPH_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2500:])


def test_reduced_precision_bf16_single_product_mode(bnn, dev):
    """set_precision('bf16'): ONE bf16 product per moment in the forward's dual-moment GEMM -- the arithmetic BASELINE
    configs[1] names ("LRT 784-400-400-10, batch 1024 bf16").  It is a REDUCED-precision mode with its own, measured
    tolerance (2.2e-3 relative L2 on the mean GEMM): outputs within 1e-2 of the fp64 oracle under the
    max norm, and demonstrably NOT within the 1e-4 contract -- which is why it is never the default."""
    dims, B = (784, 400, 400, 10), 1024
    torch.manual_seed(8)
    net = bnn.lrt.BayesianNetwork(dims)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 784, generator=g)
    eps = [torch.randn(B, d, generator=g) for d in dims[1:]]
    P = [{k: v.detach().double() for k, v in l.state_dict().items()} for l in (net.l1, net.l2, net.l3)]
    ref, _ = orc.lrt_network_forward(x.double(), P, [e.double() for e in eps])
    net = net.to(dev).train()
    for l, e in zip((net.l1, net.l2, net.l3), eps):
        l.noise = {"eps_out": e.to(dev)}
    errs = {}
    for prec in ("bf16x3", "bf16"):
        bnn.set_precision(prec)
        try:
            with torch.no_grad():
                out = net(x.to(dev), sample=True)
                assert net.l1._split_now and net.l2._split_now
        finally:
            bnn.set_precision("fp32")
        errs[prec] = rel_err(out, ref)
    assert errs["bf16x3"] < 2e-5
    assert 1e-4 < errs["bf16"] < 1e-2, errs


# --------------------------------------------------------------------------- BASELINE.json configs as parity cases
def test_baseline_config1_lrt_400_b1024_split(bnn, dev):
    """configs[1]: LBBNN-GP-MF-LRT 784-400-400-10, batch 1024, reduced-precision matrix-core path (bf16x3 here),
    whole network vs the fp32 oracle on the same seeded draws; and configs[0]'s shape (784-400-400-10, B=100)
    for the baseline LBBNN is covered by test_base_* above."""
    dims, B = (784, 400, 400, 10), 1024
    torch.manual_seed(31)
    net = bnn.lrt.BayesianNetwork(dims)
    g = torch.Generator().manual_seed(32)
    x = torch.rand(B, 784, generator=g)
    layers = [net.l1, net.l2, net.l3]
    P = [{k: v.detach().clone() for k, v in l.state_dict().items()} for l in layers]
    eps = [torch.randn(B, l.out_features, generator=g) for l in layers]
    ref_out, ref_kl = orc.lrt_network_forward(x, P, eps)
    net = net.to(dev).train()
    for l, e in zip(layers, eps):
        l.noise = {"eps_out": e.to(dev)}
    for prec, tol in (("fp32", TIGHT), ("bf16x3", 2e-5)):
        bnn.set_precision(prec)
        try:
            with torch.no_grad():
                out = net(x.to(dev), sample=True)
                kl = net.kl()
        finally:
            bnn.set_precision("fp32")
        assert rel_err(out, ref_out) < tol, prec
        assert rel_err(kl, ref_kl) < TIGHT


def test_baseline_config0_base_lbbnn_400_400_b100(bnn, dev):
    """configs[0] at the dims BASELINE.json's string names -- LBBNN-GP-MF.py, 784-400-400-10, batch 100, one MC sample (the
    reference's own BayesianNetwork() is 784-400-600-10: tests/golden/base_elbo.npz anchors that one): the whole sampled
    network on the HIP path against the oracle's restatement of LBBNN-GP-MF.py:228-255, 285-319 on the same draws --
    log-probabilities, log prior, log variational posterior, nll and the ELBO loss."""
    dims, B = (784, 400, 400, 10), 100
    torch.manual_seed(11)
    net = bnn.base.BayesianNetwork(dims)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(B, 784, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    layers = [net.l1, net.l2, net.l3]
    P = [{k: v.detach().clone() for k, v in l.state_dict().items()} for l in layers]
    gates = [torch.rand(l.out_features, l.in_features, generator=g) for l in layers]
    noises = [{"eps_w": torch.randn(l.out_features, l.in_features, generator=g), "eps_b": torch.randn(l.out_features, generator=g),
               "tau_w": 0.5 + torch.rand(1, generator=g), "tau_b": 0.5 + torch.rand(l.out_features, generator=g)} for l in layers]
    h, lp_ref, lq_ref = x, 0.0, 0.0
    for i, (p, cg, n) in enumerate(zip(P, gates, noises)):
        h, lp, lq = orc.base_forward(h, p, cg, n, mode="sample")
        lp_ref, lq_ref = lp_ref + lp, lq_ref + lq
        if i < 2:
            h = torch.relu(h)
    ref_out = torch.log_softmax(h, dim=1)
    ref_nll = torch.nn.functional.nll_loss(ref_out, y, reduction="sum")
    ref_loss = ref_nll + (lq_ref - lp_ref) / 600
    net = net.to(dev).train()
    with torch.no_grad():
        for l, n in zip(layers, noises):
            l.alpha = 1 / (1 + torch.exp(-l.lambdal))                 # as sample_elbo does (LBBNN-GP-MF.py:292-297)
            l.gamma.alpha = l.alpha
            l.noise = {k: v.to(dev) for k, v in n.items()}
        out = net(x.to(dev), gates[0].to(dev), gates[1].to(dev), gates[2].to(dev), sample=True)
        lp, lq = net.log_prior(), net.log_variational_posterior()
        nll = torch.nn.functional.nll_loss(out, y.to(dev), reduction="sum")
        loss = nll + (lq - lp) / 600
    assert rel_err(out, ref_out) < 2e-5
    assert rel_err(lp, lp_ref) < 2e-5 and rel_err(lq, lq_ref) < 2e-5
    assert rel_err(nll, ref_nll) < 2e-5 and rel_err(loss, ref_loss) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("O,I,mnf", [(37, 53, True), (64, 128, True), (130, 1200, True), (48, 64, False)])
def test_weight_pass_backward_kernel(bnn, dev, O, I, mnf):
    """K1b (lbbnn_weight_pass_backward) against fp64 autograd of the same chain: e_w, var_w, weight KL and the
    auxiliary activations act_mu / act_var as functions of (mu, rho, lambda, z_fwd, z_kl, r0_c)."""
    from bnn_amd import ops
    g = torch.Generator().manual_seed(O * 1000 + I)
    pr = bnn.Priors()
    mu = (torch.rand(O, I, generator=g) - 0.5) * 0.4
    rho = -5 + torch.rand(O, I, generator=g) * 3
    lam = torch.randn(O, I, generator=g) * 2
    dWm, dWv = torch.randn(O, I, generator=g), torch.randn(O, I, generator=g)
    zf, zk, rc = 1 + 0.1 * torch.randn(I, generator=g), 1 + 0.1 * torch.randn(I, generator=g), 0.1 * torch.randn(I, generator=g)
    dam, dav = torch.randn(O, generator=g), torch.randn(O, generator=g)
    gk = torch.tensor(1 / 600.)
    leaves = [t.double().requires_grad_(True) for t in (mu, rho, lam, zf, zk, rc)]
    m, r, l, f, k, c = leaves
    alpha, sigma = torch.sigmoid(l), torch.log1p(torch.exp(r))
    k_eff = k if mnf else torch.ones_like(k)
    Wv = sigma ** 2 * alpha ** 2
    klw = (alpha * (math.log(pr.sigma_prior) - sigma.log() - 0.5 + (alpha / pr.alpha_prior).log()
                    + (sigma ** 2 + (m * k_eff - pr.mu_prior) ** 2) / (2 * pr.sigma_prior ** 2))
           + (1 - alpha) * ((1 - alpha) / (1 - pr.alpha_prior)).log()).sum()
    obj = (dWm.double() * (m * alpha * (f if mnf else 1))).sum() + (dWv.double() * Wv).sum() + gk.double() * klw
    if mnf:
        obj = obj + (dam.double() * (c @ (k * m * alpha).T)).sum() + (dav.double() * (c ** 2 @ Wv.T)).sum()
    ref = torch.autograd.grad(obj, leaves, allow_unused=True)
    d = lambda t: t.to(dev)
    got = ops.weight_pass_backward(d(mu), d(rho), d(lam), d(dWm), d(dWv), z_fwd=d(zf) if mnf else None,
                                   z_kl=d(zk) if mnf else None, r0_c=d(rc) if mnf else None,
                                   da_mu=d(dam) if mnf else None, da_var=d(dav) if mnf else None, g_kl=d(gk), priors=pr)
    names = ["dmu", "drho", "dlambdal", "dz_fwd", "dz_kl", "dr0_c"]
    for n, a, b in zip(names, got, ref):
        if a is None:
            assert not mnf
            continue
        assert rel_err(a.cpu().double(), b) < 2e-5, n


def _mnf_planar_case(bnn, dev, B, I, O, T, seed, flow_scale=8.0):
    torch.manual_seed(seed)
    layer = bnn.mnf.BayesianLinear(I, O, T, z_flow_type="Planar", r_flow_type="Planar")
    with torch.no_grad():                       # make the flows matter: u, w ~ U(+-0.08) instead of +-0.01
        for fl in (layer.z_flow, layer.r_flow):
            for tr in fl.transforms:
                tr.u.mul_(flow_scale); tr.w.mul_(flow_scale); tr.bias.mul_(flow_scale)
        layer.q0_mean.add_(1.0)
        layer.weight_mu.mul_(10)
    g = torch.Generator().manual_seed(seed + 1)
    noise = {"eps_z": torch.randn(B, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g)}
    x = torch.rand(B, I, generator=g)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    return layer.to(dev), p, noise, x


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,O,T,train", [(64, 1200, 130, 3, True), (16, 53, 37, 2, True), (32, 1200, 64, 2, False),
                                           (8, 2500, 20, 1, True)])
def test_planar_backward_all_hip_vs_oracle_autograd(bnn, dev, B, I, O, T, train):
    """V1 + K1b + V2 + GEMMs (no torch autograd inside the layer) against autograd of the oracle in fp64."""
    layer, p, noise, x = _mnf_planar_case(bnn, dev, B, I, O, T, seed=B + I + O)
    layer.train(train)
    layer.noise = {k: v.to(dev) for k, v in noise.items() if train or k in ("eps_z", "eps_out")}
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg, sample=True)
    wgt = torch.randn(B, O, generator=torch.Generator().manual_seed(5)).to(dev)
    loss = (out * wgt).sum() + (layer.kl / 60 if train else 0)
    loss.backward()
    pc = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xc = x.double().requires_grad_(True)
    zf = orc.flow_from_state("z_flow", "Planar", pc, T)
    rf = orc.flow_from_state("r_flow", "Planar", pc, T)
    n64 = {k: v.double() for k, v in noise.items()}
    o, kl, _ = orc.mnf_forward(xc, pc, zf, rf, n64, stochastic=True, compute_kl=train)
    ((o * wgt.cpu().double()).sum() + (kl / 60 if train else 0)).backward()
    assert rel_err(out.detach().cpu().double(), o.detach()) < TOL
    assert rel_err(xg.grad.cpu().double(), xc.grad) < TOL
    for name, prm in layer.named_parameters():
        ref = pc[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        assert rel_err(prm.grad.cpu().double(), ref) < 5e-5, name


@pytest.mark.gpu
def test_lrt_backward_vs_oracle_autograd(bnn, dev):
    B, I, O = 48, 200, 72
    torch.manual_seed(3)
    layer = bnn.lrt.BayesianLinear(I, O)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train()
    g = torch.Generator().manual_seed(4)
    eps, x = torch.randn(B, O, generator=g), torch.rand(B, I, generator=g)
    layer.noise = {"eps_out": eps.to(dev)}
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg)
    ((out ** 2).sum() + layer.kl / 60).backward()
    pc = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xc = x.double().requires_grad_(True)
    o, kl, _ = orc.lrt_forward(xc, pc, eps.double())
    ((o ** 2).sum() + kl / 60).backward()
    assert rel_err(xg.grad.cpu().double(), xc.grad) < TOL
    for name, prm in layer.named_parameters():
        assert rel_err(prm.grad.cpu().double(), pc[name].grad) < 5e-5, name


# --------------------------------------------------------------------------- remaining 1-D flow types (8f-4)
_KIND = {"radial": "Radial", "householder": "Householder", "sylvester": "Sylvester", "mixed": "mixed"}


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["radial", "householder", "sylvester", "mixed"])
def test_flow_chain_vs_reference_golden(bnn, dev, golden, kind):
    """lbbnn_flow_chain against flows2.PropagateFlow outputs (tests/golden/flows_misc.npz), 1-D z."""
    g = golden("flows_misc.npz")
    for ci in range(4):
        for suffix in ("", "s"):
            c = g.case("%s%d%s" % (kind, ci, suffix))
            I, T = [int(v) for v in c["shape"]]
            flow = bnn.flows.PropagateFlow(_KIND[kind], I, T)
            flow.load_state_dict(sub(c, "p."))
            flow = flow.to(dev)
            z, ld = flow(c["z"].to(dev))
            assert rel_err(z, c["z_out"]) < TIGHT, (kind, ci, suffix)
            ref = float(c["logdet"].reshape(-1)[0])
            assert abs(float(ld) - ref) <= 2e-4 * abs(ref) + 2e-6, (kind, ci, suffix, float(ld), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("zk,rk", [("Radial", "Radial"), ("Householder", "Sylvester"), ("mixed", "Planar"),
                                   ("Sylvester", "Householder")])
def test_mnf_layer_with_1d_flow_chains_vs_oracle(bnn, dev, zk, rk):
    """MNF layer whose flows are 1-D chains (row-wise restatement): forward, KL and backward against the oracle
    (fp64 autograd).  The reference itself only runs Radial here (and reduces its norm over all B rows)."""
    B, I, O, T = 24, 96, 40, 2
    torch.manual_seed(11)
    layer = bnn.mnf.BayesianLinear(I, O, T, z_flow_type=zk, r_flow_type=rk)
    with torch.no_grad():
        for fl in (layer.z_flow, layer.r_flow):
            for prm in fl.parameters():
                prm.mul_(8.0)
        layer.q0_mean.add_(1.0)
        layer.weight_mu.mul_(10)
    g = torch.Generator().manual_seed(12)
    noise = {"eps_z": torch.randn(B, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g)}
    x = torch.rand(B, I, generator=g)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train()
    layer.noise = {k: v.to(dev) for k, v in noise.items()}
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg, sample=True)
    (out.pow(2).sum() + layer.kl / 60).backward()
    pc = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xc = x.double().requires_grad_(True)
    zf = orc.flow_from_state("z_flow", zk, pc, len(layer.z_flow.transforms))
    rf = orc.flow_from_state("r_flow", rk, pc, len(layer.r_flow.transforms))
    n64 = {k: v.double() for k, v in noise.items()}
    n64["eps_z"] = n64["eps_z"][-1:]            # row-wise: only the kept row matters
    o, kl, _ = orc.mnf_forward(xc, pc, zf, rf, n64)
    (o.pow(2).sum() + kl / 60).backward()
    assert rel_err(out.detach().cpu().double(), o.detach()) < TOL
    assert abs(float(layer.kl) - float(kl)) / abs(float(kl)) < TOL
    assert rel_err(xg.grad.cpu().double(), xc.grad) < TOL
    for name, prm in layer.named_parameters():
        ref = pc[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) < 1e-12, name
            continue
        assert rel_err(prm.grad.cpu().double(), ref) < 2e-4, name


@pytest.mark.gpu
def test_ensemble_eval_helper(bnn, dev):
    """evaluate.ensemble_eval: shapes, distinct stochastic members, deterministic posterior-mean member equal to
    the oracle's mean-path forward, density in (0,1)."""
    torch.manual_seed(2)
    net = bnn.lrt.BayesianNetwork((784, 64, 48, 10)).to(dev)
    data = torch.rand(100, 1, 28, 28, device=dev)
    target = torch.randint(0, 10, (100,), device=dev)
    r = bnn.evaluate.ensemble_eval(net, data, target, samples=4)
    assert r["outputs"].shape == (4, 100, 10)
    assert not torch.equal(r["outputs"][0], r["outputs"][1])
    assert 0 <= r["correct_ensemble"] <= 100 and 0 <= r["correct_posterior_mean"] <= 100
    assert float(r["density"].min()) > 0.3 and float(r["density"].max()) < 0.9      # alpha = sigmoid(U(0,1))
    P = [{k: v.detach().cpu() for k, v in l.state_dict().items()} for l in (net.l1, net.l2, net.l3)]
    ref, _ = orc.lrt_network_forward(data.cpu(), P, [None] * 3, stochastic=False, compute_kl=False)
    assert torch.equal(ref.argmax(1), r["pred_posterior_mean"].cpu()) or rel_err(net(data).cpu(), ref) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("family,dims,B,S", [("lrt", (784, 400, 600, 10), 1000, 10), ("mnf", (784, 1200, 1200, 10), 1000, 10),
                                             ("mnf", (784, 96, 64, 10), 37, 3), ("lrt", (64, 48, 40, 24), 130, 4)])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_ensemble_batched_equals_loop_bitwise(bnn, dev, family, dims, B, S, prec):
    """test_ensemble's TEST_SAMPLES x net(data, sample=True) on one test batch (LBBNN-GP-MF-MNF.py:286-294, ...LRT.py:241-247;
    reference sizes: 10 members x 1000 rows): the batched form -- ONE K3, ONE K1 and ONE GEMM launch per layer for all
    members (lbbnn_ensemble_operands, lbbnn_lrt_gemm_members) -- equals the loop of fused single forwards BIT FOR BIT under
    the same Philox {seed, offset}, and leaves the offset where the loop leaves it."""
    torch.manual_seed(17)
    if family == "lrt":
        net = bnn.lrt.BayesianNetwork(dims).to(dev)
    else:
        net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev)
    net.eval()
    data = torch.rand(B, dims[0], device=dev)
    st = bnn.ops.RngState.get(dev)
    bnn.set_precision(prec)
    try:
        with torch.no_grad():
            bnn.manual_seed(3, 5)
            loop = torch.stack([net(data, sample=True) for _ in range(S)])
            off_loop = int(st.t[1])
            bnn.manual_seed(3, 5)
            bat = bnn.evaluate.ensemble_forward(net, data, S)
            off_bat = int(st.t[1])
            assert bnn.evaluate._batched_ok(net, data)
    finally:
        bnn.set_precision("fp32")
    assert bat.shape == (S, B, dims[-1]) and off_loop == off_bat == 5 + S
    assert torch.equal(bat, loop)
    assert not torch.equal(bat[0], bat[1])
    r = bnn.evaluate.ensemble_eval(net, data, torch.randint(0, dims[-1], (B,), device=dev), samples=S)
    assert r["outputs"].shape == (S, B, dims[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("B,I,O", [(0, 33, 17), (1, 1, 1), (3, 5, 1), (2, 1, 7), (5, 16383, 3), (1, 2049, 130)])
def test_edge_shapes_vs_oracle(bnn, dev, B, I, O):
    """Empty batch, single row / column / feature, the largest flow dimension: LRT and MNF layers against the oracle
    (in-kernel noise re-created with lbbnn_philox_normal)."""
    from bnn_amd import ops
    torch.manual_seed(B * 7 + I + O)
    for kind in ("lrt", "mnf"):
        if kind == "lrt":
            l = bnn.lrt.BayesianLinear(I, O)
        else:
            l = bnn.mnf.BayesianLinear(I, O, 2, z_flow_type="Planar", r_flow_type="Planar")
        p = {k: v.detach().clone() for k, v in l.state_dict().items()}
        l = l.to(dev).train()
        g = torch.Generator().manual_seed(1)
        x = torch.rand(B, I, generator=g)
        noise = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
                 "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g)}
        l.noise = {k: v.to(dev) for k, v in noise.items()} if kind == "mnf" else {"eps_out": noise["eps_out"].to(dev)}
        with torch.no_grad():
            out = l(x.to(dev), sample=True)
        assert tuple(out.shape) == (B, O)
        if kind == "lrt":
            ref, kl, _ = orc.lrt_forward(x, p, noise["eps_out"])
        else:
            zf = orc.flow_from_state("z_flow", "Planar", p, 2)
            rf = orc.flow_from_state("r_flow", "Planar", p, 2)
            ref, kl, _ = orc.mnf_forward(x, p, zf, rf, noise)
        if B:
            assert rel_err(out.cpu(), ref) < TOL, kind
        assert abs(float(l.kl) - float(kl)) <= TOL * abs(float(kl)), kind


@pytest.mark.gpu
@pytest.mark.parametrize("B,O", [(5, 3), (70, 33), (64, 64), (130, 1200), (257, 132)])
@pytest.mark.parametrize("relu,stoch,explicit", [(True, True, True), (True, True, False), (False, True, False), (True, False, True)])
def test_output_grad_kernel(bnn, dev, B, O, relu, stoch, explicit):
    """lbbnn_output_grad against the torch formulas: G_m, G_v, both transposes, both column sums; eps explicit or
    re-created in-kernel from the Philox state (checked against lbbnn_philox_normal)."""
    from bnn_amd import ops
    g = torch.Generator().manual_seed(B * 31 + O)
    g_out = torch.randn(B, O, generator=g).to(dev)
    out = torch.randn(B, O, generator=g).to(dev)
    std = (0.1 + torch.rand(B, O, generator=g)).to(dev)
    st = ops.RngState.get(dev)
    rng = st.t.clone()
    stream, row_off = ops.STREAM_EPS_OUT * 64 + 5, 1000
    eps = torch.randn(B, O, generator=g).to(dev) if explicit else ops.philox_normal(rng, stream, B, O, row_off)
    gm, gv, gmT, gvT, gs, gvs = ops.output_grad(g_out, out=out if relu else None, std=std if stoch else None,
                                                eps=eps if explicit else None, rng=rng, rng_stream=stream,
                                                row_offset=row_off, relu=relu)
    ref_m = g_out * (out > 0) if relu else g_out
    assert torch.equal(gm, ref_m) and torch.equal(gmT, ref_m.t())
    assert rel_err(gs.cpu().double(), ref_m.double().sum(0).cpu()) < 1e-6
    if stoch:
        ref_v = ref_m * eps / (2 * std)
        assert rel_err(gv, ref_v) < 1e-6 and torch.equal(gvT, gv.t())
        assert rel_err(gvs.cpu().double(), gv.double().sum(0).cpu()) < 1e-6
    else:
        assert gv is None and gvT is None and gvs is None


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(bnn, dev, wd):
    """bnn_amd.optim.Adam (one multi-tensor launch) against torch.optim.Adam on the same gradients, 5 steps, tensors
    of awkward sizes (1 element, unaligned views, > one 4096 chunk)."""
    torch.manual_seed(0)
    shapes = [(1,), (7,), (33, 17), (4097,), (130, 1200), (5,)]
    base = [torch.randn(s, device=dev) for s in shapes]
    pa = [torch.nn.Parameter(t.clone()) for t in base]
    pb = [torch.nn.Parameter(t.clone()) for t in base]
    oa = bnn.optim.Adam(pa, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    for it in range(5):
        gs = [torch.randn(s, device=dev) * (0.1 + it) for s in shapes]
        for p, q, g in zip(pa, pb, gs):
            p.grad, q.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for p, q in zip(pa, pb):
        assert rel_err(p.detach(), q.detach()) < 2e-6
    assert float(oa.param_groups[0]["step_dev"]) == 5.0


@pytest.mark.gpu
def test_fused_adam_checkpoint_interchanges_with_torch_adam(bnn, dev):
    """state_dict() of bnn_amd.optim.Adam carries torch.optim.Adam's per-parameter ``step`` (bias correction continues after
    a restore) and loads into torch.optim.Adam and back; replacing the optimizer state invalidates the cached kernel
    argument lists (a stale list would update freed m / v buffers)."""
    torch.manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(37, 5, device=dev)), torch.nn.Parameter(torch.randn(11, device=dev))]
    pt = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ours, ref = bnn.optim.Adam(ps, lr=1e-2), torch.optim.Adam(pt, lr=1e-2)
    gs = [[torch.randn_like(p) for p in ps] for _ in range(6)]

    def run(opt, params, k0, k1):
        for k in range(k0, k1):
            for p, g in zip(params, gs[k]):
                p.grad = g.clone()
            opt.step()
    run(ours, ps, 0, 3); run(ref, pt, 0, 3)
    sd = ours.state_dict()
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values()) and "step_dev" not in sd["param_groups"][0]
    # restore into a FRESH instance of each kind and continue: all three trajectories must agree
    import copy                                   # (a state_dict holds the live m / v tensors: every consumer gets its own copy)
    ours2 = bnn.optim.Adam(ps, lr=1e-2); ours2.load_state_dict(copy.deepcopy(sd))
    pt2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref2 = torch.optim.Adam(pt2, lr=1e-2); ref2.load_state_dict(copy.deepcopy(sd))
    run(ours2, ps, 3, 6); run(ref, pt, 3, 6); run(ref2, pt2, 3, 6)
    for a, b, c in zip(ps, pt, pt2):
        assert rel_err(a, b) < 1e-6 and rel_err(c, b) < 1e-6
    # same instance: replacing its state must not leave the kernel updating the old buffers
    ours2.load_state_dict(copy.deepcopy(ours2.state_dict()))
    before = [p.detach().clone() for p in ps]
    run(ours2, ps, 0, 1)
    assert all(not torch.equal(a, b) for a, b in zip(before, ps))
    assert all(torch.isfinite(st["exp_avg"]).all() for st in ours2.state.values())


def test_training_with_fused_adam_decreases_loss(bnn, dev):
    torch.manual_seed(0)
    net = bnn.mnf.BayesianNetwork((784, 64, 48, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    opt = bnn.optim.Adam(net.parameters(), lr=1e-3)
    data, target = torch.rand(64, 1, 28, 28, device=dev), torch.randint(0, 10, (64,), device=dev)
    losses = []
    for _ in range(20):
        net.zero_grad()
        loss = torch.nn.functional.nll_loss(net(data, sample=True), target, reduction="sum") + net.kl() / 600
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] and all(math.isfinite(v) for v in losses)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,kchunk", [(10, 4096, 1200, 256), (1200, 4096, 784, 1376), (130, 1000, 96, 320), (33, 2048, 40, 2048)])
def test_matmul_splitk_slabs_sum_to_product(bnn, dev, M, K, N, kchunk):
    """lbbnn_matmul_splitk: the k-range slabs add up to a @ w (fp64 reference), incl. a K tail in the last slab."""
    from bnn_amd import ops
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(K, N, generator=g) * 0.1).to(dev)
    op = ops.transpose_operand(w, split=True)
    slabs = ops.matmul_splitk(a, op, K=K, N=N, kchunk=kchunk)
    assert slabs.shape == ((K + kchunk - 1) // kchunk, M, N)
    ref = a.double().cpu() @ w.double().cpu()
    assert rel_err(slabs.sum(0).cpu().double(), ref) < 2e-5
    # each slab is the product over its own k range
    ref0 = a[:, :kchunk].double().cpu() @ w[:kchunk].double().cpu()
    assert rel_err(slabs[0].cpu().double(), ref0) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("I,O,T", [(96, 40, 2), (1200, 130, 3)])
def test_backward_in_kernel_noise_equals_injected_noise(bnn, dev, I, O, T):
    """Planar MNF layer, training mode, no explicit noise: the backward kernels re-create eps_out / eps_z / eps_z2 /
    eps_act from the forward's Philox state.  Same gradients as a run with those very draws injected as tensors."""
    from bnn_amd import ops
    B = 48
    layer, p, _, x = _mnf_planar_case(bnn, dev, B, I, O, T, seed=3)
    layer.train()
    x = x.to(dev)
    bnn.manual_seed(1234, 7)                     # torch seed too: RngState.get() follows torch.initial_seed()
    snap = ops.RngState.get(x.device).t.clone()
    layer.noise = None
    out_a = layer(x.clone().requires_grad_(True), sample=True)
    (out_a.pow(2).sum() + layer.kl / 50).backward()
    grads_a = {n: q.grad.clone() for n, q in layer.named_parameters()}
    L = layer._layer_id
    layer.noise = {"eps_out": ops.philox_normal(snap, ops.STREAM_EPS_OUT * 64 + L, B, O, 0),
                   "eps_z": ops.philox_normal(snap, ops.STREAM_EPS_Z * 64 + L, 0, I),
                   "eps_z2": ops.philox_normal(snap, ops.STREAM_EPS_Z2 * 64 + L, 0, I),
                   "eps_act": ops.philox_normal(snap, ops.STREAM_EPS_ACT * 64 + L, 0, O)}
    layer.zero_grad()
    out_b = layer(x.clone().requires_grad_(True), sample=True)
    (out_b.pow(2).sum() + layer.kl / 50).backward()
    assert torch.equal(out_a, out_b)
    for n, q in layer.named_parameters():
        assert rel_err(grads_a[n], q.grad) < 1e-6, n


@pytest.mark.gpu
def test_data_parallel_bucket_path_equals_plain_step(bnn, dev):
    """World size 1 on the GPU: pack (lbbnn_multi_copy) -> [all-reduce = identity] -> Adam reading the flat bucket
    gives the same parameters as the plain loss.backward(); opt.step(); unpack restores p.grad.  (Same network object
    for both runs: the in-kernel noise streams are keyed by layer id.)"""
    import copy
    from bnn_amd.parallel import DataParallelELBO
    torch.manual_seed(5)
    net = bnn.mnf.BayesianNetwork((784, 64, 48, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    init = copy.deepcopy(net.state_dict())
    data, target = torch.rand(32, 1, 28, 28, device=dev), torch.randint(0, 10, (32,), device=dev)
    finals = []
    for mode in ("plain", "bucket"):
        net.load_state_dict(init)
        opt = bnn.optim.Adam(net.parameters(), lr=1e-3)
        dp = DataParallelELBO(net)
        for it in range(3):
            bnn.manual_seed(100 + it)                       # same draws in both runs
            net.zero_grad()
            if mode == "plain":
                loss = torch.nn.functional.nll_loss(net(data, sample=True), target, reduction="sum") + net.kl() / 10
                loss.backward()
                opt.step()
            else:
                x_r, y_r = dp.shard(data, target)
                loss = dp.loss(net(x_r, sample=True), y_r, num_batches=10)
                loss.backward()
                dp.all_reduce_grads(unpack=(it == 0))        # the first step also exercises the unpack launch
                if it == 0:
                    for p, v in zip(*dp.reduced_grads()):
                        assert torch.equal(p.grad, v)
                opt.step(grads=dp.reduced_grads())
        finals.append({k: v.detach().clone() for k, v in net.named_parameters()})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_row_sharded_forward_equals_full_batch_bitwise(bnn, dev, prec):
    """The data-parallel contract on ONE GPU (LBBNN-GP-MF-MNF.py:197-200): a 4096-row ELBO forward equals
    the two 2048-row shard forwards run with set_row_offset(0 / 2048) from the same Philox {seed, offset} -- the SAME z
    on every shard (so the same KL, bit for bit) and eps drawn by GLOBAL row index (so the same activations, bit for
    bit).  In-kernel noise, headline net, both GEMM precisions."""
    dims, B = (784, 1200, 1200, 10), 4096
    torch.manual_seed(5)
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    x = torch.rand(B, 784, device=dev)
    bnn.set_precision(prec)
    try:
        with torch.no_grad():
            bnn.manual_seed(99, 3)
            net.set_row_offset(0)
            full = net(x, sample=True).clone()
            kl_full = net.kl().clone()
            parts, kls = [], []
            for lo in (0, B // 2):
                bnn.manual_seed(99, 3)                    # every rank holds the same {seed, offset}
                net.set_row_offset(lo)
                parts.append(net(x[lo:lo + B // 2], sample=True).clone())
                kls.append(net.kl().clone())
            net.set_row_offset(0)
            # and the offsets matter: shard 1 run with row_offset 0 draws shard 0's eps
            bnn.manual_seed(99, 3)
            wrong = net(x[B // 2:], sample=True)
    finally:
        bnn.set_precision("fp32")
    assert torch.equal(torch.cat(parts), full)
    assert torch.equal(kls[0], kl_full) and torch.equal(kls[1], kl_full)
    assert not torch.equal(wrong, full[B // 2:])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RNVP", "MNF"])
def test_reference_default_flow_headline_network_vs_oracle_full_size(bnn, dev, kind):
    """The reference's DEFAULT flow type (RNVP, LBBNN-GP-MF-MNF.py:46-47; and the 'MNF' type) at the headline size
    784-1200-1200-10, B = 4096, injected draws and masks, against the fp64 oracle -- both GEMM precisions; outputs
    (global norm and element-wise), network KL and every layer's KL."""
    dims, B, T = (784, 1200, 1200, 10), 4096, 2
    torch.manual_seed(41)
    net = bnn.mnf.BayesianNetwork(dims, T, z_flow_type=kind, r_flow_type=kind)
    layers = [net.l1, net.l2, net.l3]
    g = torch.Generator().manual_seed(42)
    x = torch.rand(B, dims[0], generator=g)
    noises, P, zf, rf = [], [], [], []
    for l in layers:
        I, O = l.in_features, l.out_features
        bern = lambda: torch.bernoulli(torch.full((I,), 0.5), generator=g)
        n = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g),
             "zmask": [bern() for _ in range(T)], "zmask2": [bern() for _ in range(T)], "rmask": [bern() for _ in range(T)]}
        noises.append(n)
        sd = {k: v.detach().clone() for k, v in l.state_dict().items()}
        P.append(sd)
        zf.append(orc.flow_from_state("z_flow", kind, sd, T))
        rf.append(orc.flow_from_state("r_flow", kind, sd, T))
    ref_out, ref_kls = _oracle_mnf_net(*_to64(x, P, zf, rf, noises))
    net = net.to(dev).train()
    for l, n in zip(layers, noises):
        l.noise = {k: ([m.to(dev) for m in v] if isinstance(v, list) else v.to(dev)) for k, v in n.items()}
    for prec, atol_frac in (("fp32", 1e-6), ("bf16x3", 1e-5)):
        bnn.set_precision(prec)
        try:
            with torch.no_grad():
                out = net(x.to(dev), sample=True)
                kl = net.kl()
        finally:
            bnn.set_precision("fp32")
        assert rel_err(out, ref_out) < TOL, prec
        v = elementwise_violation(out, ref_out, rtol=1e-4, atol_frac=atol_frac)
        assert v <= 1.0, (prec, v)
        assert rel_err(kl, sum(ref_kls)) < TOL, prec
        for l, k_ref in zip(layers, ref_kls):
            assert rel_err(l.kl, k_ref) < TOL, prec


@pytest.mark.gpu
def test_vd_config4_reduced_precision_fp16_and_bf16(bnn, dev):
    """BASELINE configs[4] "fp16 MFMA": the variational-dropout net 3072-4096-4096-10 at B = 1024 with ONE fp16 product per
    moment (set_precision('fp16'): lbbnn_vd_operands(LBBNN_F_HALF16) + v_mfma_f32_16x16x32_f16), and with one bf16 product
    ('bf16').  Reduced-precision modes with their own, measured bars against the fp64 oracle (max norm on the layer-1
    activations and on the log-probabilities): fp16 < 2e-3, bf16 < 2e-2, both demonstrably outside the 1e-4 contract."""
    dims, B = (3072, 4096, 4096, 10), 1024
    torch.manual_seed(6)
    layers = [bnn.vd.BayesianLayer(dims[i], dims[i + 1]).to(dev) for i in range(3)]
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.rand(B, dims[0], device=dev, generator=g)
    zetas = [torch.randn(B, dims[i + 1], device=dev, generator=g) for i in range(3)]
    h = x.double()
    ref1 = None
    for i, l in enumerate(layers):
        h = orc.vd_forward(h, l.theta.detach().double(), l.alpha.double(), zetas[i].double())
        if i == 0:
            ref1 = h.clone()
        if i < 2:
            h = torch.relu(h)
    ref = torch.log_softmax(h, dim=1)
    errs = {}
    for prec in ("bf16x3", "fp16", "bf16"):
        bnn.set_precision(prec)
        try:
            with torch.no_grad():
                h = x
                for i, l in enumerate(layers):
                    l.noise = {"zeta": zetas[i]}
                    h = l(h)
                    if i == 0:
                        e1 = rel_err(h, ref1)
                    if i < 2:
                        h = torch.relu(h)
                out = torch.log_softmax(h, dim=1)
        finally:
            bnn.set_precision("fp32")
        errs[prec] = (e1, rel_err(out, ref))
    assert errs["bf16x3"][0] < 2e-5 and errs["bf16x3"][1] < 2e-5, errs
    assert 2e-5 < errs["fp16"][0] < 2e-3 and errs["fp16"][1] < 2e-3, errs
    assert 1e-4 < errs["bf16"][0] < 2e-2 and errs["bf16"][1] < 2e-2, errs
    assert errs["fp16"][0] < errs["bf16"][0]


@pytest.mark.gpu
def test_vd_config4_whole_network_full_size_vs_fp64(bnn, dev):
    """BASELINE configs[4] as a whole: variational-dropout layers 3072-4096-4096-10 with ReLU between them
    (variational_dropout.py:63-68,80-86) at B = 4096, injected zeta, against the oracle run in fp64 (on the GPU: three
    103-GFLOP fp64 products are minutes on the host cores) -- both GEMM precisions, global norm and element-wise."""
    dims, B = (3072, 4096, 4096, 10), 4096
    torch.manual_seed(6)
    layers = [bnn.vd.BayesianLayer(dims[i], dims[i + 1]).to(dev) for i in range(3)]
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.rand(B, dims[0], device=dev, generator=g)
    zetas = [torch.randn(B, dims[i + 1], device=dev, generator=g) for i in range(3)]
    h = x.double()
    for i, l in enumerate(layers):
        h = orc.vd_forward(h, l.theta.detach().double(), l.alpha.double(), zetas[i].double())
        if i < 2:
            h = torch.relu(h)
    ref = torch.log_softmax(h, dim=1).cpu()
    del h
    for prec, bar, atol_frac in (("fp32", TIGHT, 1e-6), ("bf16x3", 2e-5, 1e-5)):
        bnn.set_precision(prec)
        try:
            with torch.no_grad():
                h = x
                for i, l in enumerate(layers):
                    l.noise = {"zeta": zetas[i]}
                    h = l(h)
                    if i < 2:
                        h = torch.relu(h)
                out = torch.log_softmax(h, dim=1)
        finally:
            bnn.set_precision("fp32")
        assert rel_err(out, ref) < bar, prec
        v = elementwise_violation(out, ref, rtol=1e-4, atol_frac=atol_frac)
        assert v <= 1.0, (prec, v)


def _flow_with_state(bnn, kind, I, T, c, dev):
    flow = bnn.flows.PropagateFlow(kind, I, T)
    flow.load_state_dict({k: v for k, v in sub(c, "p.").items()})
    return flow.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["rnvp0", "rnvp1", "mnf0", "mnf1"])
def test_propagate_flow_dense_standalone_vs_reference_golden(bnn, dev, golden, case):
    """Stand-alone ``PropagateFlow('RNVP'|'MNF', dim, T)(z)`` on an (R,I) z -- flows2.py:41-46,206-219,233-241 -- against
    the REFERENCE's own outputs (tests/golden/flows.npz, recorded masks injected): z (R,I), logdet (R,) for RNVP and 0-d
    for the MNF type, through lbbnn_flow_dense_rows (MFMA); then the per-transform forward(z) / log_det() protocol."""
    c = golden("flows.npz").case(case)
    R, I, T = [int(v) for v in c["shape"]]
    kind = "RNVP" if case.startswith("rnvp") else "MNF"
    flow = _flow_with_state(bnn, kind, I, T, c, dev)
    flow.masks = [c["mask%d" % t].to(dev) for t in range(T)]
    z = c["z"].to(dev)
    out, ld = flow(z)
    assert out.shape == c["z_out"].shape and ld.shape == c["logdet"].shape
    assert rel_err(out, c["z_out"]) < TIGHT
    assert rel_err(ld, c["logdet"]) < 2e-5
    # the reference's loop: z = f(z); logdet += f.log_det()
    zz, tot = z, 0
    for t, f in enumerate(flow.transforms):
        f.mask_in = c["mask%d" % t].to(dev)
        zz = f(zz)
        tot = tot + f.log_det()
        m_attr = f.mask if kind == "RNVP" else f.m
        assert torch.equal(m_attr.cpu(), c["mask%d" % t])
    assert torch.equal(zz, out) and rel_err(tot, c["logdet"]) < 2e-5
    # 1-D z (what r_flow(z2) passes, LBBNN-GP-MF-MNF.py:222): row 0 alone
    flow.masks = [c["mask%d" % t][0].to(dev) for t in range(T)]
    o1, l1 = flow(z[0])
    assert o1.shape == (I,) and l1.dim() == 0
    ref = orc.flow_from_state("x", kind, {"x." + k: v for k, v in sub(c, "p.").items()}, T).run(
        c["z"][0], [c["mask%d" % t][0] for t in range(T)])
    assert rel_err(o1, ref[0]) < TIGHT and rel_err(l1, ref[1]) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kind,R,I,T", [("RNVP", 4096, 1200, 2), ("MNF", 4096, 784, 2), ("RNVP", 37, 50, 3), ("MNF", 1, 7, 1),
                                        ("RNVP", 16, 1760, 1)])
def test_flow_dense_rows_in_kernel_masks_vs_oracle(bnn, dev, kind, R, I, T):
    """The as-written row mode at the headline size (B = 4096 rows, I = 1200 / 784; SURVEY.md section 7 'R = B') and odd
    shapes: masks drawn in-kernel (Philox), returned, and fed to the oracle's row-batched restatement."""
    torch.manual_seed(3)
    flow = bnn.flows.PropagateFlow(kind, I, T).to(dev)
    for p in flow.parameters():
        p.requires_grad_(False)                           # evaluation: the one-launch row kernel (a required gradient
        if p.dim() == 2:                                  # would route the rows through the differentiable 1-D form)
            p.mul_(1.5)                                   # gates away from 0.5, log-dets of useful size
    g = torch.Generator().manual_seed(4)
    z = torch.randn(R, I, generator=g)
    flow.keep_masks = True
    bnn.manual_seed(5, 1)
    out, ld = flow(z.to(dev))
    assert not out.requires_grad and torch.isfinite(out).all() and torch.isfinite(ld).all()
    m = flow.last_masks
    assert m.shape == (T, R, I) and bool(((m == 0) | (m == 1)).all())
    assert abs(float(m.mean()) - 0.5) < (0.02 if m.numel() > 10000 else 0.5)
    if R > 1:
        assert not torch.equal(m[0, 0], m[0, 1])          # every row has its own mask (flows2.py:209)
    bnn.manual_seed(5, 1)
    out2, _ = flow(z.to(dev))
    assert torch.equal(out, out2)                         # deterministic in {seed, offset}
    sd = {"x." + k: v.detach().cpu() for k, v in flow.state_dict().items()}
    ref_z, ref_ld = orc.flow_from_state("x", kind, sd, T).run(z, [m[t].cpu() for t in range(T)])
    assert rel_err(out, ref_z) < 2e-5
    assert ld.shape == ref_ld.shape and rel_err(ld, ref_ld) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["Planar", "Radial", "Householder", "Sylvester", "mixed"])
def test_propagate_flow_vector_kinds_rows_and_transform_protocol(bnn, dev, kind):
    """1-D flow kinds on an (R,I) z: row r of the result is the 1-D flow of row r (lbbnn_flow_chain_rows; the reference
    itself raises on a 2-D z for all but Radial), and f(z) / f.log_det() per transform."""
    torch.manual_seed(8)
    R, I, T = 9, 45, 3
    flow = bnn.flows.PropagateFlow(kind, I, T).to(dev)
    z = 0.3 * torch.randn(R, I, generator=torch.Generator().manual_seed(9))
    out, ld = flow(z.to(dev))
    assert out.shape == (R, I) and ld.shape == (R,)
    sd = {"x." + k: v.detach().cpu() for k, v in flow.state_dict().items()}
    of = orc.flow_from_state("x", kind, sd, len(flow.transforms))
    for r in (0, 4, R - 1):
        zr, lr = of.run(z[r])
        assert rel_err(out[r], zr) < TIGHT
        assert abs(float(ld[r]) - float(lr)) < 2e-5 * max(1.0, abs(float(lr)))
        o1, l1 = flow(z[r].to(dev))
        assert torch.equal(o1, out[r]) and abs(float(l1.reshape(-1)[0]) - float(ld[r])) < 1e-6 * max(1.0, abs(float(ld[r])))
    zz, tot = z[2].to(dev), 0
    for f in flow.transforms:
        zz = f(zz)
        tot = tot + f.log_det()
    assert rel_err(zz, out[2]) < TIGHT
    assert abs(float(torch.as_tensor(tot).reshape(-1)[0]) - float(ld[2])) < 2e-5 * max(1.0, abs(float(ld[2])))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RNVP", "MNF", "Planar", "Sylvester"])
def test_propagate_flow_standalone_autograd_1d(bnn, dev, kind):
    """A 1-D z through the stand-alone flow WITH autograd (dense kinds: lbbnn_flow_dense_apply[_backward]): gradients of
    sum(z') + 0.7 logdet w.r.t. z and every flow parameter against fp64 autograd of the oracle."""
    torch.manual_seed(10)
    I, T = 40, 2
    flow = bnn.flows.PropagateFlow(kind, I, T).to(dev)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(I, generator=g)
    masks = [torch.bernoulli(torch.full((I,), 0.5), generator=g) for _ in range(T)] if kind in ("RNVP", "MNF") else None
    flow.masks = [m.to(dev) for m in masks] if masks else None
    zd = z.to(dev).requires_grad_(True)
    out, ld = flow(zd)
    (out.sum() + 0.7 * ld.sum()).backward()
    sd = {"x." + k: v.detach().cpu().double().requires_grad_(True) for k, v in flow.state_dict().items()}
    z64 = z.double().requires_grad_(True)
    ro, rl = orc.flow_from_state("x", kind, sd, T).run(z64, [m.double() for m in masks] if masks else None)
    (ro.sum() + 0.7 * torch.as_tensor(rl).sum()).backward()
    assert rel_err(out, ro) < TIGHT
    assert rel_err(zd.grad, z64.grad) < 2e-4
    for k, p in flow.named_parameters():
        if sd["x." + k].grad is not None:
            assert rel_err(p.grad, sd["x." + k].grad) < 2e-4, k
    # an (R,I) z with a gradient required: row by row through the same differentiable form; no_grad: the one-launch form
    z2 = torch.randn(3, I, generator=g)
    flow.masks = [torch.bernoulli(torch.full((3, I), 0.5), generator=g).to(dev) for _ in range(T)] if masks else None
    og, lg = flow(z2.to(dev))
    with torch.no_grad():
        on, ln = flow(z2.to(dev))
    assert og.requires_grad and rel_err(og, on) < TIGHT
    assert float((lg - ln).abs().max()) < 1e-6 + 2e-5 * float(ln.abs().max())


@pytest.mark.gpu
def test_bench_contract_json_line():
    """bench.py, run exactly as the driver runs it (--gpus 1 --steps 20 --warmup 5; only the CPU-baseline budget is cut):
    ONE JSON line with the contract keys, a roofline that is consistent with the timed region, an fp32 secondary leg,
    and a headline within 25 % of the committed reference run of the same command (profiles/r03_bench_driver_cmd.json: round
    3's row-scaled fp16 format, 27.9 M samples/s; round 2's bf16x3 line read 24.7-24.9 M) -- round 1's driver line (2.05 ms/step
    against 0.18) fails here.  The line also carries the strict 3 + 3 product leg and roofline.frac_rocprof, the same fraction
    computed from the COMMITTED rocprofv3 summary of this command."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--cpu-seconds", "2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "settle", "secondary",
              "ms_per_step_median", "ms_per_step_min", "ms_per_step_max"):
        assert k in d, (k, d.get("roofline_invalid"))
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "gemm_share_of_step", "avg_launch_us", "sampled_in"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # self-consistency: the bracketed GEMMs are part of a step, so they cannot take longer than one
    assert 0.3 < rf["gemm_share_of_step"] <= 1.0, rf
    assert rf["avg_launch_us"] < d["ms_per_step"] * 1e3
    assert "after the timed region" in rf["sampled_in"]
    # the timed region has no outlier step hiding in the mean
    assert d["ms_per_step_max"] < 3.0 * d["ms_per_step_median"], d
    assert abs(d["ms_per_step"] - d["ms_per_step_median"]) < 0.25 * d["ms_per_step_median"], d
    # within 25 % of the committed run of the same command (recorded launch plan: ~30 us of host time per step)
    assert d["launch"].startswith("recorded launch plan") and d["launch_fallback_reason"] is None
    assert d["precision"] == "fp16x3f" and d["dtype"].startswith("f16")
    assert "frac_rocprof" in rf and 0.5 * rf["frac"] < rf["frac_rocprof"] < 2.0 * rf["frac"], rf
    st3 = d["secondary_strict_fp16x3"]
    assert st3["precision"] == "fp16x3" and 0.5 * d["value"] < st3["value"] <= 1.1 * d["value"], st3
    ref_path = os.path.join(root, "profiles", "r03_bench_driver_cmd.json")
    ref = json.loads(open(ref_path).read().strip().splitlines()[-1])
    assert 0.75 * ref["value"] < d["value"] < 1.33 * ref["value"], (d["value"], ref["value"], d)
    sec = d["secondary"]
    assert sec["dtype"] == "f32" and sec["roofline"]["peak"] == 157.3 and 0.2 < sec["roofline"]["frac"] <= 1.0
    assert 0.5 * ref["secondary"]["value"] < sec["value"] < 2.0 * ref["secondary"]["value"], sec
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind,B,I,O,T", [("RNVP", 16, 96, 40, 2), ("MNF", 16, 96, 40, 2), ("RNVP", 32, 1200, 64, 2),
                                          ("MNF", 8, 1200, 32, 1)])
@pytest.mark.parametrize("mode", ["layer_hip", "single_wg", "torch"])
def test_dense_flow_hip_backward_vs_oracle_autograd(bnn, dev, kind, B, I, O, T, mode, monkeypatch):
    """MNF layer with RNVP / MNF-type flows: forward, KL and every gradient against fp64 autograd of the oracle, with the
    vector-sized chain differentiated by (layer_hip, the default) lbbnn_mnf_flow_dense_backward on the intermediates the
    forward kept, (single_wg) torch plus one HIP launch per flow application (lbbnn_flow_dense_apply[_backward]), or
    (torch) torch formulas only."""
    from bnn_amd import _grad, layers
    monkeypatch.setattr(layers, "_DENSE_HIP_BWD", mode == "layer_hip")
    monkeypatch.setattr(_grad, "_DENSE_HIP", mode == "single_wg")
    torch.manual_seed(21)
    layer = bnn.mnf.BayesianLinear(I, O, T, z_flow_type=kind, r_flow_type=kind)
    with torch.no_grad():
        layer.q0_mean.add_(1.0)
        layer.weight_mu.mul_(10)
    g = torch.Generator().manual_seed(22)
    bern = lambda: torch.bernoulli(torch.full((I,), 0.5), generator=g)
    noise = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g),
             "zmask": [bern() for _ in range(T)], "zmask2": [bern() for _ in range(T)], "rmask": [bern() for _ in range(T)]}
    x = torch.rand(B, I, generator=g)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train()
    layer.noise = {k: ([m.to(dev) for m in v] if isinstance(v, list) else v.to(dev)) for k, v in noise.items()}
    xg = x.to(dev).requires_grad_(True)
    out = layer(xg, sample=True)
    (out.pow(2).sum() + layer.kl / 60).backward()
    pc = {k: v.double().requires_grad_(True) for k, v in p.items()}
    xc = x.double().requires_grad_(True)
    zf = orc.flow_from_state("z_flow", kind, pc, T)
    rf = orc.flow_from_state("r_flow", kind, pc, T)
    n64 = {k: ([m.double() for m in v] if isinstance(v, list) else v.double()) for k, v in noise.items()}
    o, kl, _ = orc.mnf_forward(xc, pc, zf, rf, n64)
    (o.pow(2).sum() + kl / 60).backward()
    assert rel_err(out.detach().cpu().double(), o.detach()) < TOL
    assert abs(float(layer.kl) - float(kl)) / abs(float(kl)) < TOL
    assert rel_err(xg.grad.cpu().double(), xc.grad) < TOL
    for name, prm in layer.named_parameters():
        ref = pc[name].grad
        if ref is None or float(ref.abs().max()) == 0.0:
            assert prm.grad is None or float(prm.grad.abs().max()) < 1e-12, name
            continue
        assert rel_err(prm.grad.cpu().double(), ref) < 5e-4, name


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RNVP", "MNF", "Householder"])
def test_fused_network_forward_non_planar_flows_vs_oracle(bnn, dev, kind):
    """No-grad network forward with dense (K4 batched over the layers) or 1-D chain flows, then K1 with flows_done:
    outputs and KL against the oracle network with the same draws and masks."""
    torch.manual_seed(31)
    dims, B, T = (96, 72, 40, 10), 24, 2
    net = bnn.mnf.BayesianNetwork(dims, T, z_flow_type=kind, r_flow_type=kind).to(dev).train()
    g = torch.Generator().manual_seed(32)
    x = torch.rand(B, dims[0], generator=g)
    layers = [net.l1, net.l2, net.l3]
    noises, P, zf, rf = [], [], [], []
    for l in layers:
        I, O = l.in_features, l.out_features
        n = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g)}
        if kind in ("RNVP", "MNF"):
            bern = lambda: torch.bernoulli(torch.full((I,), 0.5), generator=g)
            n.update(zmask=[bern() for _ in range(T)], zmask2=[bern() for _ in range(T)], rmask=[bern() for _ in range(T)])
        noises.append(n)
        l.noise = {k: ([m.to(dev) for m in v] if isinstance(v, list) else v.to(dev)) for k, v in n.items()}
        sd = {k: v.detach().cpu() for k, v in l.state_dict().items()}
        P.append(sd)
        zf.append(orc.flow_from_state("z_flow", kind, sd, len(l.z_flow.transforms)))
        rf.append(orc.flow_from_state("r_flow", kind, sd, len(l.r_flow.transforms)))
    with torch.no_grad():
        out = net(x.to(dev), sample=True)
        kl = net.kl()
    ref_out, ref_kl = orc.mnf_network_forward(x, P, zf, rf, noises)
    assert rel_err(out.cpu(), ref_out) < TOL
    assert abs(float(kl) - float(ref_kl)) / abs(float(ref_kl)) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("kind,want_kl", [("RNVP", True), ("MNF", True), ("RNVP", False)])
def test_dense_flow_hip_backward_in_kernel_noise(bnn, dev, kind, want_kl, monkeypatch):
    """In-kernel Philox draws + device-drawn masks: lbbnn_mnf_flow_dense_backward (draws re-created in the kernel) against
    the torch-autograd chain on the same draws (re-created as tensors), same seeds; eval-mode sample=True has no KL branch
    (r-flow gradients zero-filled)."""
    from bnn_amd import layers
    B, I, O, T = 24, 200, 48, 2
    torch.manual_seed(5)
    layer = bnn.mnf.BayesianLinear(I, O, T, z_flow_type=kind, r_flow_type=kind).to(dev)
    layer.train(want_kl)
    x = torch.rand(B, I, device=dev)
    grads = {}
    for mode in ("hip", "torch"):
        monkeypatch.setattr(layers, "_DENSE_HIP_BWD", mode == "hip")
        bnn.manual_seed(77, 3)                                # also seeds torch's device generator (the masks)
        layer.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        out = layer(xg, sample=True)
        loss = out.pow(2).sum() + (layer.kl / 60 if want_kl else 0)
        loss.backward()
        grads[mode] = {"x": xg.grad.clone(), **{n: (p.grad.clone() if p.grad is not None else None) for n, p in layer.named_parameters()}}
    for n, g in grads["hip"].items():
        ref = grads["torch"][n]
        if ref is None or float(ref.abs().max()) == 0.0:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        assert rel_err(g.cpu(), ref.cpu()) < 1e-4, n


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RNVP", "MNF"])
def test_dense_network_training_forward_batched_flows(bnn, dev, kind, monkeypatch):
    """Training step of a 3-layer net with dense flows, in-kernel noise: the default path (all layers' flows in one batched
    launch sequence before the per-layer autograd forwards, lbbnn_mnf_flow_dense_backward on what they kept) against the
    per-layer torch-autograd chain on the same draws (shared RNG offset, same seeds): loss and every gradient."""
    from bnn_amd import layers
    torch.manual_seed(3)
    net = bnn.mnf.BayesianNetwork((40, 56, 24, 10), 2, z_flow_type=kind, r_flow_type=kind).to(dev).train()
    x = torch.rand(32, 40, device=dev)
    y = torch.randint(0, 10, (32,), device=dev)
    res = {}
    for mode in ("hip", "torch"):
        monkeypatch.setattr(layers, "_DENSE_HIP_BWD", mode == "hip")
        bnn.manual_seed(13, 5)                                 # also seeds torch's generator (the masks)
        net.zero_grad(set_to_none=True)
        out = net(x, sample=True)
        loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + net.kl() / 10
        loss.backward()
        res[mode] = (float(loss.detach()), {n: p.grad.clone() for n, p in net.named_parameters()})
        off = int(bnn.ops.RngState.get(dev).t[1])
        assert off == 6, off                                   # one shared offset, advanced once per network forward
    assert abs(res["hip"][0] - res["torch"][0]) / abs(res["torch"][0]) < 1e-5
    for n, g in res["hip"][1].items():
        # (5e-4: two fp32 evaluation orders of four chained 75x75 MLP gradients; measured up to 3.3e-4 on q0_mean)
            assert rel_err(g.cpu(), res["torch"][1][n].cpu()) < 5e-4, n


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RNVP", "Planar"])
def test_vector_backward_overlap_same_gradients(bnn, dev, kind):
    """layers.vector_backward_overlap(): the vector-sized backward chains run on a side stream (forked after K1b, joined
    on exit) -- same kernels, same inputs, so every gradient must be bitwise what the single-stream backward gives."""
    from bnn_amd import layers
    torch.manual_seed(4)
    net = bnn.mnf.BayesianNetwork((48, 64, 40, 10), 2, z_flow_type=kind, r_flow_type=kind).to(dev).train()
    x = torch.rand(64, 48, device=dev)
    y = torch.randint(0, 10, (64,), device=dev)
    res = []
    for overlap in (False, True):
        bnn.manual_seed(21, 2)
        net.zero_grad(set_to_none=True)
        loss = torch.nn.functional.nll_loss(net(x, sample=True), y, reduction="sum") + net.kl() / 10
        if overlap:
            with layers.vector_backward_overlap():
                loss.backward()
        else:
            loss.backward()
        torch.cuda.synchronize()
        res.append({n: p.grad.clone() for n, p in net.named_parameters()})
    for n in res[0]:
        assert torch.equal(res[0][n], res[1][n]), n


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["RNVP", "Planar"])
@pytest.mark.parametrize("case", ["two_forwards", "tensor_hook", "frozen_vector_param", "grad_already_set"])
def test_vector_backward_overlap_falls_back_when_adoption_is_not_certain(bnn, dev, kind, case):
    """The deferred vector chains write into the tensors backward() returned, which is sound only if autograd adopts each
    of them as .grad.  When that is not certain -- the layer runs twice in the graph (2-sample ELBO: autograd SUMS the two
    returned tensors before any flush), a gradient hook reads the tensor, a vector parameter is frozen (its gradient is
    dropped), or .grad is already set -- the chain must run in place: gradients bitwise equal to the plain backward."""
    from bnn_amd import layers
    torch.manual_seed(4)
    net = bnn.mnf.BayesianNetwork((48, 64, 40, 10), 2, z_flow_type=kind, r_flow_type=kind).to(dev).train()
    x = torch.rand(64, 48, device=dev)
    y = torch.randint(0, 10, (64,), device=dev)
    seen = []
    if case == "tensor_hook":
        net.l2.q0_mean.register_hook(lambda g: seen.append(float(g.abs().sum())) or None)
    if case == "frozen_vector_param":
        net.l1.r0_b1.requires_grad_(False)

    def loss_fn():
        n_s = 2 if case == "two_forwards" else 1
        tot = 0
        for _ in range(n_s):
            tot = tot + torch.nn.functional.nll_loss(net(x, sample=True), y, reduction="sum") + net.kl() / 10
        return tot / n_s

    res = []
    for overlap in (False, True):
        bnn.manual_seed(21, 2)
        net.zero_grad(set_to_none=True)
        if case == "grad_already_set":
            for p in net.parameters():
                p.grad = torch.full_like(p, 0.25)
        loss = loss_fn()
        if overlap:
            with layers.vector_backward_overlap():
                loss.backward()
            assert not layers._OVERLAP["planar"] and not layers._OVERLAP["dense"] and not layers._OVERLAP["adopt"]
        else:
            loss.backward()
        torch.cuda.synchronize()
        res.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    assert res[0].keys() == res[1].keys()
    for n in res[0]:
        assert torch.isfinite(res[0][n]).all(), n
        assert torch.equal(res[0][n], res[1][n]), n
    if case == "tensor_hook":
        assert len(seen) == 2 and seen[0] == seen[1] and seen[0] > 0     # the hook saw the finished gradient both times


@pytest.mark.gpu
def test_vector_backward_overlap_drops_pending_chains_on_exception(bnn, dev):
    """An exception inside the backward pass: the context manager must not launch the chains filed so far (autograd has
    already released what they would write); the next, clean step gives the plain backward's gradients."""
    from bnn_amd import layers
    torch.manual_seed(4)
    net = bnn.mnf.BayesianNetwork((48, 64, 40, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    x = torch.rand(64, 48, device=dev)
    y = torch.randint(0, 10, (64,), device=dev)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise ValueError("boom")

    net.zero_grad(set_to_none=True)
    xr = x.clone().requires_grad_(True)
    out = net(Boom.apply(xr), sample=True)          # l3, l2, l1 run their backward first, then Boom raises
    loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + net.kl() / 10
    with pytest.raises((ValueError, RuntimeError)):
        with layers.vector_backward_overlap():
            loss.backward()
    assert not layers._OVERLAP["planar"] and not layers._OVERLAP["adopt"] and not layers._OVERLAP["on"]
    torch.cuda.synchronize()
    res = []
    for overlap in (False, True):
        bnn.manual_seed(21, 2)
        net.zero_grad(set_to_none=True)
        loss = torch.nn.functional.nll_loss(net(x, sample=True), y, reduction="sum") + net.kl() / 10
        if overlap:
            with layers.vector_backward_overlap():
                loss.backward()
        else:
            loss.backward()
        torch.cuda.synchronize()
        res.append({n: p.grad.clone() for n, p in net.named_parameters()})
    for n in res[0]:
        assert torch.equal(res[0][n], res[1][n]), n


@pytest.mark.gpu
@pytest.mark.parametrize("O,I,split,mnf", [(96, 160, True, True), (96, 160, False, True), (50, 77, False, False), (1200, 784, True, True)])
def test_weight_operands_t_matches_weight_pass_plus_transpose(bnn, dev, O, I, split, mnf):
    """lbbnn_weight_operands_t ((e_w z)^T | var_w^T from the parameters in one pass) is bit-identical to the path it
    replaces, lbbnn_weight_pass (fp32 operands) + 2 x lbbnn_transpose_operand, in both operand formats -- compared through
    the GEMM that consumes them (same operand bits => same product bits)."""
    from bnn_amd import ops, _lib
    g = torch.Generator().manual_seed(8)
    mu = (0.2 * torch.randn(O, I, generator=g)).to(dev)
    rho = (-4.5 + 0.3 * torch.randn(O, I, generator=g)).to(dev)
    lam = torch.randn(O, I, generator=g).to(dev)
    z = (1 + 0.1 * torch.randn(I, generator=g)).to(dev) if mnf else None
    ld_i = ops.operand_ld(I)
    e_w, v_w = torch.empty(O, ld_i, device=dev), torch.empty(O, ld_i, device=dev)
    ops.weight_pass(mu, rho, lam, z_fwd=z, priors=bnn.Priors(), e_w=e_w, var_w=v_w)
    ref_e = ops.transpose_operand(e_w[:, :I], split=split)
    ref_v = ops.transpose_operand(v_w[:, :I], split=split)
    ld = ops.operand_ld(O)
    e_t, v_t = torch.empty(I, ld, device=dev), torch.empty(I, ld, device=dev)
    _lib.check(_lib.lib().lbbnn_weight_operands_t(mu.data_ptr(), rho.data_ptr(), lam.data_ptr(), z.data_ptr() if mnf else None,
                                                  e_t.data_ptr(), v_t.data_ptr(), ld, O, I, ops.F_SPLIT16 if split else 0,
                                                  torch.cuda.current_stream(dev).cuda_stream), "lbbnn_weight_operands_t")
    torch.cuda.synchronize()
    assert torch.equal(e_t.view(torch.int32), ref_e.view(torch.int32))
    assert torch.equal(v_t.view(torch.int32), ref_v.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("dims,B", [((784, 1200, 1200, 10), 4096), ((40, 48, 24, 10), 64), ((64, 16, 16, 10), 32)])
def test_gemm_hosted_finalize_equals_standalone_finalize(bnn, dev, dims, B):
    """lbbnn_lrt_gemm_finalize: the KL tail computed by the extra workgroup of a GEMM launch (big-tile kernel, 2-wave
    small-tile kernel) or, where the kernel cannot host it (O <= 16: skinny kernel), by the fall-back launch, against the
    stand-alone lbbnn_layers_finalize on the same inputs: per-layer KLs and total (same arithmetic, 1e-6 relative), and
    the RNG offset advanced exactly once per forward."""
    from bnn_amd import _lib, ops
    bnn.manual_seed(31, 4)
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    x = torch.rand(B, dims[0], device=dev)
    with torch.no_grad():
        out = net(x, sample=True)
        kls = [float(l.kl) for l in (net.l1, net.l2, net.l3)]
        total = float(net.kl())
        st = ops.RngState.get(dev)
        assert int(st.t[1]) == 5 and int(st.t[3]) == 4          # live offset advanced once; the snapshot holds the offset used
        # stand-alone finalize on the buffers the forward left behind (K1 outputs, scal), reading the snapshot
        layers = [net.l1, net.l2, net.l3]
        descs = (_lib.LayerDesc * 3)()
        ref = torch.empty(4, device=dev)
        keep = [l._fill_desc(descs[i], (True, True, i < 2), ref[i]) for i, l in enumerate(layers)]
        _lib.check(_lib.lib().lbbnn_layers_finalize(descs, 3, st.t[2:4].data_ptr(), 0, ref[3:].data_ptr(),
                                                    torch.cuda.current_stream(dev).cuda_stream), "lbbnn_layers_finalize")
        torch.cuda.synchronize()
        del keep
    for a, b in zip(kls + [total], ref.tolist()):
        assert abs(a - b) <= 1e-6 * abs(b), (kls, total, ref.tolist())
    assert torch.isfinite(out).all()


@pytest.mark.gpu
def test_dense_masks_drawn_in_kernel(bnn, dev):
    """Without explicit masks the first launch of the dense-flow kernels draws the Bernoulli(0.5) masks from the layer's
    Philox state (lbbnn_dense_layer_t::draw_masks): values in {0,1}, about half ones, different per transform / call,
    reproducible from (seed, offset), and the forward that used them agrees with the oracle fed the same masks."""
    from bnn_amd import ops
    I, O, T, B = 1200, 64, 2, 16
    torch.manual_seed(2)
    layer = bnn.mnf.BayesianLinear(I, O, T).to(dev).train()          # RNVP by default
    x = torch.rand(B, I, device=dev)
    got = []
    for rep in range(2):
        bnn.manual_seed(99, 7)
        with torch.no_grad():
            out = layer(x, sample=True)
        m = layer._last_masks
        assert m["_in_kernel"]
        rows = torch.stack([*m["zmask"], *m["zmask2"], *m["rmask"]]).clone()
        got.append((rows, out.clone(), float(layer.kl)))
    rows = got[0][0]
    assert torch.equal(rows, got[1][0]) and torch.equal(got[0][1], got[1][1]) and got[0][2] == got[1][2]
    assert bool(((rows == 0) | (rows == 1)).all())
    means = rows.mean(1)
    assert float((means - 0.5).abs().max()) < 0.06, means            # 1200 draws each: sigma = 0.014
    for a in range(rows.shape[0]):
        for b in range(a + 1, rows.shape[0]):
            assert float((rows[a] != rows[b]).float().mean()) > 0.4   # independent masks
    # oracle on the same draws: eps from the Philox state the forward used, masks as drawn
    st = ops.RngState.get(dev)
    r = st.t.clone(); r[1] = 7
    L = layer._layer_id
    noise = {"eps_z": ops.philox_normal(r, ops.STREAM_EPS_Z * 64 + L, 0, I).cpu().reshape(1, I),
             "eps_z2": ops.philox_normal(r, ops.STREAM_EPS_Z2 * 64 + L, 0, I).cpu().reshape(1, I),
             "eps_act": ops.philox_normal(r, ops.STREAM_EPS_ACT * 64 + L, 0, O).cpu(),
             "eps_out": ops.philox_normal(r, ops.STREAM_EPS_OUT * 64 + L, B, O, 0).cpu(),
             "zmask": [rows[t].cpu() for t in range(T)], "zmask2": [rows[T + t].cpu() for t in range(T)],
             "rmask": [rows[2 * T + t].cpu() for t in range(T)]}
    p = {k: v.detach().cpu() for k, v in layer.state_dict().items()}
    zf = orc.flow_from_state("z_flow", "RNVP", p, T)
    rf = orc.flow_from_state("r_flow", "RNVP", p, T)
    o, kl, _ = orc.mnf_forward(x.cpu(), p, zf, rf, noise)
    assert rel_err(got[0][1].cpu(), o) < TOL
    assert abs(got[0][2] - float(kl)) / abs(float(kl)) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N,split", [(256, 1200, 784, True), (130, 96, 200, False), (64, 40, 33, False), (4096, 1200, 1200, True)])
def test_gemm_combine_epilogue(bnn, dev, M, K, N, split):
    """lbbnn_lrt_gemm_combine: add + 2 x (.) (a . w) against fp64, and against the two-step form it replaces
    (mean-only GEMM, then lbbnn_dx_combine) -- same products, one rounding order apart."""
    from bnn_amd import ops
    g = torch.Generator().manual_seed(12)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (0.1 * torch.randn(K, N, generator=g)).to(dev)
    x = torch.rand(M, N, generator=g).to(dev)
    add = torch.randn(M, N, generator=g).to(dev)
    ref = add.double() + 2 * x.double() * (a.double() @ w.double())
    split = split and ops.split_eligible(K, N)
    op = ops.transpose_operand(w, split=split)
    two_step = ops.dx_combine(add.clone(), ops.lrt_gemm(a, op, None, I=K, O=N, mean_only=True, split=split), x)
    fused = ops.lrt_gemm_combine(a, op, K=K, N=N, comb_x=x, comb_add=add.clone(), split=split)
    tol = 2e-5 if split else 2e-6
    assert rel_err(fused.double().cpu(), ref.cpu()) < tol
    assert rel_err(fused.cpu(), two_step.cpu()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["lrt", "Planar"])
def test_network_training_path_equals_per_layer_path(bnn, dev, family):
    """The training forward of a network (K3 / K1 of all layers in one launch per kind, every KL tail in the first GEMM's
    launch, one shared RNG offset, log_softmax in the head's GEMM epilogue, the KL total from the device-side finalize)
    against driving the same layers one by one (the head with the same fused log_softmax, the KL as the three-term torch
    sum): loss and every gradient bitwise equal."""
    for prec in ("fp32", "bf16x3"):
        bnn.set_precision(prec)
        torch.manual_seed(0)
        dims = (40, 64, 32, 10)
        net = (bnn.lrt.BayesianNetwork(dims) if family == "lrt" else
               bnn.mnf.BayesianNetwork(dims, 2, z_flow_type=family, r_flow_type=family)).to(dev).train()
        x = torch.rand(48, 40, device=dev)
        y = torch.randint(0, 10, (48,), device=dev)
        res = []
        for mode in ("net", "layers"):
            bnn.manual_seed(5, 3)
            net.zero_grad(set_to_none=True)
            if mode == "net":
                out, kl = net(x, sample=True), None
                kl = net.kl()
            else:
                h = x
                for i, l in enumerate((net.l1, net.l2, net.l3)):
                    l._advance_rng = False
                    l._lsm_now = i == 2                   # the head returns log-probabilities from its GEMM epilogue
                    h = l.forward(h, True, _relu=(i < 2))
                    l._advance_rng = True
                    l._lsm_now = False
                out = h
                kl = net.l1.kl + net.l2.kl + net.l3.kl
            loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + kl / 10
            loss.backward()
            res.append((float(loss.detach()), {n: p.grad.clone() for n, p in net.named_parameters()}))
        assert res[0][0] == res[1][0], (prec, res[0][0], res[1][0])
        for n in res[0][1]:
            assert torch.equal(res[0][1][n], res[1][1][n]), (prec, n)
    bnn.set_precision("fp32")


@pytest.mark.gpu
def test_dense_layer_random_shape_sweep():
    """tools/layer_fuzz.py with FLOW=any: 18 random layers across planar / RNVP / MNF-type flows (unaligned B / I / O down to
    I = 5, 1-3 transforms, both precisions): output, KL and every gradient of the HIP backward (incl.
    lbbnn_mnf_flow_dense_backward) against fp64 autograd of the oracle."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FLOW="any")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "layer_fuzz.py"), "7", "18"], capture_output=True, text=True,
                       timeout=600, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "random layers ok" in r.stdout


@pytest.mark.gpu
def test_gemm_random_shape_sweep():
    """tools/gemm_fuzz.py: 40 random (B, I, O) shapes through K1 + the dual-moment GEMM in both precisions against fp64
    (tails in every dimension, clamped rows, split eligibility boundaries)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_fuzz.py"), "3", "40"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "random shapes ok" in r.stdout


@pytest.mark.gpu
def test_layer_random_shape_sweep():
    """tools/layer_fuzz.py: 16 random planar MNF layers (unaligned B / I / O, 1-3 flow steps, both precisions, ReLU on/off):
    output, KL and every gradient of the all-HIP backward against fp64 autograd of the oracle, tolerance 2e-4 relative."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "layer_fuzz.py"), "5", "16"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "random layers ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("medimean", [True, False])
def test_base_mean_branches_with_log_probs_under_autograd(bnn, dev, golden, medimean):
    """LBBNN-GP-MF.py:236-251 under autograd (VERDICT r02 missing #2): the medimean / joint-mean branch with
    calculate_log_probs=True gives the values of the HIP no-grad pass on the same tau draws, and gradients equal to fp64
    autograd of the reference formulas on the CPU."""
    c = golden("base.npz").case("c1")
    B, I, O = [int(v) for v in c["shape"]]
    layer = _load_layer(bnn.base.BayesianLinear(I, O, 1), sub(c, "p."), dev).eval()
    x, cg = c["x"].to(dev), c["cgamma"].to(dev)
    layer.alpha = c["alpha_attr"].to(dev)
    layer.gamma.alpha = (1 / (1 + torch.exp(-layer.lambdal))).detach()
    layer.noise = {"tau_w": c["tau_w"].to(dev), "tau_b": c["tau_b"].to(dev)}
    with torch.no_grad():
        out0 = layer(x, cg, sample=False, medimean=medimean, calculate_log_probs=True)
        lp0, lq0 = layer.log_prior.clone(), layer.log_variational_posterior.clone()
    out = layer(x, cg, sample=False, medimean=medimean, calculate_log_probs=True)
    lp, lq = layer.log_prior, layer.log_variational_posterior
    assert out.requires_grad and lp.requires_grad and lq.requires_grad
    assert rel_err(out, out0) < 1e-5 and rel_err(lp, lp0) < 2e-5 and rel_err(lq, lq0) < 2e-5
    (out.sum() + 0.01 * (lq - lp)).backward()
    # fp64 autograd of the same formulas (the oracle's densities)
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in layer.named_parameters()}
    x64, cg64 = x.cpu().double(), cg.cpu().double()
    w = (cg64 if medimean else c["alpha_attr"].double()) * P["weight_mu"]
    b = P["bias_mu"]
    lp64 = (orc.gaussgamma_log_prob(w, cg64, P["weight_a"], P["weight_b"], c["tau_w"].double(), False)
            + orc.gaussgamma_log_prob(b, torch.ones_like(b), P["bias_a"], P["bias_b"], c["tau_b"].double(), False)
            + orc.betabinomial_log_prob(cg64, P["pa"], P["pb"], False))
    ga = layer.gamma.alpha.detach().cpu().double()
    lq64 = (orc.gaussian_full_log_prob(w, cg64, P["weight_mu"], P["weight_rho"]) + orc.bernoulli_log_prob(cg64, ga, False)
            + orc.gaussian_log_prob(b, P["bias_mu"], P["bias_rho"]))
    out64 = torch.nn.functional.linear(x64, w, b)
    (out64.sum() + 0.01 * (lq64 - lp64)).backward()
    for k, p in layer.named_parameters():
        if P[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        assert rel_err(p.grad, P[k].grad) < 1e-4, (k, rel_err(p.grad, P[k].grad))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 13, 26, 34])
def test_planar_backward_tiny_shape_of_the_round2_fuzz_outlier(bnn, dev, seed):
    """VERDICT r02 weak #1b: round 2's fuzz campaign met ONE fp32-mode gradient (r0_c, a (B, I, O, T) = (3, 5, 7, 3) planar
    layer) 1.21e-4 off.  tools/fuzz_r0c.py repeats that shape over 40 seeds (profiles/r03_fuzz_r0c_shape.txt): r0_c stays under
    4.4e-6, the worst gradient of any parameter is 4.4e-5 (r0_b2) -- five-element gradients whose entries are ~1e-5, sums over
    seven output features that nearly cancel, so the ~1e-6 of the hardware exp / log forms in K1b / V1 is amplified by the
    conditioning of the sum, not by an arithmetic mistake.  Here: the four worst seeds of that sweep, r0_c <= 5e-5, every
    gradient inside the 1e-4 contract."""
    B, I, O, T = 3, 5, 7, 3
    torch.manual_seed(seed)
    layer = bnn.mnf.BayesianLinear(I, O, T, z_flow_type="Planar", r_flow_type="Planar")
    with torch.no_grad():
        for fl in (layer.z_flow, layer.r_flow):
            for tr in fl.transforms:
                tr.u.mul_(6.0); tr.w.mul_(6.0); tr.bias.mul_(6.0)
        layer.q0_mean.add_(1.0); layer.weight_mu.mul_(10)
    g = torch.Generator().manual_seed(1000 + seed)
    noise = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(B, O, generator=g),
             "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(O, generator=g)}
    x = torch.rand(B, I, generator=g)
    wgt = torch.randn(B, O, generator=g)
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train()
    layer.noise = {k: v.to(dev) for k, v in noise.items()}
    out = layer(x.to(dev), sample=True)
    ((out * wgt.to(dev)).sum() + layer.kl / 60).backward()
    pc = {k: v.double().requires_grad_(True) for k, v in p.items()}
    zf = orc.flow_from_state("z_flow", "Planar", pc, T); rf = orc.flow_from_state("r_flow", "Planar", pc, T)
    o, kl, _ = orc.mnf_forward(x.double(), pc, zf, rf, {k: v.double() for k, v in noise.items()})
    ((o * wgt.double()).sum() + kl / 60).backward()
    for name, prm in layer.named_parameters():
        ref = pc[name].grad
        if ref is not None and float(ref.abs().max()) > 0:
            e = rel_err(prm.grad, ref)
            assert e < (5e-5 if name == "r0_c" else 1e-4), (name, e)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C", [(4096, 10), (33, 7), (1, 16)])
def test_fused_elbo_loss_and_log_softmax_backward_vs_torch(bnn, dev, B, C):
    """bnn_amd.elbo_loss == F.nll_loss(reduction='sum') + kl / num_batches (train(), LBBNN-GP-MF-MNF.py:268-271) in value and
    in both gradients; ops.log_softmax_backward == autograd of F.log_softmax."""
    g = torch.Generator().manual_seed(B + C)
    logits = torch.randn(B, C, generator=g).to(dev).requires_grad_(True)
    y = torch.randint(0, C, (B,), generator=g).to(dev)
    kl = torch.tensor(1234.5, device=dev, requires_grad=True)
    lp = torch.log_softmax(logits, dim=1)
    loss = bnn.elbo_loss(lp, y, kl, 600)
    loss.backward()
    g_fused, gk_fused = logits.grad.clone(), kl.grad.clone()
    logits.grad = None; kl.grad = None
    ref = torch.nn.functional.nll_loss(torch.log_softmax(logits.double(), dim=1), y, reduction="sum") + kl.double() / 600
    ref.backward()
    assert rel_err(loss, ref) < 1e-6
    assert rel_err(g_fused, logits.grad) < 1e-6 and rel_err(gk_fused, kl.grad) < 1e-6
    assert rel_err(bnn.elbo_loss(lp.detach(), y), torch.nn.functional.nll_loss(lp.detach().double(), y, reduction="sum")) < 1e-6
    gout = torch.randn(B, C, generator=g).to(dev)
    lpd = lp.detach()
    l2 = logits.detach().double().requires_grad_(True)
    (torch.log_softmax(l2, dim=1) * gout.double()).sum().backward()
    assert rel_err(bnn.ops.log_softmax_backward(gout, lpd), l2.grad) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["fp32", "fp16x3f"])
def test_training_forward_fuses_log_softmax_and_sums_kl_on_device(bnn, dev, prec):
    """The network's training forward returns log-probabilities straight from the head's GEMM epilogue (no torch softmax
    kernels) and net.kl() is the device-side total of the KL finalize: values and gradients equal those of the same network
    run layer by layer with torch's log_softmax and the three-term KL sum (same explicit draws)."""
    dims, B = (784, 96, 64, 10), 48
    torch.manual_seed(4)
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    net.set_precision(prec)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 784, generator=g).to(dev)
    y = torch.randint(0, 10, (B,), generator=g).to(dev)
    for l in net._layers():
        l.noise = {"eps_z": torch.randn(1, l.in_features, generator=g).to(dev), "eps_out": torch.randn(B, l.out_features, generator=g).to(dev),
                   "eps_z2": torch.randn(1, l.in_features, generator=g).to(dev), "eps_act": torch.randn(l.out_features, generator=g).to(dev)}
    out = net(x, sample=True)
    kl = net.kl()
    assert type(kl.grad_fn).__name__.startswith("_SumKLFn") and float((out.exp().sum(1) - 1).abs().max()) < 1e-5
    loss = bnn.elbo_loss(out, y, kl, 100)
    loss.backward()
    got = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad(set_to_none=True)
    h = x
    for i, l in enumerate(net._layers()):
        h = l(h, sample=True, _relu=(i < 2))
    out2 = torch.log_softmax(h, dim=1)
    kl2 = net.l1.kl + net.l2.kl + net.l3.kl
    loss2 = torch.nn.functional.nll_loss(out2, y, reduction="sum") + kl2 / 100
    loss2.backward()
    assert rel_err(out, out2) < 1e-6 and rel_err(kl, kl2) < 1e-6 and rel_err(loss, loss2) < 1e-6
    for k, p in net.named_parameters():
        assert rel_err(got[k], p.grad) < 2e-5, (k, rel_err(got[k], p.grad))


@pytest.mark.parametrize("B,C,I,S,stoch", [(4096, 10, 1200, 16, True), (130, 3, 70, 4, True), (64, 16, 33, 7, False), (257, 1, 64, 3, True)])
def test_head_dw_slabs_vs_fp64(bnn, dev, B, C, I, S, stoch):
    """lbbnn_head_dw: the split-K slabs of dW_m = G_m^T x and dW_v = G_v^T x^2 (autograd of the two torch.mm of
    LBBNN-GP-MF-LRT.py:172-173 for a <= 16-class head) against fp64 -- every slab on its own row range (ragged last slab,
    columns past a multiple of 64, non-dense row strides), 2e-6 of the slab's max; bitwise reproducible."""
    ops = bnn.ops
    g = torch.Generator().manual_seed(B + C + I)
    gm_full = torch.randn(B, C + 3, generator=g).to(dev)
    gv_full = torch.randn(B, C + 3, generator=g).to(dev)
    x_full = torch.randn(B, I + 5, generator=g).to(dev)
    gm, gv, x = gm_full[:, :C], gv_full[:, :C], x_full[:, :I]
    dWm, dWv = ops.head_dw(gm, gv if stoch else None, x, nslabs=S)
    assert dWm.shape == (S, C, I) and (dWv is None) == (not stoch)
    r = (B + S - 1) // S
    for s in range(S):
        lo, hi = s * r, min((s + 1) * r, B)
        ref_m = gm[lo:hi].double().T @ x[lo:hi].double()
        assert rel_err(dWm[s], ref_m) < 2e-6
        if stoch:
            assert rel_err(dWv[s], gv[lo:hi].double().T @ (x[lo:hi].double() ** 2)) < 2e-6
    again = ops.head_dw(gm, gv if stoch else None, x, nslabs=S)
    assert torch.equal(again[0], dWm) and (not stoch or torch.equal(again[1], dWv))


def test_head_weight_gradients_same_through_head_dw_and_the_gemm_route(bnn, dev, monkeypatch):
    """A 784-48-32-10 planar MNF net, one ELBO backward: the head's weight gradients through lbbnn_head_dw against the
    route it replaces (x^T | (x^2)^T operands + two split-K GEMM launches): every parameter gradient of the net within 2e-5
    (both are fp32-accumulate sums of the same products in different orders; bf16x3 operands on the GEMM route)."""
    from bnn_amd import layers
    grads = {}
    for mode in (True, False):
        monkeypatch.setattr(layers, "_HEAD_DW", mode)
        bnn.manual_seed(5, 0)
        torch.manual_seed(5)
        net = bnn.mnf.BayesianNetwork((784, 48, 32, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
        xg = torch.Generator().manual_seed(1)
        x = torch.rand(256, 784, generator=xg).to(dev)
        y = torch.randint(0, 10, (256,), generator=xg).to(dev)
        out = net(x, sample=True)
        loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + net.kl() / 10
        loss.backward()
        grads[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        del loss, out
    for n in grads[True]:
        assert rel_err(grads[True][n], grads[False][n]) < 2e-5, n


def test_deferred_column_sums_are_bitwise_the_stand_alone_ones(bnn, dev):
    """lbbnn_reduce_partials_batch: the second level of lbbnn_output_grad's and lbbnn_weight_pass_backward's column sums for
    several layers in one launch == the launches those entry points make themselves, bit for bit (one shared device body);
    ragged shapes, a posterior-mean (one-vector) job, NULL outputs, more than one launch group."""
    ops = bnn.ops
    g = torch.Generator().manual_seed(3)
    jobs, want = [], []
    for (B, O) in [(4096, 1200), (130, 70), (64, 10)]:
        gout = torch.randn(B, O, generator=g).to(dev)
        out = torch.randn(B, O, generator=g).to(dev)
        std = (0.1 + torch.rand(B, O, generator=g)).to(dev)
        eps = torch.randn(B, O, generator=g).to(dev)
        for stoch in (True, False):
            kw = dict(out=out, std=std if stoch else None, eps=eps if stoch else None, relu=True)
            ref = ops.output_grad(gout, **kw)
            got = ops.output_grad(gout, defer_sums=jobs, **kw)
            want.append((ref[4], got[4])); want.append((ref[5], got[5]))
    for (O, I) in [(1200, 784), (10, 1200), (33, 68)]:
        mu = (0.1 * torch.randn(O, I, generator=g)).to(dev)
        rho = (-4.5 + 0.3 * torch.randn(O, I, generator=g)).to(dev)
        lam = torch.randn(O, I, generator=g).to(dev)
        dWm, dWv = torch.randn(O, I, generator=g).to(dev), torch.randn(O, I, generator=g).to(dev)
        zf, zk, rc = ((1 + 0.1 * torch.randn(I, generator=g)).to(dev) for _ in range(3))
        dam, dav = torch.randn(O, generator=g).to(dev), torch.randn(O, generator=g).to(dev)
        gk = torch.tensor(0.7, device=dev)
        kw = dict(z_fwd=zf, z_kl=zk, r0_c=rc, da_mu=dam, da_var=dav, g_kl=gk, priors=bnn.Priors())
        ref = ops.weight_pass_backward(mu, rho, lam, dWm, dWv, **kw)
        got = ops.weight_pass_backward(mu, rho, lam, dWm, dWv, defer_sums=jobs, **kw)
        for k in range(3):
            assert torch.equal(ref[k], got[k])
        want += [(ref[3], got[3]), (ref[4], got[4]), (ref[5], got[5])]
    assert len(jobs) == 9                                   # two launch groups (8 + 1)
    ops.reduce_partials_flush(jobs)
    assert not jobs
    torch.cuda.synchronize()
    for r, t in want:
        assert (r is None) == (t is None)
        if r is not None:
            assert torch.equal(r, t)


def test_training_step_same_with_and_without_deferred_column_sums(bnn, dev, monkeypatch):
    """One ELBO backward of a 784-48-32-10 planar MNF net inside vector_backward_overlap: every parameter gradient bitwise
    the same whether the layers' column sums are finished by their own launches or by the one batched launch at the end."""
    from bnn_amd import layers
    grads = {}
    for mode in (True, False):
        monkeypatch.setattr(layers, "_DEFER_SUMS", mode)
        bnn.manual_seed(5, 0)
        torch.manual_seed(5)
        net = bnn.mnf.BayesianNetwork((784, 48, 32, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
        xg = torch.Generator().manual_seed(1)
        x = torch.rand(256, 784, generator=xg).to(dev)
        y = torch.randint(0, 10, (256,), generator=xg).to(dev)
        out = net(x, sample=True)
        loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + net.kl() / 10
        with layers.vector_backward_overlap():
            loss.backward()
        grads[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        del loss, out
    for n in grads[True]:
        assert torch.equal(grads[True][n], grads[False][n]), n


@pytest.mark.parametrize("kind", ["Planar", "RNVP"])
def test_batched_v1_from_the_kl_sum_backward_is_bitwise_the_per_layer_one(bnn, dev, monkeypatch, kind):
    """lbbnn_mnf_aux_backward_batch: all layers' V1 launched once from the backward of the network's KL sum (losses._SumKLFn)
    against one lbbnn_mnf_aux_backward per layer -- same device body, so every parameter gradient of an ELBO backward is
    bitwise the same; and a loss that does not go through net.kl() (per-layer KLs added by hand) still gets its V1."""
    from bnn_amd import layers
    grads = {}
    for mode in ("batch", "per_layer", "manual_kl"):
        monkeypatch.setattr(layers, "_V1_BATCH", mode != "per_layer")
        bnn.manual_seed(5, 0)
        torch.manual_seed(5)
        net = bnn.mnf.BayesianNetwork((784, 48, 32, 10), 2, z_flow_type=kind, r_flow_type=kind).to(dev).train()
        xg = torch.Generator().manual_seed(1)
        x = torch.rand(256, 784, generator=xg).to(dev)
        y = torch.randint(0, 10, (256,), generator=xg).to(dev)
        out = net(x, sample=True)
        kl = net.kl() if mode != "manual_kl" else (net.l1.kl + net.l2.kl + net.l3.kl)
        loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + kl / 10
        loss.backward()
        grads[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        del loss, out, kl
    for n in grads["batch"]:
        assert torch.equal(grads["batch"][n], grads["per_layer"][n]), n
        assert rel_err(grads["manual_kl"][n], grads["per_layer"][n]) < 1e-6, n


def test_loss_backward_hands_the_logits_gradient_to_the_head(bnn, dev, monkeypatch):
    """lbbnn_elbo_loss_backward_logits: with bnn_amd.elbo_loss on a training forward whose head has its log_softmax fused, the
    loss's backward launch also forms the gradient with respect to the logits and the head's backward takes it instead of
    launching lbbnn_log_softmax_backward: every parameter gradient bitwise the same as with the two launches; the reference's
    own spelling of the loss (F.nll_loss + kl / N) keeps taking the separate launch."""
    from bnn_amd import losses, ops
    grads, calls = {}, {True: 0, False: 0, "torch": 0}
    real = ops.log_softmax_backward
    for mode in (True, False, "torch"):
        monkeypatch.setattr(losses, "_FUSE_LSM_BWD", mode is True)

        def counted(*a, _m=mode, **k):
            calls[_m] += 1
            return real(*a, **k)
        monkeypatch.setattr(ops, "log_softmax_backward", counted)
        bnn.manual_seed(5, 0)
        torch.manual_seed(5)
        net = bnn.mnf.BayesianNetwork((784, 48, 32, 10), 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
        xg = torch.Generator().manual_seed(1)
        x = torch.rand(256, 784, generator=xg).to(dev)
        y = torch.randint(0, 10, (256,), generator=xg).to(dev)
        out = net(x, sample=True)
        if mode == "torch":
            loss = torch.nn.functional.nll_loss(out, y, reduction="sum") + net.kl() / 10
        else:
            loss = bnn.elbo_loss(out, y, net.kl(), 10)
        loss.backward()
        grads[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        del loss, out
    assert calls == {True: 0, False: 1, "torch": 1}, calls                # the hand-over happened exactly where it should
    for n in grads[True]:
        assert torch.equal(grads[True][n], grads[False][n]), n
        assert rel_err(grads["torch"][n], grads[False][n]) < 1e-5, n


@pytest.mark.gpu
@pytest.mark.parametrize("stochastic_kl", [(True, True), (True, False), (False, False)])
def test_lrt_bias_gradients_from_the_hip_kernel(bnn, dev, stochastic_kl, monkeypatch):
    """lbbnn_bias_backward: the LRT layer's bias_mu / bias_rho gradients (activation mean + bias_mu, activation variance +
    softplus(bias_rho)^2, KL bias term: LBBNN-GP-MF-LRT.py:171-173, 189-196) in one launch instead of a torch autograd graph over
    the two vectors -- against fp64 autograd of the oracle, and against the torch-graph route kept behind LBBNN_LRT_BIAS_HIP=0;
    training-mode forward with / without the KL in the loss, and the posterior-mean forward of eval mode (no variance term, no KL:
    bias_rho gets no gradient)."""
    from bnn_amd import layers
    sample, use_kl = stochastic_kl
    I, O, B = 72, 130, 33
    torch.manual_seed(21)
    layer = bnn.lrt.BayesianLinear(I, O)
    with torch.no_grad():
        layer.bias_rho.add_(6.0); layer.bias_mu.add_(0.3)          # sigma_b ~ 0.05 instead of e^-9: both KL terms matter
    p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).train(sample)
    g = torch.Generator().manual_seed(22)
    x, eps, w = torch.rand(B, I, generator=g), torch.randn(B, O, generator=g), torch.randn(B, O, generator=g)
    grads = {}
    for hip in (True, False):
        monkeypatch.setattr(layers, "_LRT_BIAS_HIP", hip)
        layer.zero_grad()
        layer.noise = {"eps_out": eps.to(dev)}
        out = layer(x.to(dev), sample=sample)
        ((out * w.to(dev)).sum() + (layer.kl / 7 if use_kl else 0)).backward()
        grads[hip] = {n: (q.grad.clone() if q.grad is not None else torch.zeros_like(q)) for n, q in layer.named_parameters()}
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
    o, kl, _ = orc.lrt_forward(x.double(), p64, eps.double() if sample else None, stochastic=sample, compute_kl=sample)
    ((o * w.double()).sum() + (kl / 7 if use_kl else 0)).backward()
    for n in ("bias_mu", "bias_rho"):
        ref = p64[n].grad
        if ref is None or float(ref.abs().max()) == 0:
            assert float(grads[True][n].abs().max()) == 0, n
            continue
        assert rel_err(grads[True][n], ref) < 2e-6, (n, rel_err(grads[True][n], ref))
        assert rel_err(grads[True][n], grads[False][n].cpu().double()) < 2e-6, n
    for n in ("weight_mu", "weight_rho", "lambdal"):
        assert torch.equal(grads[True][n], grads[False][n]), n
    # the one-launch form (second level of the column sums + bias gradients: lbbnn_bias_backward_partials) is bitwise the
    # stand-alone sums followed by lbbnn_bias_backward
    from bnn_amd import ops
    gm = torch.randn(B, O, device=dev)
    std = torch.rand(B, O, device=dev) + 0.1
    job = []
    ops.output_grad(gm, std=std if sample else None, eps=eps.to(dev) if sample else None, defer_sums=job)
    *_, g_sum, gv_sum = ops.output_grad(gm, std=std if sample else None, eps=eps.to(dev) if sample else None)
    gk = torch.full((), 0.37, device=dev) if use_kl else None
    a = ops.bias_backward_partials(job[0], B, layer.bias_mu, layer.bias_rho, gk, layer.priors)
    b = ops.bias_backward(layer.bias_mu, layer.bias_rho, g_sum, gv_sum, gk, layer.priors)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
