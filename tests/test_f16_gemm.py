"""GPU tier: the row-scaled FP16 operand format (LBBNN_F_F16S, include/lbbnn.h; round 3) through the C ABI against fp64.

Bars, stated up front (tools/format_error.py gives the expected values):
  fp16x3  (3 + 3 products)  max|err| <= 2e-6 max|out|  AND element-wise |err| <= 1e-6 max|out| + 1e-4 |ref|: the bar the exact
          fp32-MFMA path is held to (the format itself is 2-3e-8; what is left is the fp32 accumulate and the hardware
          exp / log forms of the weight pass, the same for every precision)
  fp16x3f (3 + 1 products)  max|err| <= 4e-5 max|out|  AND element-wise |err| <= 3e-5 max|out| + 1e-4 |ref|  (measured 1.4-1.8e-5
          at the headline contraction lengths K >= 784; the single variance product's rounding averages out as 1 / sqrt(K), so
          short contractions sit higher: 2.1e-5 at K = 96)
Contract of BASELINE.json: 1e-4 relative.  Reference arithmetic: LBBNN-GP-MF-LRT.py:170-175, LBBNN-GP-MF-MNF.py:195-200."""
import pytest
import torch

from conftest import elementwise_violation, rel_err
from oracle import lbbnn_oracle as orc

pytestmark = pytest.mark.gpu

BARS = {"fp16x3": (2e-6, 1e-6), "fp16x3f": (4e-5, 3e-5)}      # (max-norm bar, atol fraction of the element-wise bar)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def bnn():
    import bnn_amd
    return bnn_amd


def _decode_units(buf, rows, I):
    """fp16 hi | lo units (128-B lines: unit 2g = hi of k in [8g, 8g+8), unit 2g+1 = lo) -> (hi, lo) as (rows, ld) fp64."""
    ld = buf.shape[1]
    h = buf.cpu().contiguous().view(torch.float16).view(rows, ld // 32, 4, 2, 8).double()     # [row][chunk][group][hi/lo][8]
    hi = h[:, :, :, 0, :].reshape(rows, ld)
    lo = h[:, :, :, 1, :].reshape(rows, ld)
    return hi, lo


def _decode_rows16(buf, rows):
    """plain fp16 rows of ld halves (the hi-only var_w of the 3 + 1 form), packed at the start of an fp32-typed (rows, ld)
    buffer -> (rows, ld) fp64; the buffer's second half is unused."""
    ld = buf.shape[1]
    return buf.cpu().contiguous().view(torch.float16).reshape(-1)[:rows * ld].view(rows, ld).double()


def _operands(bnn, dev, I, O, g, mnf=True, fmt=2):
    ops = bnn.ops
    p = orc.init_mnf_params(I, O, g)
    z = 1 + 0.1 * torch.randn(I, generator=g)
    d = {k: v.to(dev) for k, v in p.items()}
    ld = ops.operand_ld(I)
    ws = {"e_w": torch.empty(O, ld, device=dev), "var_w": torch.empty(O, ld, device=dev), "bias_var": torch.empty(O, device=dev),
          "e_scale": torch.empty(O, device=dev), "v_scale": torch.empty(O, device=dev)}
    ops.weight_pass(d["weight_mu"], d["weight_rho"], d["lambdal"], z_fwd=z.to(dev), bias_rho=d["bias_rho"],
                    priors=bnn.Priors(), e_w=ws["e_w"], var_w=ws["var_w"], bias_var=ws["bias_var"], split=fmt,
                    e_scale=ws["e_scale"], v_scale=ws["v_scale"])
    alpha = orc.alpha_of(p["lambdal"].double()); sigma = orc.sigma_of(p["weight_rho"].double())
    ew = p["weight_mu"].double() * alpha * z.double()
    vw = sigma ** 2 * alpha ** 2
    return p, d, ws, ew, vw


@pytest.mark.parametrize("O,I", [(80, 64), (400, 784), (1200, 1200), (33, 1272), (17, 40)])
@pytest.mark.parametrize("fmt", [2, 3])
def test_weight_pass_f16_operands_and_scales(bnn, dev, O, I, fmt):
    """K1 in the row-scaled fp16 format: (hi + lo) * scale reproduces the fp64 operands to 2e-6 relative per ROW maximum,
    the scales are exact powers of two that put each row's maximum into [2^13, 2^14), the k tail is zero."""
    g = torch.Generator().manual_seed(O * 7 + I)
    p, d, ws, ew, vw = _operands(bnn, dev, I, O, g, fmt=fmt)
    if fmt == 3:
        # var_w of the 3 + 1 form: one RNE fp16 per weight, plain rows of ld halves
        vh = _decode_rows16(ws["var_w"], O)
        s = ws["v_scale"].cpu().double()
        assert torch.equal(torch.frexp(s)[0], torch.full_like(s, 0.5))
        assert float(((vh[:, :I] * s[:, None] - vw).abs() / vw.abs().amax(dim=1, keepdim=True)).max()) < 2.0 ** -11
        assert float(vh[:, I:].abs().max() if vh.shape[1] > I else 0.0) == 0.0
        assert bool(((vh.amax(dim=1) >= 2.0 ** 13 * 0.999) & (vh.amax(dim=1) <= 2.0 ** 14)).all())
    for name, ref, sc in (("e_w", ew, ws["e_scale"]),) + ((("var_w", vw, ws["v_scale"]),) if fmt == 2 else ()):
        hi, lo = _decode_units(ws[name], O, I)
        s = sc.cpu().double()
        assert torch.equal(torch.frexp(s)[0], torch.full_like(s, 0.5)), name          # exact powers of two
        got = (hi + lo)[:, :I] * s[:, None]
        rowmax = ref.abs().amax(dim=1, keepdim=True)
        assert float(((got - ref).abs() / rowmax).max()) < 2e-6, name
        scaled_max = (hi.abs().amax(dim=1))
        assert bool(((scaled_max >= 2.0 ** 13 * 0.999) & (scaled_max <= 2.0 ** 14)).all()), (name, scaled_max.min(), scaled_max.max())
        assert float(hi[:, I:].abs().max() if hi.shape[1] > I else 0.0) == 0.0 and float(lo[:, I:].abs().max() if lo.shape[1] > I else 0.0) == 0.0
        # lo is the residual of hi: |lo| <= half an ulp of hi
        assert float((lo.abs() / hi.abs().clamp_min(2.0 ** -14)).max()) <= 2.0 ** -10


@pytest.mark.parametrize("B,I", [(5, 64), (128, 784), (33, 1200), (4096, 784)])
def test_format_x_planes(bnn, dev, B, I):
    g = torch.Generator().manual_seed(B + I)
    x = (torch.rand(B, I, generator=g) * torch.where(torch.rand(B, I, generator=g) < 0.3, 0.0, 1.0)).to(dev)
    pl = bnn.ops.format_x(x)
    hi, lo = _decode_units(pl, B, I)
    x64 = x.cpu().double()
    assert torch.equal(hi[:, :I], x64.to(torch.float32).to(torch.float16).double())
    assert float(((hi + lo)[:, :I] - x64).abs().max()) <= 2.0 ** -22
    if pl.shape[1] > I:
        assert float(hi[:, I:].abs().max()) == 0.0 and float(lo[:, I:].abs().max()) == 0.0


@pytest.mark.parametrize("B,I,O", [(128, 64, 80), (100, 784, 400), (257, 1200, 1200), (64, 40, 17), (4000, 784, 1200),
                                   (3333, 1200, 1192), (2100, 96, 1200), (1, 8, 24)])
@pytest.mark.parametrize("prec", ["fp16x3", "fp16x3f"])
@pytest.mark.parametrize("xsrc", ["f32", "planes"])
def test_gemm16_vs_fp64(bnn, dev, B, I, O, prec, xsrc):
    """The dual-moment GEMM + sampling epilogue in both forms of the format, x as fp32 rows (split in registers) and as
    planes, odd shapes and K tails included, against fp64."""
    ops = bnn.ops
    g = torch.Generator().manual_seed(B + I + O)
    x = torch.rand(B, I, generator=g)
    p, d, ws, ew, vw = _operands(bnn, dev, I, O, g, fmt=3 if prec == "fp16x3f" else 2)
    eps = torch.randn(B, O, generator=g)
    xin = x.to(dev)
    if xsrc == "planes":
        xin = ops.format_x(xin)
    out, _ = ops.lrt_gemm16(xin, ws["e_w"], ws["var_w"], ws["e_scale"], ws["v_scale"], I=I, O=O, bias_mean=d["bias_mu"],
                            bias_var=ws["bias_var"], eps=eps.to(dev), var1=(prec == "fp16x3f"), x_planes=(xsrc == "planes"))
    x64 = x.double()
    ref = x64 @ ew.T + p["bias_mu"].double() + torch.sqrt((x64 ** 2) @ vw.T + orc.sigma_of(p["bias_rho"].double()) ** 2) * eps.double()
    bar, atol = BARS[prec]
    e = rel_err(out, ref)
    assert e < bar, e
    v = elementwise_violation(out, ref, rtol=1e-4, atol_frac=atol)
    assert v <= 1.0, v


@pytest.mark.parametrize("B,I,O", [(128, 64, 80), (300, 784, 1200), (37, 96, 40)])
@pytest.mark.parametrize("prec", ["fp16x3", "fp16x3f"])
def test_gemm16_plane_output_feeds_the_next_layer_bitwise(bnn, dev, B, I, O, prec):
    """out_planes of a ReLU layer == lbbnn_format_x of its fp32 output, bit for bit (tail zero), with and without the fp32
    copy; in-kernel Philox noise == the same draws handed in explicitly; std_out is sqrt(var)."""
    ops = bnn.ops
    g = torch.Generator().manual_seed(B * 3 + I + O)
    x = torch.rand(B, I, generator=g).to(dev)
    p, d, ws, ew, vw = _operands(bnn, dev, I, O, g, fmt=3 if prec == "fp16x3f" else 2)
    rng = torch.tensor([1234, 7, 0, 0], dtype=torch.int64, device=dev)
    kw = dict(I=I, O=O, bias_mean=d["bias_mu"], bias_var=ws["bias_var"], rng=rng, rng_stream=5, row_offset=11, relu=True,
              var1=(prec == "fp16x3f"))
    args = (x, ws["e_w"], ws["var_w"], ws["e_scale"], ws["v_scale"])
    pl = torch.zeros(B, ops.plane_ld(O), device=dev)
    std = torch.empty(B, O, device=dev)
    out, _ = ops.lrt_gemm16(*args, out_planes=pl, std_out=std, **kw)
    assert torch.equal(pl, ops.format_x(out))
    pl2 = torch.zeros_like(pl)
    none, _ = ops.lrt_gemm16(*args, out_planes=pl2, want_out=False, **kw)
    assert none is None and torch.equal(pl2, pl)
    eps = ops.philox_normal(rng, 5, B, O, 11)
    kw2 = dict(kw); kw2.pop("rng"); kw2.pop("rng_stream"); kw2.pop("row_offset")
    out_e, _ = ops.lrt_gemm16(*args, eps=eps, **kw2)
    assert torch.equal(out_e, out)
    x64 = x.cpu().double()
    var = (x64 ** 2) @ vw.T + orc.sigma_of(p["bias_rho"].double()) ** 2
    # (sqrt(var) carries the weight pass's hardware exp / log forms, ~2e-6 on var_w in every precision; the single variance
    # product of the fast mode adds its 2^-12 roundings / sqrt(K))
    assert rel_err(std, var.sqrt()) < (1e-4 if prec == "fp16x3f" else 5e-6)
    # the planes are what the next layer reads: x given as planes == x given as the fp32 rows they were made from, when the
    # fp32 rows ARE representable (hi + lo exact): feed out (already rounded to hi + lo by construction? no -- so compare values)
    o2 = ops.lrt_gemm16(ops.format_x(x), *args[1:], x_planes=True, eps=eps, **kw2)[0]
    assert rel_err(o2, out) < (3e-5 if prec == "fp16x3f" else 2e-6)
    if prec == "fp16x3f":
        # 3 + 1 products: the fp32-row form splits x into the same hi | lo halves lbbnn_format_x writes and takes s from them
        # by the same four packed instructions -- the same bits, whichever way x arrives
        assert torch.equal(o2, out)


def test_range_overflow_is_loud(bnn, dev):
    """fp16 holds |x| < 4096 in this format (x^2 2^-8 must stay under 65504): a larger activation gives a NON-FINITE output
    row, never a silently saturated one; the other rows are untouched."""
    ops = bnn.ops
    B, I, O = 64, 64, 80
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, I, generator=g)
    x[7, 13] = 5000.0
    x[9, 3] = 1.0e6
    p, d, ws2, ew, vw = _operands(bnn, dev, I, O, torch.Generator().manual_seed(4), fmt=2)
    p, d, ws3, ew, vw = _operands(bnn, dev, I, O, torch.Generator().manual_seed(4), fmt=3)
    ws = ws2
    eps = torch.randn(B, O, generator=g).to(dev)
    for xin, planes in ((x.to(dev), False), (ops.format_x(x.to(dev)), True)):
        for var1 in (False, True):
            ws = ws3 if var1 else ws2
            out, _ = ops.lrt_gemm16(xin, ws["e_w"], ws["var_w"], ws["e_scale"], ws["v_scale"], I=I, O=O,
                                    bias_mean=d["bias_mu"], bias_var=ws["bias_var"], eps=eps, var1=var1, x_planes=planes)
            fin = torch.isfinite(out).all(dim=1).cpu()
            assert not bool(fin[7]) and not bool(fin[9]), (planes, var1)
            assert bool(fin[torch.arange(B)[(torch.arange(B) != 7) & (torch.arange(B) != 9)]].all())
    # ... and in range up to the documented limit the result is still accurate
    ws = ws2
    x2 = torch.rand(B, I, generator=g) * 4000.0
    out, _ = ops.lrt_gemm16(x2.to(dev), ws["e_w"], ws["var_w"], ws["e_scale"], ws["v_scale"], I=I, O=O,
                            bias_mean=d["bias_mu"], bias_var=ws["bias_var"], eps=eps)
    x64 = x2.double()
    ref = x64 @ ew.T + p["bias_mu"].double() + torch.sqrt((x64 ** 2) @ vw.T + orc.sigma_of(p["bias_rho"].double()) ** 2) * eps.cpu().double()
    assert rel_err(out, ref) < 2e-6


def _headline_case(bnn, seed=21):
    dims, B, T = (784, 1200, 1200, 10), 4096, 2
    torch.manual_seed(seed)
    net = bnn.mnf.BayesianNetwork(dims, T, z_flow_type="Planar", r_flow_type="Planar")
    layers = [net.l1, net.l2, net.l3]
    g = torch.Generator().manual_seed(seed + 1)
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
    return net, layers, x, noises, P, zf, rf


@pytest.mark.parametrize("prec", ["fp16x3", "fp16x3f"])
@pytest.mark.parametrize("first", ["planes", "f32"])
def test_f16_headline_network_vs_oracle(bnn, dev, prec, first, monkeypatch):
    """BASELINE configs[2] (784-1200-1200-10 MNF / planar, B = 4096) under net.set_precision(prec), the fused no-grad forward
    (activations handed on as planes) AND the per-layer autograd path, against the oracle run in fp64: outputs to the bars
    above, every layer's KL to 1e-4 (the KL never touches the 16-bit operands)."""
    from test_parity_gpu import _oracle_mnf_net, _to64
    from bnn_amd import layers as L
    monkeypatch.setattr(L, "_F16_FIRST_PLANES", first == "planes")
    net, layers, x, noises, P, zf, rf = _headline_case(bnn)
    ref64, kls64 = _oracle_mnf_net(*_to64(x, P, zf, rf, noises))
    net = net.to(dev).train()
    for l, n in zip(layers, noises):
        l.noise = {k: v.to(dev) for k, v in n.items()}
    net.set_precision(prec)
    assert bnn.get_precision() == "fp32" and bnn.get_precision(net.l2) == prec          # per network, not global
    with torch.no_grad():
        out = net(x.to(dev), sample=True)
        assert net.l1._split_now == net.l2._split_now == (3 if prec == "fp16x3f" else 2) and net.l3._split_now == 0
        kls = [l.kl.clone() for l in layers]
    bar, atol = BARS[prec]
    # the log-probabilities are O(2.3): the bars apply to the hidden activations; the head is the exact fp32 skinny kernel
    assert rel_err(out, ref64) < bar, rel_err(out, ref64)
    assert elementwise_violation(out, ref64, rtol=1e-4, atol_frac=atol) <= 1.0
    for k, k_ref in zip(kls, kls64):
        assert rel_err(k, k_ref) < 1e-4
    # hidden activations themselves, layer by layer through the autograd Function (fp32 x in, fp32 out + std for the backward)
    h1 = net.l1(x.to(dev), sample=True, _relu=True)
    ref_h1 = _hidden_ref(x, P[0], zf[0], noises[0])
    assert rel_err(h1, ref_h1) < bar, rel_err(h1, ref_h1)
    assert elementwise_violation(h1, ref_h1, rtol=1e-4, atol_frac=atol) <= 1.0
    loss = h1.sum() + net.l1.kl
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in net.l1.parameters() if p.grad is not None)


def _hidden_ref(x, P, zflow, noise):
    """relu(layer-1 activations) of the MNF layer in fp64 (LBBNN-GP-MF-MNF.py:190-200) on the draws of `noise`."""
    p = {k: v.double() for k, v in P.items() if torch.is_tensor(v) and v.is_floating_point()}
    z0 = p["q0_mean"] + torch.exp(p["q0_log_var"]).sqrt() * noise["eps_z"].double()[-1]
    z = z0
    for tr in zflow.transforms:
        u, w, b = tr["u"].double(), tr["w"].double(), tr["bias"].double()
        z = z + u * torch.tanh(torch.dot(w, z) + b[0])
    alpha = orc.alpha_of(p["lambdal"]); sigma = orc.sigma_of(p["weight_rho"])
    x64 = x.double()
    e_b = (x64 * z) @ (p["weight_mu"] * alpha).T + p["bias_mu"]
    var_b = (x64 ** 2) @ (sigma ** 2 * alpha ** 2).T + orc.sigma_of(p["bias_rho"]) ** 2
    return torch.relu(e_b + var_b.sqrt() * noise["eps_out"].double())


@pytest.mark.parametrize("prec", ["fp16x3", "fp16x3f"])
def test_f16_row_sharded_forward_and_plan_and_graph_bitwise(bnn, dev, prec):
    """With in-kernel noise: (1) a 4096-row forward == its two 2048-row shards run with set_row_offset (data-parallel
    contract) bit for bit, KL included; (2) a recorded LaunchPlan and (3) a HIP-graph replay reproduce the
    eager forward bit for bit from the same Philox {seed, offset}."""
    from bnn_amd import graphs
    dims, B = (784, 1200, 1200, 10), 4096
    torch.manual_seed(5)
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    net.set_precision(prec)
    x = torch.rand(B, 784, device=dev)
    with torch.no_grad():
        bnn.manual_seed(99, 3)
        full = net(x, sample=True).clone()
        kl_full = net.kl().clone()
        assert net.l1._split_now >= 2 and net.l2._split_now >= 2
        parts = []
        for lo in (0, B // 2):
            bnn.manual_seed(99, 3)
            net.set_row_offset(lo)
            parts.append(net(x[lo:lo + B // 2], sample=True).clone())
            assert torch.equal(net.kl(), kl_full)
        net.set_row_offset(0)
        assert torch.equal(torch.cat(parts), full)
        plan = graphs.LaunchPlan(net, x, sample=True)
        bnn.manual_seed(99, 3)
        o, k = plan()
        assert torch.equal(o, full) and torch.equal(k, kl_full)
        o2 = plan()[0].clone()
        assert not torch.equal(o2, full)                      # fresh noise on the next call
        net(x, sample=True)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            og = net(x, sample=True)
            kg = net.kl()
        bnn.manual_seed(99, 3)
        gr.replay()
        assert torch.equal(og, full) and torch.equal(kg, kl_full)


def test_f16_falls_back_where_the_format_does_not_apply(bnn, dev):
    """Posterior-mean forwards (no variance product), rows wider than one register batch of the weight pass and layers whose
    shapes the split kernels do not take keep the other formats, silently and correctly."""
    torch.manual_seed(1)
    net = bnn.lrt.BayesianNetwork((784, 400, 400, 10)).to(dev)
    net.set_precision("fp16x3")
    x = torch.rand(64, 784, device=dev)
    net.eval()
    with torch.no_grad():
        det = net(x, sample=False)
        assert net.l1._split_now == 1 and net.l2._split_now == 1            # mean-only: bf16x3
        net.set_precision("fp32")
        assert rel_err(det, net(x, sample=False)) < 2e-5
    wide = bnn.lrt.BayesianNetwork((784, 1600, 1600, 10)).to(dev).train()
    wide.set_precision("fp16x3f")
    with torch.no_grad():
        wide(x, sample=True)
    assert wide.l1._split_now == 1 and wide.l2._split_now == 1             # l3 rows of 1600 weights: the whole net keeps bf16x3


@pytest.mark.parametrize("prec", ["fp16x3", "fp16x3f"])
@pytest.mark.parametrize("dims,B", [((784, 1200, 1200, 10), 4096), ((64, 96, 40, 7), 100), ((784, 400, 88, 16), 33)])
def test_head_fold_equals_the_unfolded_forward(bnn, dev, prec, dims, B, monkeypatch):
    """The 10-class head folded into the previous GEMM's epilogue (lbbnn_gemm_desc_t::head_*, layers._HEAD_FOLD) against the
    same forward with the head as its own launch (the exact-fp32 skinny kernel on the stored activations): same draws
    (in-kernel Philox, same counters), outputs within the format's bar of each other, KL bit-identical; and the folded path
    never stores the last hidden activation.  Odd sizes: O not a multiple of 80, 7 and 16 classes, ragged batch."""
    from bnn_amd import layers as L
    torch.manual_seed(3)
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    net.set_precision(prec)
    x = torch.rand(B, dims[0], device=dev)
    res = {}
    for fold in (True, False):
        monkeypatch.setattr(L, "_HEAD_FOLD", fold)
        with torch.no_grad():
            bnn.manual_seed(7, 5)
            out = net(x, sample=True).clone()
            res[fold] = (out, net.kl().clone())
        assert net.l2._split_now >= 2 and net.l3._split_now == 0
    bar, atol = BARS[prec]
    assert torch.isfinite(res[True][0]).all()
    assert rel_err(res[True][0], res[False][0]) < bar, rel_err(res[True][0], res[False][0])
    assert elementwise_violation(res[True][0], res[False][0].double(), rtol=1e-4, atol_frac=atol) <= 1.0
    assert torch.equal(res[True][1], res[False][1])
    assert float((res[True][0].exp().sum(dim=1) - 1).abs().max()) < 1e-5            # rows are log-probabilities


def test_fp16x3f_keeps_three_variance_products_on_short_rows(bnn, dev):
    """"fp16x3f" is the 3 + 1 form from ops.F16_VAR1_MIN_I weights per row up: the roundings of the single variance product
    average out as 1 / sqrt(I) (tools/gemm16_fuzz.py: 8e-5 of max|out| on the raw entry point at I = 8 ... 104, 1.4-1.8e-5 at
    the headline's 784 / 1200).  A shorter row takes format 2 (all three products) under the same precision setting and
    meets the 3 + 3 form's bar."""
    from bnn_amd import ops
    assert ops.F16_VAR1_MIN_I == 256
    for I, want in ((96, 2), (248, 2), (256, 3), (784, 3)):
        torch.manual_seed(I)
        layer = bnn.mnf.BayesianLinear(I, 80, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
        layer.precision = "fp16x3f"
        g = torch.Generator().manual_seed(I + 1)
        x = torch.rand(64, I, generator=g)
        noise = {"eps_z": torch.randn(1, I, generator=g), "eps_out": torch.randn(64, 80, generator=g),
                 "eps_z2": torch.randn(1, I, generator=g), "eps_act": torch.randn(80, generator=g)}
        layer.noise = {k: v.to(dev) for k, v in noise.items()}
        with torch.no_grad():
            out = layer(x.to(dev), sample=True)
        assert layer._split_now == want, (I, layer._split_now)
        p = {k: v.detach().cpu().double() for k, v in layer.state_dict().items()}
        zf, rf = orc.flow_from_state("z_flow", "Planar", p, 2), orc.flow_from_state("r_flow", "Planar", p, 2)
        ref, _, _ = orc.mnf_forward(x.double(), p, zf, rf, {k: v.double() for k, v in noise.items()})
        assert rel_err(out, ref) < (BARS["fp16x3"][0] if want == 2 else BARS["fp16x3f"][0]), (I, rel_err(out, ref))


@pytest.mark.parametrize("dims", [(784, 80, 17, 10), (100, 33, 64, 16), (784, 1400, 64, 10)])
def test_network_with_a_layer_outside_the_fp16_row_kernel_keeps_bf16x3(bnn, dev, dims):
    """The network's weight pass is ONE launch; the row-scaled fp16 format exists in its vector row kernel only (rows of
    whole float4s, at most 1280 weights).  A network with one layer outside that (a 17- or 33-wide row, a 1400-wide one)
    runs its eligible layers in bf16x3 under an fp16 precision setting, training and inference, instead of failing in
    lbbnn_layers_operands_snap (found by tools/net_train_fuzz.py)."""
    torch.manual_seed(3)
    net = bnn.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar").to(dev).train()
    net.set_precision("fp16x3f")
    g = torch.Generator().manual_seed(4)
    B = 64
    x = torch.rand(B, dims[0], generator=g)
    y = torch.randint(0, dims[3], (B,), generator=g)
    lay = [net.l1, net.l2, net.l3]
    noises = [{"eps_z": torch.randn(1, l.in_features, generator=g), "eps_out": torch.randn(B, l.out_features, generator=g),
               "eps_z2": torch.randn(1, l.in_features, generator=g), "eps_act": torch.randn(l.out_features, generator=g)} for l in lay]
    for l, n in zip(lay, noises):
        l.noise = {k: v.to(dev) for k, v in n.items()}
    out = net(x.to(dev), sample=True)
    assert all(l._split_now in (0, 1) for l in lay) and net.l1._split_now == (1 if dims[0] % 8 == 0 else 0)
    loss = torch.nn.functional.nll_loss(out, y.to(dev), reduction="sum") + net.kl() / 10
    loss.backward()
    P = [{k: v.detach().cpu().double() for k, v in l.state_dict().items()} for l in lay]
    zf = [orc.flow_from_state("z_flow", "Planar", p, 2) for p in P]
    rf = [orc.flow_from_state("r_flow", "Planar", p, 2) for p in P]
    ref, ref_kl = orc.mnf_network_forward(x.double(), P, zf, rf, [{k: v.double() for k, v in n.items()} for n in noises])
    assert rel_err(out, ref) < 2e-6
    with torch.no_grad():
        out2 = net(x.to(dev), sample=True)
    assert rel_err(out2, ref) < 2e-6
