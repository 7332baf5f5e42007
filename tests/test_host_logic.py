"""CPU tier: host-side logic of the drop-in modules (construction, seeded parity with the
reference's initial values, state_dict names, loud failure without a GPU)."""
import os

import pytest
import torch

from conftest import ROOT, sub


def test_lrt_seeded_construction_matches_reference(golden):
    """torch.manual_seed(s); BayesianLinear(I,O) yields the reference's values (creation order
    LBBNN-GP-MF-LRT.py:137-155) -- checked against the parameters stored in the golden file."""
    import bnn_amd
    g = golden("lrt.npz")
    for ci, (B, I, O) in enumerate([(3, 6, 4), (5, 33, 17), (8, 784, 10), (4, 12, 20)]):
        c = g.case("c%d" % ci)
        torch.manual_seed(100 + ci)
        layer = bnn_amd.lrt.BayesianLinear(I, O)
        sd = layer.state_dict()
        ref = sub(c, "p.")
        assert sorted(sd) == sorted(ref)
        for k in ref:
            assert torch.equal(sd[k], ref[k]), k


def test_mnf_seeded_construction_matches_reference(golden):
    import bnn_amd
    g = golden("mnf.npz")
    cases = [("Planar", 3, 6, 4, 2), ("Planar", 5, 33, 17, 2), ("Planar", 8, 784, 10, 2), ("Planar", 4, 40, 24, 3),
             ("RNVP", 3, 6, 4, 2), ("RNVP", 5, 33, 17, 2), ("MNF", 3, 6, 4, 2), ("MNF", 5, 33, 17, 2)]
    for ci, (kind, B, I, O, T) in enumerate(cases):
        c = g.case("c%d" % ci)
        torch.manual_seed(200 + ci)
        layer = bnn_amd.mnf.BayesianLinear(I, O, T, z_flow_type=kind, r_flow_type=kind)
        sd = layer.state_dict()
        ref = sub(c, "p.")
        assert sorted(sd) == sorted(ref), kind
        for k in ref:
            assert torch.equal(sd[k], ref[k]), (kind, k)


def test_network_structure_and_names():
    import bnn_amd
    net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="Planar", r_flow_type="Planar")
    names = [n for n, _ in net.named_parameters()]
    assert "l1.weight_mu" in names and "l3.r_flow.transforms.1.bias" in names
    assert net.l1.kl == 0 and net.kl() == 0
    n_params = sum(p.numel() for p in net.parameters())
    # 3*O*I + 2*O + 5*I + 2 flows * T * (2I+1)   
    want = sum(3 * o * i + 2 * o + 5 * i + 4 * (2 * i + 1) for i, o in [(784, 1200), (1200, 1200), (1200, 10)])
    assert n_params == want
    ref_net = bnn_amd.lrt.BayesianNetwork()
    assert (ref_net.l1.in_features, ref_net.l2.out_features, ref_net.l3.out_features) == (784, 600, 10)


def test_network_level_rule_for_the_row_scaled_fp16_format():
    """One weight-pass launch per network: the fp16 format is chosen only when EVERY layer fits its vector row kernel (rows
    of whole float4s, at most 1280 weights) -- layers._number_layers; the headline network does."""
    import bnn_amd
    for dims, ok in (((784, 1200, 1200, 10), True), ((784, 80, 17, 10), False), ((100, 33, 64, 16), False),
                     ((784, 1400, 64, 10), False), ((784, 96, 64, 10), True)):
        net = bnn_amd.mnf.BayesianNetwork(dims, 2, z_flow_type="Planar", r_flow_type="Planar")
        assert all(l._f16s_net_ok == ok for l in net._layers()), dims
        net.set_precision("fp16x3f")
        want = ((3 if dims[0] >= 256 else 2) if ok else 1) if dims[0] % 8 == 0 else 0
        assert net.l1._split(None, (True, True, False)) == want, dims


def test_eval_helpers_match_reference_objects():
    import bnn_amd
    torch.manual_seed(0)
    l = bnn_amd.lrt.BayesianLinear(12, 5)
    a = 1 / (1 + torch.exp(-l.lambdal))
    assert torch.equal(l.alpha_q, a) and torch.equal(l.gamma.alpha, a)
    gs = l.gamma.rsample()                      # test_ensemble's density probe (…LRT.py:242-246)
    assert set(gs.unique().tolist()) <= {0.0, 1.0}
    assert torch.allclose(l.weight.sigma, torch.log1p(torch.exp(l.weight_rho)))
    assert l.gamma.exact is True


def test_forward_fails_loudly_on_cpu():
    import bnn_amd
    l = bnn_amd.lrt.BayesianLinear(12, 5)
    with pytest.raises(RuntimeError, match="no CPU"):
        l(torch.rand(2, 12))
    m = bnn_amd.mnf.BayesianLinear(12, 5, 2, z_flow_type="Planar", r_flow_type="Planar")
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.rand(2, 12), sample=True)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bnn_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="not built"):
        _lib.lib()


def test_product_package_never_imports_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "bayesian-neural-nets_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_base_and_vd_seeded_construction_matches_reference(golden):
    import bnn_amd
    g = golden("base.npz")
    for ci, (B, I, O) in enumerate([(3, 6, 4), (5, 33, 17), (8, 784, 10)]):
        c = g.case("c%d" % ci)
        torch.manual_seed(500 + ci)
        layer = bnn_amd.base.BayesianLinear(I, O, 1)
        ref = sub(c, "p.")
        sd = layer.state_dict()
        assert sorted(sd) == sorted(ref)
        for k in ref:
            assert torch.equal(sd[k], ref[k]), k
        assert torch.equal(layer.alpha, c["alpha_init"])            # placeholder draw, LBBNN-GP-MF.py:204
    g = golden("vd.npz")
    for ci, (B, I, O) in enumerate([(3, 6, 4), (5, 33, 17), (8, 300, 10)]):
        c = g.case("c%d" % ci)
        torch.manual_seed(600 + ci)
        layer = bnn_amd.vd.BayesianLayer(I, O)
        assert torch.equal(layer.theta.detach(), c["theta"])
        assert torch.equal(layer.alpha, c["alpha"])
        assert [n for n, _ in layer.named_parameters()] == ["theta"]      # alpha is NOT a parameter there either


def test_product_sources_carry_no_lab_code_and_export_no_lab_symbol():
    """the measurement hooks (-DLAB_* ablation returns, stamps, role logic) are gone from the product
    kernel sources -- the lab builds of tools/lab compile the round-2 tree fetched from history -- and the product library
    exports nothing named lbbnn_lab_*."""
    import glob
    import re
    import subprocess
    csrc = os.path.join(ROOT, "bayesian-neural-nets_amd", "csrc")
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        txt = open(f).read()
        assert not re.search(r"\bLAB_[A-Z0-9_]+\b", txt), f
        assert "lbbnn_lab_" not in txt, f
    so = os.path.join(csrc, "liblbbnn_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert "lbbnn_lrt_gemm_ex" in out and "lbbnn_lab" not in out


def test_prior_helpers_log_prob_match_the_reference_formulas():
    """GaussGamma.log_prob / BetaBinomial.log_prob (LBBNN-GP-MF.py:140-151, 162-173) as callable torch densities (VERDICT r02
    missing #3): checked on CPU against the formulas written out independently, exact and relaxed gates."""
    import math
    import torch
    from bnn_amd import base
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 5, generator=g, dtype=torch.float64)
    gam = torch.rand(7, 5, generator=g, dtype=torch.float64)
    a, b = torch.tensor([1.05], dtype=torch.float64), torch.tensor([1.02], dtype=torch.float64)
    tau = torch.tensor([0.7], dtype=torch.float64)
    for exact in (False, True):
        gg = base.GaussGamma(a, b); gg.exact = exact
        gq = torch.round(gam) if exact else gam
        ref = (gq * (a * torch.log(b) + (a - 0.5) * tau - b * tau - torch.lgamma(a) - 0.5 * math.log(2 * math.pi))
               - tau * x ** 2 + (1 - gq) + 1e-8).sum()
        assert torch.allclose(gg.log_prob(x, gam, tau=tau), ref, rtol=1e-12)
        bb = base.BetaBinomial(a, b); bb.exact = exact
        lg = torch.lgamma
        one = torch.ones_like(gam)
        ref = (lg(one) + lg(gq + a) + lg(1 + b - gq) + lg(one * (a + b)) - lg(a + gq) - lg(2 - gq) - lg(one * (1 + a + b))
               - lg(one * a) - lg(one * b)).sum()
        assert torch.allclose(bb.log_prob(gam, pa=None, pb=None), ref, rtol=1e-12)
    assert torch.isfinite(base.GaussGamma(a, b).log_prob(x, gam))            # draws its own tau (:141)
