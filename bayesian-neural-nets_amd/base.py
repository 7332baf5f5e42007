"""Baseline LBBNN (explicit latent-binary gate x Gaussian weight sampling), names as in LBBNN-GP-MF.py:

``BayesianLinear(in_features, out_features, layer_id)`` :182-255 -- ``forward(input, cgamma, sample=False,
medimean=False, calculate_log_probs=False)``, attributes ``log_prior``, ``log_variational_posterior``,
``alpha``, ``gammas``, ``gamma``, ``weight_prior``, ``bias_prior``, ``gamma_prior`` (each with ``.exact``);
``BayesianNetwork`` :259-319 with ``sample_elbo``.  The fused pass (sample W = gamma*(mu+sigma*eps), all four
Monte-Carlo log-probability sums) and ``F.linear`` run in HIP (lbbnn_gate_sample + lbbnn_lrt_gemm).
"""
import ctypes
import itertools
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .distributions import Bernoulli, Gaussian, TEMPER_PRIOR  # noqa: F401  (TEMPER_PRIOR re-exported: the reference scripts read it here)

_ids = itertools.count(32)
SAMPLES = 1              # LBBNN-GP-MF.py:38
# torch.distributions' argument checks of the Gamma / Beta / RelaxedBernoulli draws below: each is a compare + reduce + a
# host read-back of one bool -- ~35 device-to-host syncs and ~60 launches per sample_elbo step on a GPU (the reference runs on the
# CPU, where they cost nothing).  The parameters checked are softplus / sigmoid outputs and the positive prior constants: a
# check can only fire on NaN parameters.  True restores torch's default behaviour (a ValueError at the draw).
VALIDATE_ARGS = False
NUM_BATCHES = 600        # len(train_loader) with BATCH_SIZE = 100 on MNIST (LBBNN-GP-MF.py:34,67)


class GaussGamma(object):
    """Normal-Gamma prior helper (LBBNN-GP-MF.py:132-151): a, b and the ``exact`` switch.  On the hot path the density is
    evaluated inside the HIP pass (lbbnn_gate_sample) with the Gamma(a, b) draw of :141 taken by ``rsample_tau``;
    ``log_prob`` is the same density as torch ops, for callers that evaluate the prior outside ``forward`` and for the
    differentiable posterior-mean branches of the layer."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.exact = False

    def rsample_tau(self):
        return torch.distributions.Gamma(self.a, self.b, validate_args=VALIDATE_ARGS).rsample()

    def log_prob(self, input, gamma, tau=None):
        """:140-151.  ``tau``: the Gamma(a, b) draw (default: a fresh ``rsample``, as the reference draws one per call)."""
        if tau is None:
            tau = self.rsample_tau()                                                     # :141
        g = torch.round(gamma.detach()) if self.exact else gamma                         # :142-143
        a, b = self.a, self.b
        return (g * (a * torch.log(b) + (a - 0.5) * tau - b * tau - torch.lgamma(a)
                     - 0.5 * math.log(2 * math.pi)) - tau * torch.pow(input, 2) + (1 - g) + 1e-8).sum()   # :144-150


class BetaBinomial(object):
    """LBBNN-GP-MF.py:155-179."""

    def __init__(self, pa, pb):
        self.pa, self.pb = pa, pb
        self.exact = False

    def log_prob(self, input, pa=None, pb=None):
        """:162-173 (the reference takes pa / pb arguments and ignores them in favour of the attributes; so does this)."""
        g = torch.round(input.detach()) if self.exact else input
        one = torch.ones_like(input)
        pa_, pb_ = self.pa, self.pb
        return (torch.lgamma(one) + torch.lgamma(g + one * pa_) + torch.lgamma(one * (1 + pb_) - g)
                + torch.lgamma(one * (pa_ + pb_)) - torch.lgamma(one * pa_ + g) - torch.lgamma(one * 2 - g)
                - torch.lgamma(one * (1 + pa_ + pb_)) - torch.lgamma(one * pa_) - torch.lgamma(one * pb_)).sum()

    def rsample(self):
        p = torch.distributions.Beta(self.pa, self.pb, validate_args=VALIDATE_ARGS).rsample()
        return torch.distributions.RelaxedBernoulli(probs=p, temperature=0.001, validate_args=VALIDATE_ARGS).rsample()


class _BaseFn(torch.autograd.Function):
    """Forward: lbbnn_gate_sample + the mean-only GEMM.  Backward (round 2: all on the HIP kernels; round 1 recomputed the
    layer with torch ops): lbbnn_output_grad (G^T, column sums) -> dW = G^T x on the GEMM kernel -> lbbnn_gate_backward
    (K6b: every (O,I)-, (O)- and scalar-sized gradient in one pass + tail, and the sampled W as a dense matrix) ->
    dX = G W on the GEMM kernel.  ``galpha`` is ``layer.gamma.alpha`` as an autograd input: Bernoulli.log_prob
    differentiates through it (LBBNN-GP-MF.py:125-127, alpha = sigmoid(lambdal) set by sample_elbo :292-297)."""

    @staticmethod
    def forward(ctx, layer, x, cgamma, tau_w, tau_b, galpha, cfg, *params):
        out, lp, lq, saved = layer._forward_hip(x, cgamma, tau_w, tau_b, cfg, save_rng=True)
        ctx.layer, ctx.cfg, ctx.saved = layer, cfg, saved
        from . import graphs
        graphs.mark_autograd_node(ctx, layer)          # capture guard: graphs.assert_no_live_graph
        ctx.save_for_backward(x, tau_w, tau_b, *params)
        z = out.new_zeros(())
        return out, (lp if lp is not None else z), (lq if lq is not None else z)

    @staticmethod
    def backward(ctx, g_out, g_lp, g_lq):
        from .layers import _hip_matmul_nt
        layer, cfg, saved = ctx.layer, ctx.cfg, ctx.saved
        x, tau_w, tau_b, *params = ctx.saved_tensors
        P = dict(zip(layer._names, params))
        O, I, dev = layer.out_features, layer.in_features, x.device
        f = dict(dtype=torch.float32, device=dev)
        noise = saved.get("noise") or {}
        gm, _, gmT, _, g_sum, _ = ops.output_grad(g_out.contiguous())
        dW = _hip_matmul_nt(gmT, ops.transpose_operand, x)                       # (O,I) = G^T x
        a = _lib.GateBwdArgs()
        keep = []

        def dptr(t):
            if t is None:
                return None
            u = t.detach()
            if u.dtype != torch.float32 or not u.is_contiguous():
                u = u.float().contiguous()
            keep.append(u)
            return ops._ptr(u, "tensor")
        a.mu, a.rho = dptr(P["weight_mu"]), dptr(P["weight_rho"])
        a.gamma_alpha, a.cgamma = dptr(saved["gamma_alpha"]), dptr(saved["cg"])
        a.eps_w, a.eps_b = dptr(noise.get("eps_w")), dptr(noise.get("eps_b"))
        a.bias_mu, a.bias_rho, a.bias_a, a.bias_b = dptr(P["bias_mu"]), dptr(P["bias_rho"]), dptr(P["bias_a"]), dptr(P["bias_b"])
        a.tau_b, a.tau_w = dptr(tau_b), dptr(tau_w)
        a.weight_a, a.weight_b, a.pa, a.pb = dptr(P["weight_a"]), dptr(P["weight_b"]), dptr(P["pa"]), dptr(P["pb"])
        a.dW, a.g_sum = dW.data_ptr(), g_sum.data_ptr()
        a.g_lp, a.g_lq = dptr(g_lp.reshape(1)), dptr(g_lq.reshape(1))
        outs = {n: torch.empty((O, I), **f) for n in ("d_mu", "d_rho", "d_cgamma", "d_alpha", "w_out")}
        vecs = {n: torch.empty(O, **f) for n in ("d_bias_mu", "d_bias_rho", "d_bias_a", "d_bias_b", "d_tau_b")}
        scal, rows = torch.empty(5, **f), torch.empty(3 * O, **f)
        for n, t in list(outs.items()) + list(vecs.items()):
            setattr(a, n, t.data_ptr())
        a.d_scalars, a.rows = scal.data_ptr(), rows.data_ptr()
        a.O, a.I, a.exact, a.layer_id = O, I, cfg[2], layer._layer_id
        rng = saved.get("rng")
        _lib.check(_lib.lib().lbbnn_gate_backward(ctypes.byref(a), rng.data_ptr() if rng is not None else None, ops._stream()),
                   "lbbnn_gate_backward")
        del keep
        gx = _hip_matmul_nt(gm, ops.transpose_operand, outs["w_out"]) if ctx.needs_input_grad[1] else None   # G W
        grads = {"weight_mu": outs["d_mu"], "weight_rho": outs["d_rho"], "weight_a": scal[0:1], "weight_b": scal[1:2],
                 "lambdal": None, "pa": scal[3:4], "pb": scal[4:5], "bias_mu": vecs["d_bias_mu"],
                 "bias_rho": vecs["d_bias_rho"], "bias_a": vecs["d_bias_a"], "bias_b": vecs["d_bias_b"]}
        return (None, gx, outs["d_cgamma"], scal[2:3].reshape(tau_w.shape), vecs["d_tau_b"].reshape(tau_b.shape),
                outs["d_alpha"], None, *[grads[n] for n in layer._names])


class BayesianLinear(nn.Module):
    _names = ("weight_mu", "weight_rho", "weight_a", "weight_b", "lambdal", "pa", "pb",
              "bias_mu", "bias_rho", "bias_a", "bias_b")

    def __init__(self, in_features, out_features, layer_id):
        super().__init__()
        self.layer = layer_id
        self.in_features, self.out_features = in_features, out_features
        O, I = out_features, in_features
        # creation order == LBBNN-GP-MF.py:192-219 (seeded construction reproduces the reference's values)
        self.weight_mu = nn.Parameter(torch.Tensor(O, I).uniform_(-0.2, 0.2))
        self.weight_rho = nn.Parameter(torch.Tensor(O, I).uniform_(-5, -4))
        self.weight = Gaussian(self.weight_mu, self.weight_rho)
        self.weight_a = nn.Parameter(torch.Tensor(1).uniform_(1, 1.1))
        self.weight_b = nn.Parameter(torch.Tensor(1).uniform_(1, 1.1))
        self.weight_prior = GaussGamma(self.weight_a, self.weight_b)
        self.lambdal = nn.Parameter(torch.Tensor(O, I).uniform_(0, 1))
        self.gammas = torch.Tensor(O, I).uniform_(0.99, 1)
        self.alpha = torch.Tensor(O, I).uniform_(0.999, 0.9999)
        self.gamma = Bernoulli(self.alpha, exact=False)
        self.pa = nn.Parameter(torch.Tensor(1).uniform_(1, 1.1))
        self.pb = nn.Parameter(torch.Tensor(1).uniform_(1, 1.1))
        self.gamma_prior = BetaBinomial(pa=self.pa, pb=self.pb)
        self.bias_mu = nn.Parameter(torch.Tensor(O).uniform_(-0.2, 0.2))
        self.bias_rho = nn.Parameter(torch.Tensor(O).uniform_(-5, -4))
        self.bias = Gaussian(self.bias_mu, self.bias_rho)
        self.bias_a = nn.Parameter(torch.Tensor(O).uniform_(1, 1.1))
        self.bias_b = nn.Parameter(torch.Tensor(O).uniform_(1, 1.1))
        self.bias_prior = GaussGamma(self.bias_a, self.bias_b)
        self.log_prior = 0
        self.log_variational_posterior = 0
        self.lagrangian = 0
        self.noise = None              # parity tests: {"eps_w","eps_b","tau_w","tau_b"}
        self._layer_id = next(_ids) % 64
        self._ws = None

    def _workspace(self):
        dev = self.weight_mu.device
        if self._ws is None or self._ws["dev"] != dev:
            O, ld = self.out_features, ops.operand_ld(self.in_features)
            f = dict(dtype=torch.float32, device=dev)
            self._ws = {"dev": dev, "w": torch.empty((O, ld), **f), "rows": torch.empty(4 * O, **f)}
        return self._ws

    def _exact_bits(self):
        return ((1 if self.weight_prior.exact else 0) | (2 if self.bias_prior.exact else 0)
                | (4 if self.gamma_prior.exact else 0) | (8 if self.gamma.exact else 0))

    def _forward_hip(self, x, cgamma, tau_w, tau_b, cfg, save_rng=False):
        mode, want_lp, exact = cfg
        ws = self._workspace()
        dev = x.device
        noise = self.noise or {}
        O, I = self.out_features, self.in_features
        split = (ops.split_precision() and ops.split_eligible(I, O)
                 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)
        a = _lib.GateArgs()
        P = lambda t: None if t is None else ops._ptr(t.detach() if t.requires_grad else t, "tensor")
        gamma_alpha = self.gamma.alpha
        gamma_alpha = gamma_alpha.detach().to(dev).float().contiguous() if torch.is_tensor(gamma_alpha) else None
        alpha_attr = self.alpha.detach().to(dev).float().contiguous() if torch.is_tensor(self.alpha) else None
        cg = None if cgamma is None else cgamma.detach().to(dev).float().contiguous()
        eps_w, eps_b = noise.get("eps_w"), noise.get("eps_b")
        rng, st, saved = None, None, {"noise": self.noise, "alpha_attr": alpha_attr, "gamma_alpha": gamma_alpha}
        if mode == 0 and (eps_w is None or eps_b is None):
            st = ops.RngState.get(dev)
            rng = st.t
            saved["rng"] = rng.clone() if save_rng else None
        bias = torch.empty(O, dtype=torch.float32, device=dev)
        lp = torch.empty((), dtype=torch.float32, device=dev) if want_lp else None
        lq = torch.empty((), dtype=torch.float32, device=dev) if want_lp else None
        a.mu, a.rho, a.gamma_alpha, a.cgamma = P(self.weight_mu), P(self.weight_rho), P(gamma_alpha), P(cg)
        a.eps_w, a.alpha_attr = P(eps_w), P(alpha_attr)
        a.bias_mu, a.bias_rho, a.eps_b = P(self.bias_mu), P(self.bias_rho), P(eps_b)
        a.bias_a, a.bias_b, a.tau_b = P(self.bias_a), P(self.bias_b), P(tau_b)
        a.weight_a, a.weight_b, a.tau_w, a.pa, a.pb = P(self.weight_a), P(self.weight_b), P(tau_w), P(self.pa), P(self.pb)
        a.w_out, a.bias_out, a.rows = ws["w"].data_ptr(), bias.data_ptr(), ws["rows"].data_ptr()
        a.log_prior, a.log_q = P(lp), P(lq)
        a.O, a.I, a.ld, a.mode, a.exact, a.want_lp = O, I, ops.operand_ld(I), mode, exact, int(want_lp)
        a.flags, a.layer_id = (ops.F_SPLIT16 if split else 0), self._layer_id
        _lib.check(_lib.lib().lbbnn_gate_sample(ctypes.byref(a), rng.data_ptr() if rng is not None else None,
                                                ops._stream()), "lbbnn_gate_sample")
        out = ops.lrt_gemm(x, ws["w"], None, I=I, O=O, bias_mean=bias, mean_only=True, split=split)   # F.linear :255
        if st is not None:
            st.advance(1)
        saved["cg"] = cg
        return out, lp, lq, saved

    def _mean_branch_autograd(self, x, cgamma, tau_w, tau_b, mode):
        """LBBNN-GP-MF.py:236-255 for the two deterministic branches, as autograd-visible torch ops (GPU tensors)."""
        alpha_attr = self.alpha.to(x.device) if torch.is_tensor(self.alpha) else self.alpha
        weight = cgamma * self.weight.mu if mode == 1 else alpha_attr * self.weight.mu           # :236-242
        bias = self.bias.mu
        alpha_new = 1 / (1 + torch.exp(-self.lambdal))                                             # :246
        lp = (self.weight_prior.log_prob(weight, cgamma, tau=tau_w) + self.bias_prior.log_prob(bias, torch.ones_like(bias), tau=tau_b)
              + self.gamma_prior.log_prob(cgamma, pa=self.pa, pb=self.pb))                         # :247-249
        galpha = self.gamma.alpha
        gm = Bernoulli(galpha.to(x.device) if torch.is_tensor(galpha) else galpha, exact=self.gamma.exact)
        lq = self.weight.full_log_prob(input=weight, gamma=cgamma) + gm.log_prob(cgamma) + self.bias.log_prob(bias)   # :250-251
        del alpha_new
        return F.linear(x, weight, bias), lp, lq                                                   # :255

    def _noise_for_backward(self, saved):
        n = dict(saved.get("noise") or {})
        if "eps_w" not in n and saved.get("rng") is not None:
            O, I, L = self.out_features, self.in_features, self._layer_id
            n["eps_w"] = ops.philox_normal(saved["rng"], ops.STREAM_EPS_W * 64 + L, O, I)
            n["eps_b"] = ops.philox_normal(saved["rng"], ops.STREAM_EPS_B * 64 + L, 0, O)
        return n

    def forward(self, input, cgamma, sample=False, medimean=False, calculate_log_probs=False):
        if not input.is_cuda:
            raise RuntimeError("bnn_amd: forward needs a HIP device tensor (input is on %s); there is no CPU path"
                               % input.device)
        if self.training or sample:
            self.gammas = cgamma                                      # :231
            mode = 0
        elif medimean:
            mode = 1
        else:
            mode = 2
        want_lp = bool(self.training or calculate_log_probs)
        noise = self.noise or {}
        dev = input.device
        tau_w = tau_b = None
        if want_lp:
            tau_w = noise["tau_w"] if "tau_w" in noise else self.weight_prior.rsample_tau()      # :141
            tau_b = noise["tau_b"] if "tau_b" in noise else self.bias_prior.rsample_tau()
        cfg = (mode, want_lp, self._exact_bits())
        x = input.float()
        params = [getattr(self, n) for n in self._names]
        needs = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        if needs and want_lp and mode == 0:
            galpha = self.gamma.alpha if torch.is_tensor(self.gamma.alpha) else torch.as_tensor(self.gamma.alpha)
            out, lp, lq = _BaseFn.apply(self, x, cgamma.to(dev), tau_w, tau_b, galpha.to(dev), cfg, *params)
        else:
            if needs and want_lp:
                # medimean / joint-mean branch with calculate_log_probs under autograd (LBBNN-GP-MF.py:236-251): not a path
                # the reference's train() differentiates (:331-337 samples), so it is composed from differentiable torch
                # ops on the GPU tensors (the helper objects' densities); every other combination runs the HIP pass
                out, lp, lq = self._mean_branch_autograd(x, cgamma.to(dev), tau_w, tau_b, mode)
                self.alpha = 1 / (1 + torch.exp(-self.lambdal))
                self.log_prior, self.log_variational_posterior = lp, lq
                return out
            out, lp, lq, _ = self._forward_hip(x, cgamma, tau_w, tau_b, cfg)
        if want_lp:
            self.alpha = 1 / (1 + torch.exp(-self.lambdal))           # :246
            self.log_prior, self.log_variational_posterior = lp, lq
        else:
            self.log_prior, self.log_variational_posterior = 0, 0     # :253
        return out


class BayesianNetwork(nn.Module):
    """LBBNN-GP-MF.py:259-319 (reference dims 784-400-600-10; ``dims=`` added)."""

    def __init__(self, dims=(28 * 28, 400, 600, 10)):
        super().__init__()
        self.dims = tuple(dims)
        self.l1 = BayesianLinear(dims[0], dims[1], 1)
        self.l2 = BayesianLinear(dims[1], dims[2], 1)
        self.l3 = BayesianLinear(dims[2], dims[3], 1)
        for i, l in enumerate((self.l1, self.l2, self.l3)):
            l._layer_id = 32 + i              # per-network Philox stream ids (not the process-wide counter)

    def forward(self, x, g1, g2, g3, sample=False, medimean=False):
        x = x.view(-1, self.dims[0])
        x = F.relu(self.l1.forward(x, g1, sample, medimean))
        x = F.relu(self.l2.forward(x, g2, sample, medimean))
        return F.log_softmax(self.l3.forward(x, g3, sample, medimean), dim=1)

    def log_prior(self):
        return self.l1.log_prior + self.l2.log_prior + self.l3.log_prior

    def log_variational_posterior(self):
        return (self.l1.log_variational_posterior + self.l2.log_variational_posterior
                + self.l3.log_variational_posterior)

    def sample_elbo(self, input, target, samples=SAMPLES, *, num_batches=None):
        """:285-319, same positional arguments.  NUM_BATCHES / SAMPLES are module globals there (:34-67) and here
        (``bnn_amd.base.NUM_BATCHES = 600``, ``SAMPLES = 1``); ``num_batches=`` overrides the former per call."""
        if num_batches is None:
            num_batches = NUM_BATCHES
        dev = input.device
        lps, lqs, nlls = [], [], []
        for _ in range(samples):
            gs = []
            for l in (self.l1, self.l2, self.l3):
                l.alpha = 1 / (1 + torch.exp(-l.lambdal))             # :292-297
                l.gamma.alpha = l.alpha
                gs.append(l.gamma.rsample().to(dev))                  # :300-302
            out = self.forward(input, gs[0], gs[1], gs[2], sample=True, medimean=False)
            lps.append(self.log_prior())
            lqs.append(self.log_variational_posterior())
            nlls.append(F.nll_loss(out, target, reduction="sum"))
        log_prior = torch.stack(lps).mean()
        log_q = torch.stack(lqs).mean()
        nll = torch.stack(nlls).mean()
        loss = nll + (log_q - log_prior) / num_batches                # :318
        return loss, log_prior, log_q, nll
