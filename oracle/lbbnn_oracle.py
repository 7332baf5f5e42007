"""CPU oracle for the Bayesian linear-layer hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch PyTorch-CPU restatement of the arithmetic of
LarsELund/Bayesian-Neural-Nets' ``BayesianLinear.forward`` family.  It is the
*checker* the HIP kernels are compared against and the ``cpu_baseline`` leg of
``bench.py``; it is never imported by the product package
(``bayesian-neural-nets_amd/``), which has no CPU fallback at all.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.

Parity pin: every function below is checked against golden vectors generated
in the build container from the reference's own classes (see
``tests/golden/make_golden.py``; fixtures in ``tests/golden/*.npz``), to 1e-6
relative.  The reference has no tests of its own, so those
fixtures are the pin.

Conventions
-----------
* All randomness is *injected*: functions take the N(0,1) / Bernoulli draws the
  reference would have made, in the reference's draw order.
* Parameters are plain dicts keyed by the reference's ``state_dict`` names
  (``weight_mu, weight_rho, lambdal, bias_mu, bias_rho, q0_mean, q0_log_var,
  r0_c, r0_b1, r0_b2`` ...).
* Everything is dtype-preserving: feed float64 tensors to get a float64
  "truth" for error-budget checks.
* File:line citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- priors
@dataclass
class Priors:
    """Prior constants of one layer.

    The reference stores them as full constant tensors
    (LBBNN-GP-MF-LRT.py:142-143,151,159-160; LBBNN-GP-MF-MNF.py:145-146,154,162-163);
    the sim-study copies use other constants (LBBNN-GP-MF-MNFsim_study.py:157-174),
    hence configurable scalars here.
    """
    mu_prior: float = 0.0
    sigma_prior: float = 1.0
    alpha_prior: float = 0.05
    bias_mu_prior: float = 0.0
    bias_sigma_prior: float = 1.0


# --------------------------------------------------------------------------- G1 / G2
def sigma_of(rho: Tensor) -> Tensor:
    """Gaussian.sigma = log1p(exp(rho))  (LBBNN-GP-MF-LRT.py:80-82)."""
    return torch.log1p(torch.exp(rho))


def alpha_of(lambdal: Tensor) -> Tensor:
    """alpha_q = 1/(1+exp(-lambdal))  (LBBNN-GP-MF-LRT.py:167)."""
    return 1 / (1 + torch.exp(-lambdal))


def gaussian_log_prob_iid(x: Tensor, mu: Tensor, rho: Tensor) -> Tensor:
    """Gaussian.log_prob_iid (LBBNN-GP-MF.py:94-97)."""
    s = sigma_of(rho)
    return -math.log(math.sqrt(2 * math.pi)) - torch.log(s) - ((x - mu) ** 2) / (2 * s ** 2)


def gaussian_log_prob(x: Tensor, mu: Tensor, rho: Tensor) -> Tensor:
    """Gaussian.log_prob (LBBNN-GP-MF.py:89-92)."""
    return gaussian_log_prob_iid(x, mu, rho).sum()


def gaussian_full_log_prob(x: Tensor, gamma: Tensor, mu: Tensor, rho: Tensor) -> Tensor:
    """Gaussian.full_log_prob (LBBNN-GP-MF.py:99-101)."""
    return torch.log(gamma * torch.exp(gaussian_log_prob_iid(x, mu, rho)) + (1 - gamma) + 1e-8).sum()


def bernoulli_log_prob(gamma: Tensor, alpha: Tensor, exact: bool) -> Tensor:
    """Bernoulli.log_prob (LBBNN-GP-MF.py:122-128)."""
    g = torch.round(gamma.detach()) if exact else gamma
    return (g * torch.log(alpha + 1e-8) + (1 - g) * torch.log(1 - alpha + 1e-8)).sum()


# --------------------------------------------------------------------------- KL pieces
def kl_bias_term(bias_mu: Tensor, bias_rho: Tensor, pr: Priors) -> Tensor:
    """kl_bias (LBBNN-GP-MF-LRT.py:185-186; LBBNN-GP-MF-MNF.py:227-228)."""
    sb = sigma_of(bias_rho)
    return (torch.log(pr.bias_sigma_prior / sb) - 0.5
            + (sb ** 2 + (bias_mu - pr.bias_mu_prior) ** 2) / (2 * pr.bias_sigma_prior ** 2)).sum()


def kl_weight_elem(mu_eff: Tensor, sigma: Tensor, alpha: Tensor, pr: Priors) -> Tensor:
    """Per-weight KL integrand (LBBNN-GP-MF-LRT.py:189-192).

    ``mu_eff`` is ``weight_mu`` for LRT and ``weight_mu * z2`` for MNF
    (LBBNN-GP-MF-MNF.py:230-233).
    """
    return (alpha * (torch.log(pr.sigma_prior / sigma) - 0.5 + torch.log(alpha / pr.alpha_prior)
                     + (sigma ** 2 + (mu_eff - pr.mu_prior) ** 2) / (2 * pr.sigma_prior ** 2))
            + (1 - alpha) * torch.log((1 - alpha) / (1 - pr.alpha_prior)))


# --------------------------------------------------------------------------- L2: LRT layer
def lrt_forward(x: Tensor, p: Dict[str, Tensor], eps: Optional[Tensor], *,
                stochastic: bool = True, compute_kl: bool = True,
                priors: Priors = Priors()) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """BayesianLinear.forward of the LRT script (LBBNN-GP-MF-LRT.py:166-197).

    ``stochastic`` = ``self.training or sample``; ``compute_kl`` =
    ``self.training or calculate_log_probs``.  ``eps`` is the (B,O) draw of :174.
    Returns (activations, kl, intermediates).
    """
    alpha = alpha_of(p["lambdal"])                                    # :167
    inter: Dict[str, Tensor] = {"alpha": alpha}
    e_w = p["weight_mu"] * alpha                                      # :170 / :178
    if stochastic:
        sigma = sigma_of(p["weight_rho"])
        var_w = sigma ** 2 * alpha ** 2                               # :171
        e_b = torch.mm(x, e_w.T) + p["bias_mu"]                       # :172
        var_b = torch.mm(x ** 2, var_w.T) + sigma_of(p["bias_rho"]) ** 2   # :173
        act = e_b + torch.sqrt(var_b) * eps                           # :175
        inter.update(e_w=e_w, var_w=var_w, e_b=e_b, var_b=var_b)
    else:
        act = torch.mm(x, e_w.T) + p["bias_mu"]                       # :178-180
        inter.update(e_w=e_w)
    if compute_kl:
        kl = lrt_kl(p, priors)
    else:
        kl = torch.zeros((), dtype=x.dtype)                           # :196 (int 0 there)
    return act, kl, inter


def lrt_kl(p: Dict[str, Tensor], priors: Priors = Priors()) -> Tensor:
    """Closed-form KL of the LRT layer (LBBNN-GP-MF-LRT.py:185-194)."""
    alpha = alpha_of(p["lambdal"])
    sigma = sigma_of(p["weight_rho"])
    return kl_bias_term(p["bias_mu"], p["bias_rho"], priors) + \
        kl_weight_elem(p["weight_mu"], sigma, alpha, priors).sum()


# --------------------------------------------------------------------------- F1: planar
def planar_step_1d(z: Tensor, u: Tensor, w: Tensor, b: Tensor) -> Tuple[Tensor, Tensor]:
    """One PlanarTransform on a 1-D z (flows2.py:86-95)."""
    inner = torch.dot(w, z) + b                                       # :87  (b has shape (1,))
    t = torch.tanh(inner)
    z_new = z + u * t                                                 # :88
    psi = (1 - t ** 2) * w                                            # :89
    logdet = torch.log(torch.abs(1 + torch.dot(u, psi)))             # :95  shape (1,)
    return z_new, logdet


# --------------------------------------------------------------------------- F4: radial / Householder / Sylvester
def radial_step_1d(z: Tensor, t: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """RadialTransform on a 1-D z (flows2.py:57-69): the norm runs over the whole vector (``dim=[]``),
    H1/H2 are (1,)-shaped and are ADDED to every element (as written)."""
    alpha = torch.nn.functional.softplus(t["log_alpha"])              # :58 nn.Softplus()
    diff = z - t["z_0"]
    r = torch.sqrt((diff ** 2).sum())                                 # :61
    H1 = t["beta"] / (alpha + r)                                      # :62
    H2 = -t["beta"] * r * (alpha + r) ** (-2)                         # :63
    d = z.shape[-1]
    logdet = (d - 1) * torch.log(1 + H1) + torch.log(1 + H1 + H2)     # :68
    return z + H1 + H2, logdet


def householder_step_1d(z: Tensor, t: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """HouseholderTransform (flows2.py:128-135); log_det is the integer 0."""
    v = t["v"]
    return z - 2 * v * torch.dot(v, z) / (v ** 2).sum(), z.new_zeros(1)


def sylvester_step_1d(z: Tensor, t: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """SylvesterTransform, M = 5 (flows2.py:112-120): z + A tanh(Bz + b); log det(I + diag(h'(Bz+b)) B A)."""
    lin = t["B"] @ z + t["b"]
    z_new = z + t["A"] @ torch.tanh(lin)
    M = t["b"].shape[0]
    mat = torch.eye(M, dtype=z.dtype) + torch.diag(1 - torch.tanh(lin) ** 2) @ (t["B"] @ t["A"])
    return z_new, torch.log(torch.det(mat)).reshape(1)


def vector_step_1d(z: Tensor, t: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """One 1-D transform, type read off the parameter names (so 'mixed' chains work: flows2.py:31-37)."""
    if "v" in t:
        return householder_step_1d(z, t)
    if "z_0" in t:
        return radial_step_1d(z, t)
    if "A" in t:
        return sylvester_step_1d(z, t)
    return planar_step_1d(z, t["u"], t["w"], t["bias"])


def vector_flow_1d(z: Tensor, tr: Sequence[Dict[str, Tensor]]) -> Tuple[Tensor, Tensor]:
    logdet = z.new_zeros(1)
    for t in tr:
        z, ld = vector_step_1d(z, t)
        logdet = logdet + ld
    return z, logdet


def planar_flow_1d(z: Tensor, tr: Sequence[Dict[str, Tensor]]) -> Tuple[Tensor, Tensor]:
    """PropagateFlow('Planar').forward on 1-D z (flows2.py:41-46). logdet has shape (1,)."""
    logdet = 0
    for t in tr:
        z, ld = planar_step_1d(z, t["u"], t["w"], t["bias"])
        logdet = logdet + ld
    return z, logdet


def planar_flow_rows(z: Tensor, tr: Sequence[Dict[str, Tensor]]) -> Tuple[Tensor, Tensor]:
    """Row-wise planar flow on (R,I) z -- SURVEY.md 8(a) row F1 restatement.

    The reference's ``torch.dot`` rejects 2-D z (flows2.py:87), so for ``z_flow`` the
    contract is the per-row application of the 1-D transform; returns z (R,I) and
    logdet (R,1)->(R,) after squeeze by the caller.
    """
    logdet = torch.zeros(z.shape[0], 1, dtype=z.dtype)
    for t in tr:
        inner = z @ t["w"] + t["bias"]                                # (R,)
        th = torch.tanh(inner)
        z = z + th[:, None] * t["u"][None, :]
        psi_dot_u = (1 - th ** 2) * torch.dot(t["u"], t["w"])
        # same value as dot(u, (1-th^2)*w) up to rounding
        logdet = logdet + torch.log(torch.abs(1 + psi_dot_u))[:, None]
    return z, logdet


# --------------------------------------------------------------------------- F2: RNVP
def _leaky(x: Tensor, a: float = 0.1) -> Tensor:
    return torch.where(x >= 0, x, a * x)


def rnvp_step(z: Tensor, t: Dict[str, Tensor], mask: Tensor) -> Tuple[Tensor, Tensor]:
    """One RNVP transform (flows2.py:206-219). ``mask`` is the Bernoulli(0.5) draw of :209.

    Parameter names follow the reference state_dict: ``network.{0,2,4,6}.{weight,bias}``,
    ``t.{weight,bias}``, ``s.{weight,bias}`` (MLP = 4 Linear layers with LeakyReLU(0.1)
    between, last activation dropped, flows2.py:176-185,198-202).
    """
    z1, z2 = (1 - mask) * z, mask * z                                 # :211
    y = z2
    for li, idx in enumerate((0, 2, 4, 6)):
        y = y @ t[f"network.{idx}.weight"].T + t[f"network.{idx}.bias"]
        if li < 3:
            y = _leaky(y)
    shift = y @ t["t.weight"].T + t["t.bias"]                         # :213
    scale = y @ t["s.weight"].T + t["s.bias"]
    gate = torch.sigmoid(scale)                                       # :214
    x = (z1 * gate + (1 - gate) * shift) + z2                         # :215
    logdet = ((1 - mask) * gate.log()).sum(-1)                        # :219
    return x, logdet


def rnvp_flow(z: Tensor, tr: Sequence[Dict[str, Tensor]], masks: Sequence[Tensor]):
    logdet = 0
    for t, m in zip(tr, masks):
        z, ld = rnvp_step(z, t, m)
        logdet = logdet + ld
    return z, logdet


# --------------------------------------------------------------------------- F3: MNF flow
def mnfflow_step(z: Tensor, t: Dict[str, Tensor], mask: Tensor) -> Tuple[Tensor, Tensor]:
    """One MNF transform (flows2.py:233-241). Params ``f,g,k .{weight,bias}``."""
    h = torch.tanh((mask * z) @ t["f.weight"].T + t["f.bias"])        # :235
    mu = h @ t["g.weight"].T + t["g.bias"]                            # :236
    sig = torch.sigmoid(h @ t["k.weight"].T + t["k.bias"])            # :237
    znew = mask * z + (1 - mask) * (z * sig + (1 - sig) * mu)         # :238
    logdet = ((1 - mask) * sig.log()).sum()                           # :241 (sum over ALL elems)
    return znew, logdet


def mnfflow_flow(z: Tensor, tr: Sequence[Dict[str, Tensor]], masks: Sequence[Tensor]):
    logdet = 0
    for t, m in zip(tr, masks):
        z, ld = mnfflow_step(z, t, m)
        logdet = logdet + ld
    return z, logdet


# --------------------------------------------------------------------------- flow dispatch
@dataclass
class Flow:
    """A PropagateFlow (flows2.py:14-46): type name + per-transform parameter dicts."""
    kind: str                                   # 'Planar' | 'RNVP' | 'MNF' | 'Radial' | 'Householder' | 'Sylvester' | 'mixed'
    transforms: List[Dict[str, Tensor]] = field(default_factory=list)

    def run(self, z: Tensor, masks: Optional[Sequence[Tensor]] = None):
        if self.kind == "Planar":
            if z.dim() == 1:
                return planar_flow_1d(z, self.transforms)
            return planar_flow_rows(z, self.transforms)
        if self.kind == "RNVP":
            return rnvp_flow(z, self.transforms, masks)
        if self.kind == "MNF":
            return mnfflow_flow(z, self.transforms, masks)
        if self.kind in ("Radial", "Householder", "Sylvester", "mixed"):
            if z.dim() == 1:
                return vector_flow_1d(z, self.transforms)
            # row-wise restatement: the 1-D transform applied to every row
            outs = [vector_flow_1d(row, self.transforms) for row in z]
            return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])
        raise ValueError(f"flow kind {self.kind!r} not in the oracle")


def flow_from_state(prefix: str, kind: str, state: Dict[str, Tensor], T: int) -> Flow:
    """Collect ``<prefix>.transforms.<n>.<name>`` entries of a state_dict into a Flow."""
    trs = []
    for n in range(T):
        pre = f"{prefix}.transforms.{n}."
        trs.append({k[len(pre):]: v for k, v in state.items() if k.startswith(pre)})
    return Flow(kind, trs)


# --------------------------------------------------------------------------- M2/M3: MNF layer
def mnf_sample_z(p: Dict[str, Tensor], eps_z: Tensor, z_flow: Flow,
                 masks: Optional[Sequence[Tensor]] = None):
    """BayesianLinear.sample_z (LBBNN-GP-MF-MNF.py:182-187).

    ``eps_z`` is (R,I) (R = batch_size).  Returns (zs[-1] = LAST ROW, logdet.squeeze(), z0).
    """
    q0_std = p["q0_log_var"].exp().sqrt()                             # :183 (repeat == broadcast)
    z0 = p["q0_mean"] + q0_std * eps_z                                # :185
    zs, logdet = z_flow.run(z0, masks)                                # :186
    if not torch.is_tensor(logdet):
        logdet = torch.tensor(float(logdet), dtype=z0.dtype)
    return zs[-1], logdet.squeeze(), z0                               # :187


def mnf_forward(x: Tensor, p: Dict[str, Tensor], z_flow: Flow, r_flow: Flow,
                noise: Dict[str, Tensor], *, stochastic: bool = True, compute_kl: bool = True,
                priors: Priors = Priors()):
    """BayesianLinear.forward of the MNF script (LBBNN-GP-MF-MNF.py:190-239).

    noise keys (reference draw order):
      eps_z   (B,I)  randn_like of :184, first sample_z (:194 / :203)
      zmask   list of T (B,I) Bernoulli masks (RNVP/MNF flow types only)
      eps_out (B,O)  randn of :199 (stochastic only)
      eps_z2  (1,I)  second sample_z (:210)              (compute_kl only)
      zmask2  list of T (1,I) masks
      eps_act (O,)   randn_like of :218
      rmask   list of T (I,) masks for r_flow(:222)
    """
    alpha = alpha_of(p["lambdal"])                                    # :191
    inter: Dict[str, Tensor] = {"alpha": alpha}
    z_k, _, _ = mnf_sample_z(p, noise["eps_z"], z_flow, noise.get("zmask"))   # :194 / :203
    e_w = p["weight_mu"] * alpha                                      # :195 / :204
    inter.update(z_k=z_k, e_w=e_w)
    if stochastic:
        sigma = sigma_of(p["weight_rho"])
        var_w = sigma ** 2 * alpha ** 2                               # :196
        e_b = torch.mm(x * z_k, e_w.T) + p["bias_mu"]                 # :197
        var_b = torch.mm(x ** 2, var_w.T) + sigma_of(p["bias_rho"]) ** 2   # :198
        act = e_b + torch.sqrt(var_b) * noise["eps_out"]              # :199-200
        inter.update(var_w=var_w, e_b=e_b, var_b=var_b)
    else:
        act = torch.mm(x * z_k, e_w.T) + p["bias_mu"]                 # :203-206

    if not compute_kl:
        return act, torch.zeros((), dtype=x.dtype), inter             # :237

    z2, log_det_q, z0 = mnf_sample_z(p, noise["eps_z2"], z_flow, noise.get("zmask2"))  # :210
    sigma = sigma_of(p["weight_rho"])
    W_mean = z2 * p["weight_mu"] * alpha                              # :211
    W_var = sigma ** 2 * alpha ** 2                                   # :212
    log_q0 = (-0.5 * math.log(math.pi) - 0.5 * p["q0_log_var"]
              - 0.5 * ((z0 - p["q0_mean"]) ** 2 / p["q0_log_var"].exp())).sum()   # :213-214 (log pi!)
    log_q = -log_det_q + log_q0                                       # :215
    act_mu = p["r0_c"] @ W_mean.T                                     # :216
    act_var = p["r0_c"] ** 2 @ W_var.T                                # :217
    act_inner = act_mu + act_var.sqrt() * noise["eps_act"]            # :218
    a = torch.tanh(act_inner)                                         # :219
    mean_r = p["r0_b1"].outer(a).mean(-1)                             # :220
    log_var_r = p["r0_b2"].outer(a).mean(-1)                          # :221
    z_b, log_det_r = r_flow.run(z2, noise.get("rmask"))               # :222  (z2 is 1-D)
    if not torch.is_tensor(log_det_r):
        log_det_r = torch.tensor(float(log_det_r), dtype=x.dtype)
    log_rb = (-0.5 * math.log(math.pi) - 0.5 * log_var_r
              - 0.5 * ((z_b[-1] - mean_r) ** 2 / log_var_r.exp())).sum()   # :223-224 (z_b[-1] scalar!)
    log_r = log_det_r + log_rb                                        # :225
    kl_b = kl_bias_term(p["bias_mu"], p["bias_rho"], priors)          # :227-228
    kl_w = kl_weight_elem(p["weight_mu"] * z2, sigma, alpha, priors).sum()   # :230-233
    kl = kl_b + kl_w + log_q - log_r                                  # :235
    inter.update(z2=z2, z0_kl=z0, log_det_q=log_det_q, log_q0=log_q0, act_mu=act_mu,
                 act_var=act_var, z_b=z_b, log_det_r=log_det_r, log_rb=log_rb,
                 kl_bias=kl_b, kl_weight=kl_w, mean_r=mean_r, log_var_r=log_var_r)
    return act, kl, inter


# --------------------------------------------------------------------------- N1: networks
def _net_tail(h: Tensor) -> Tensor:
    return torch.log_softmax(h, dim=1)


def lrt_network_forward(x: Tensor, layers: Sequence[Dict[str, Tensor]], eps: Sequence[Tensor], *,
                        stochastic: bool = True, compute_kl: bool = True,
                        priors: Priors = Priors()):
    """BayesianNetwork.forward + kl() (LBBNN-GP-MF-LRT.py:206-214): ReLU, ReLU, log_softmax."""
    h = x.reshape(x.shape[0], -1)
    kl = torch.zeros((), dtype=x.dtype)
    for i, p in enumerate(layers):
        h, k, _ = lrt_forward(h, p, eps[i] if stochastic else None, stochastic=stochastic,
                              compute_kl=compute_kl, priors=priors)
        kl = kl + k
        if i < len(layers) - 1:
            h = torch.relu(h)
    return _net_tail(h), kl


def mnf_network_forward(x: Tensor, layers: Sequence[Dict[str, Tensor]],
                        z_flows: Sequence[Flow], r_flows: Sequence[Flow],
                        noise: Sequence[Dict[str, Tensor]], *, stochastic: bool = True,
                        compute_kl: bool = True, priors: Priors = Priors()):
    """BayesianNetwork.forward + kl() (LBBNN-GP-MF-MNF.py:252-260)."""
    h = x.reshape(x.shape[0], -1)
    kl = torch.zeros((), dtype=x.dtype)
    for i, p in enumerate(layers):
        h, k, _ = mnf_forward(h, p, z_flows[i], r_flows[i], noise[i], stochastic=stochastic,
                              compute_kl=compute_kl, priors=priors)
        kl = kl + k
        if i < len(layers) - 1:
            h = torch.relu(h)
    return _net_tail(h), kl


# --------------------------------------------------------------------------- B1/B2: base LBBNN
def gaussgamma_log_prob(x: Tensor, gamma: Tensor, a: Tensor, b: Tensor, tau: Tensor,
                        exact: bool) -> Tensor:
    """GaussGamma.log_prob (LBBNN-GP-MF.py:140-151); ``tau`` = the Gamma(a,b).rsample() of :141."""
    g = torch.round(gamma.detach()) if exact else gamma
    return (g * (a * torch.log(b) + (a - 0.5) * tau - b * tau - torch.lgamma(a)
                 - 0.5 * math.log(2 * math.pi)) - tau * torch.pow(x, 2) + (1 - g) + 1e-8).sum()


def betabinomial_log_prob(gamma: Tensor, pa: Tensor, pb: Tensor, exact: bool) -> Tensor:
    """BetaBinomial.log_prob (LBBNN-GP-MF.py:162-173): nine lgamma terms per element."""
    g = torch.round(gamma.detach()) if exact else gamma
    one = torch.ones_like(gamma)
    return (torch.lgamma(one) + torch.lgamma(g + one * pa)
            + torch.lgamma(one * (1 + pb) - g) + torch.lgamma(one * (pa + pb))
            - torch.lgamma(one * pa + g)
            - torch.lgamma(one * 2 - g) - torch.lgamma(one * (1 + pa + pb))
            - torch.lgamma(one * pa) - torch.lgamma(one * pb)).sum()


def base_forward(x: Tensor, p: Dict[str, Tensor], cgamma: Tensor, noise: Dict[str, Tensor], *,
                 mode: str = "sample", compute_lp: bool = True, alpha_attr: Optional[Tensor] = None,
                 gamma_alpha: Optional[Tensor] = None, exact: Dict[str, bool] = None):
    """BayesianLinear.forward of the baseline LBBNN (LBBNN-GP-MF.py:228-255).

    mode: 'sample' (training or sample, :230-234), 'medimean' (:236-238), 'mean' (:240-242).
    noise: eps_w (O,I), eps_b (O,) [sample mode]; tau_w (1,), tau_b (O,) [compute_lp].
    ``alpha_attr`` is the value of ``self.alpha`` *before* this call (used by the 'mean'
    branch, which reads it before :246 refreshes it).
    ``gamma_alpha`` is ``self.gamma.alpha`` (the Bernoulli object's own alpha): ``sample_elbo``
    sets it to sigmoid(lambdal) before every call (:292-297), which is the default here.
    ``exact`` = {'weight_prior','bias_prior','gamma_prior','gamma'} -> bool (all False in training).
    Returns (out, log_prior, log_variational_posterior).
    """
    ex = dict(weight_prior=False, bias_prior=False, gamma_prior=False, gamma=False)
    if exact:
        ex.update(exact)
    if mode == "sample":
        ws = p["weight_mu"] + sigma_of(p["weight_rho"]) * noise["eps_w"]     # :232 (Gaussian.rsample :85-87)
        weight = cgamma * ws                                                 # :233
        bias = p["bias_mu"] + sigma_of(p["bias_rho"]) * noise["eps_b"]       # :234
    elif mode == "medimean":
        weight = cgamma * p["weight_mu"]                                     # :237
        bias = p["bias_mu"]
    else:
        weight = alpha_attr * p["weight_mu"]                                 # :241
        bias = p["bias_mu"]
    if compute_lp:
        alpha = alpha_of(p["lambdal"]) if gamma_alpha is None else gamma_alpha   # :246 / :293
        log_prior = (gaussgamma_log_prob(weight, cgamma, p["weight_a"], p["weight_b"],
                                         noise["tau_w"], ex["weight_prior"])
                     + gaussgamma_log_prob(bias, torch.ones_like(bias), p["bias_a"], p["bias_b"],
                                           noise["tau_b"], ex["bias_prior"])
                     + betabinomial_log_prob(cgamma, p["pa"], p["pb"], ex["gamma_prior"]))   # :247-249
        log_q = (gaussian_full_log_prob(weight, cgamma, p["weight_mu"], p["weight_rho"])
                 + bernoulli_log_prob(cgamma, alpha, ex["gamma"])
                 + gaussian_log_prob(bias, p["bias_mu"], p["bias_rho"]))                     # :250-251
    else:
        log_prior = torch.zeros((), dtype=x.dtype)
        log_q = torch.zeros((), dtype=x.dtype)
    out = x @ weight.T + bias                                                # :255 F.linear
    return out, log_prior, log_q


# --------------------------------------------------------------------------- V1/V2: variational dropout
def vd_forward(x: Tensor, theta: Tensor, alpha: Tensor, zeta: Tensor) -> Tensor:
    """BayesianLayer.forward (variational_dropout.py:63-68). theta is (I,O) (NN layout)."""
    phi = torch.matmul(x, theta)                                      # :64
    delta = torch.matmul(x ** 2, theta ** 2) * alpha                  # :65
    return phi + torch.sqrt(delta) * zeta                             # :67


def vd_kl(alphas: Sequence[Tensor]) -> Tensor:
    """KL polynomial of loss_fn (variational_dropout.py:97-102)."""
    c1, c2, c3 = 1.16145124, -1.50204118, 0.58629921
    kl = 0
    for a in alphas:
        kl = kl + (0.5 * torch.log(a) + c1 * a + c2 * a ** 2 + c3 * a ** 3).sum()
    return kl


# --------------------------------------------------------------------------- init helpers
def init_lrt_params(I: int, O: int, gen: torch.Generator, mu_range: float = 0.2) -> Dict[str, Tensor]:
    """Parameter init distributions of BayesianLinear.__init__ (LBBNN-GP-MF-LRT.py:137-155).

    (Own generator => NOT the reference's seeded values; used for synthetic benchmarks.)
    """
    def U(shape, lo, hi):
        return torch.empty(shape).uniform_(lo, hi, generator=gen)
    return {
        "weight_mu": U((O, I), -mu_range, mu_range),
        "weight_rho": U((O, I), -5, -4),
        "lambdal": U((O, I), 0, 1),
        "bias_mu": U((O,), -0.2, 0.2),
        "bias_rho": U((O,), -5, -4),
    }


def init_mnf_params(I: int, O: int, gen: torch.Generator) -> Dict[str, Tensor]:
    """LBBNN-GP-MF-MNF.py:140-172 init distributions (weight_mu ~ U(+-0.01))."""
    p = init_lrt_params(I, O, gen, mu_range=0.01)
    N = lambda n: torch.randn(n, generator=gen)
    p["q0_mean"] = 0.1 * N(I)
    p["q0_log_var"] = -9 + 0.1 * N(I)
    p["r0_c"] = 0.1 * N(I)
    p["r0_b1"] = 0.1 * N(I)
    p["r0_b2"] = 0.1 * N(I)
    return p


def init_planar_flow(I: int, T: int, gen: torch.Generator) -> Flow:
    """PlanarTransform.__init__ (flows2.py:76-78): u,w ~ U(-0.01,0.01) (I,), bias (1,)."""
    trs = []
    for _ in range(T):
        trs.append({
            "u": torch.empty(I).uniform_(-0.01, 0.01, generator=gen),
            "w": torch.empty(I).uniform_(-0.01, 0.01, generator=gen),
            "bias": torch.empty(1).uniform_(-0.01, 0.01, generator=gen),
        })
    return Flow("Planar", trs)
