"""Differentiable PyTorch-ROCm restatements of the VECTOR-sized layer arithmetic, on GPU tensors.

What still enters here from ``autograd.Function.backward`` (never from a forward, never on CPU):
  * layers whose flows are 1-D chains (Radial / Householder / Sylvester / mixed): ``mnf_vector_graph`` over (I,) / (O,)
    vectors, with the noise the forward kernels drew re-created by ``lbbnn_philox_normal``;
  * the LRT layer's two bias terms (``lrt_vector_graph``);
  * RNVP / MNF-type layers only when ``LBBNN_DENSE_TORCH_BWD=1`` asks for it (A/B timing, second opinion in tests).
Everything (O,I)- or (B,O)-sized -- and, for planar and dense-flow MNF layers, every vector-sized gradient too -- is
computed by the HIP kernels (``layers._BayesLinearFn.backward``).
"""
import math

import torch


def _sigma(rho):
    return torch.log1p(torch.exp(rho))


def _alpha(lam):
    return 1 / (1 + torch.exp(-lam))


def _kl_bias(bias_mu, bias_rho, pr):
    sb = _sigma(bias_rho)
    return (torch.log(pr.bias_sigma_prior / sb) - 0.5
            + (sb ** 2 + (bias_mu - pr.bias_mu_prior) ** 2) / (2 * pr.bias_sigma_prior ** 2)).sum()


def _kl_weight(mu_eff, sigma, alpha, pr):
    return (alpha * (torch.log(pr.sigma_prior / sigma) - 0.5 + torch.log(alpha / pr.alpha_prior)
                     + (sigma ** 2 + (mu_eff - pr.mu_prior) ** 2) / (2 * pr.sigma_prior ** 2))
            + (1 - alpha) * torch.log((1 - alpha) / (1 - pr.alpha_prior))).sum()


def _planar(z, tr):
    logdet = z.new_zeros(())
    for (u, w, b) in tr:
        inner = torch.dot(w, z) + b[0]
        th = torch.tanh(inner)
        z = z + u * th
        logdet = logdet + torch.log(torch.abs(1 + (1 - th ** 2) * torch.dot(u, w)))
    return z, logdet


def _leaky(x, a=0.1):
    return torch.where(x >= 0, x, a * x)


def _dense(z, kind, tr, masks):
    """RNVP (flows2.py:206-219) / MNF (flows2.py:233-241) flow on a 1-D z.
    tr: list of parameter dicts keyed like the module's state_dict; masks: list of (I,) tensors."""
    logdet = z.new_zeros(())
    for t, m in zip(tr, masks):
        if kind == "RNVP":
            y = m * z
            for li, idx in enumerate((0, 2, 4, 6)):
                y = t["network.%d.weight" % idx] @ y + t["network.%d.bias" % idx]
                if li < 3:
                    y = _leaky(y)
            shift = t["t.weight"] @ y + t["t.bias"]
            gate = torch.sigmoid(t["s.weight"] @ y + t["s.bias"])
            z = ((1 - m) * z) * gate + (1 - gate) * shift + m * z
            logdet = logdet + ((1 - m) * gate.log()).sum()
        else:
            h = torch.tanh(t["f.weight"] @ (m * z) + t["f.bias"])
            mu = t["g.weight"] @ h + t["g.bias"]
            sig = torch.sigmoid(t["k.weight"] @ h + t["k.bias"])
            z = m * z + (1 - m) * (z * sig + (1 - sig) * mu)
            logdet = logdet + ((1 - m) * sig.log()).sum()
    return z, logdet


def _vector(z, tr):
    """Radial / Householder / Sylvester / planar steps on a 1-D z (flows2.py:48-135), typed by parameter names."""
    logdet = z.new_zeros(())
    for t in tr:
        if "v" in t:
            v = t["v"]
            z = z - 2 * v * torch.dot(v, z) / (v ** 2).sum()
        elif "z_0" in t:
            alpha = torch.nn.functional.softplus(t["log_alpha"])
            r = torch.sqrt(((z - t["z_0"]) ** 2).sum())
            H1 = t["beta"] / (alpha + r)
            H2 = -t["beta"] * r * (alpha + r) ** (-2)
            logdet = logdet + ((z.shape[0] - 1) * torch.log(1 + H1) + torch.log(1 + H1 + H2)).sum()
            z = z + H1 + H2
        elif "A" in t:
            lin = t["B"] @ z + t["b"]
            h = torch.tanh(lin)
            z = z + t["A"] @ h
            mat = torch.eye(lin.shape[0], dtype=z.dtype, device=z.device) + torch.diag(1 - h ** 2) @ (t["B"] @ t["A"])
            logdet = logdet + torch.log(torch.det(mat))
        else:
            z, ld = _planar(z, [(t["u"], t["w"], t["bias"])])
            logdet = logdet + ld
    return z, logdet


# ------------------------------------------------------------------------------------------------------------------
# Dense coupling flows: forward and backward as ONE HIP launch each (lbbnn_flow_dense_apply[_backward]) instead of the
# ~75 torch ops per application of the formulas in _dense above.
_RNVP_NAMES = ("network.0.weight", "network.0.bias", "network.2.weight", "network.2.bias", "network.4.weight",
               "network.4.bias", "network.6.weight", "network.6.bias", "t.weight", "t.bias", "s.weight", "s.bias")
_MNF_NAMES = ("f.weight", "f.bias", "g.weight", "g.bias", "k.weight", "k.bias")


def _dense_descs(kind, params, masks, grads=None):
    """ctypes arrays of lbbnn_dense_transform_t (and lbbnn_dense_grad_t) from the flat parameter list."""
    import ctypes
    from . import _lib
    names = _RNVP_NAMES if kind == "RNVP" else _MNF_NAMES
    T = len(masks)
    arr = (_lib.DenseTransform * max(T, 1))()
    garr = (_lib.DenseGrad * max(T, 1))() if grads is not None else None
    n = len(names)
    for t in range(T):
        p = dict(zip(names, params[t * n:(t + 1) * n]))
        d = arr[t]
        if kind == "RNVP":
            d.kind, d.hidden = 0, p["network.0.weight"].shape[0]
            d.w_in, d.b_in = p["network.0.weight"].data_ptr(), p["network.0.bias"].data_ptr()
            for l, idx in enumerate((2, 4, 6)):
                d.w_mid[l], d.b_mid[l] = p["network.%d.weight" % idx].data_ptr(), p["network.%d.bias" % idx].data_ptr()
            d.w_a, d.b_a, d.w_b, d.b_b = (p["t.weight"].data_ptr(), p["t.bias"].data_ptr(), p["s.weight"].data_ptr(),
                                          p["s.bias"].data_ptr())
        else:
            d.kind, d.hidden = 1, p["f.weight"].shape[0]
            d.w_in, d.b_in = p["f.weight"].data_ptr(), p["f.bias"].data_ptr()
            d.w_a, d.b_a, d.w_b, d.b_b = (p["g.weight"].data_ptr(), p["g.bias"].data_ptr(), p["k.weight"].data_ptr(),
                                          p["k.bias"].data_ptr())
        d.mask_fwd = masks[t].data_ptr()
        if grads is not None:
            g = dict(zip(names, grads[t * n:(t + 1) * n]))
            gd = garr[t]
            if kind == "RNVP":
                gd.w_in, gd.b_in = g["network.0.weight"].data_ptr(), g["network.0.bias"].data_ptr()
                for l, idx in enumerate((2, 4, 6)):
                    gd.w_mid[l], gd.b_mid[l] = g["network.%d.weight" % idx].data_ptr(), g["network.%d.bias" % idx].data_ptr()
                gd.w_a, gd.b_a, gd.w_b, gd.b_b = (g["t.weight"].data_ptr(), g["t.bias"].data_ptr(), g["s.weight"].data_ptr(),
                                                  g["s.bias"].data_ptr())
            else:
                gd.w_in, gd.b_in = g["f.weight"].data_ptr(), g["f.bias"].data_ptr()
                gd.w_a, gd.b_a, gd.w_b, gd.b_b = (g["g.weight"].data_ptr(), g["g.bias"].data_ptr(), g["k.weight"].data_ptr(),
                                                  g["k.bias"].data_ptr())
    return arr, garr, T


class _DenseFlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, nmask, z, *rest):
        from . import _lib
        masks = [m.reshape(-1).contiguous().float() for m in rest[:nmask]]
        params = [p.contiguous() for p in rest[nmask:]]
        z = z.contiguous()
        arr, _, T = _dense_descs(kind, params, masks)
        I = z.shape[0]
        z_out = torch.empty_like(z)
        logdet = torch.empty((), dtype=torch.float32, device=z.device)
        stream = torch.cuda.current_stream(z.device).cuda_stream
        _lib.check(_lib.lib().lbbnn_flow_dense_apply(arr, T, 0, z.data_ptr(), I, z_out.data_ptr(), logdet.data_ptr(), stream),
                   "lbbnn_flow_dense_apply")
        ctx.kind, ctx.nmask = kind, nmask
        ctx.save_for_backward(z, *masks, *params)
        return z_out, logdet

    @staticmethod
    def backward(ctx, dz, dld):
        from . import _lib
        saved = ctx.saved_tensors
        z, masks, params = saved[0], list(saved[1:1 + ctx.nmask]), list(saved[1 + ctx.nmask:])
        grads = [torch.empty_like(p) for p in params]
        arr, garr, T = _dense_descs(ctx.kind, params, masks, grads)
        I = z.shape[0]
        if dz is None:
            dz = torch.zeros_like(z)
        dz = dz.contiguous()
        dz_in = torch.empty_like(z)
        work = torch.empty(_lib.lib().lbbnn_flow_dense_apply_workspace(I, T), dtype=torch.float32, device=z.device)
        stream = torch.cuda.current_stream(z.device).cuda_stream
        dldp = dld.contiguous().data_ptr() if dld is not None else None
        _lib.check(_lib.lib().lbbnn_flow_dense_apply_backward(arr, garr, T, 0, z.data_ptr(), dz.data_ptr(), dldp, I,
                                                              dz_in.data_ptr(), work.data_ptr(), stream),
                   "lbbnn_flow_dense_apply_backward")
        return (None, None, dz_in, *([None] * ctx.nmask), *grads)


def _dense_hip(z, kind, tr, masks):
    names = _RNVP_NAMES if kind == "RNVP" else _MNF_NAMES
    flat = [t[n] for t in tr for n in names]
    return _DenseFlowFn.apply(kind, len(masks), z, *masks, *flat)


import os as _os
# The single-workgroup HIP forward/backward of the coupling MLPs is parity-green but SLOWER than the torch formulas of
# _dense for the headline sizes (one CU is instruction-bound on 1200 rows x 75 columns: 0.36 ms per launch, graphed
# training step 8.8 ms against 5.7 ms), so it is opt-in until it is spread over workgroups.
_DENSE_HIP = _os.environ.get("LBBNN_DENSE_HIP_BWD", "0") == "1"


def _dense_pair(za, zb, kind, tr, masks_a, masks_b):
    """The same dense flow applied to TWO vectors (the forward draw and the KL draw of the z flow) as one batch of two
    rows: half the torch launches of two _dense calls.  Returns (za', zb', logdet_b)."""
    Z = torch.stack((za, zb))                                   # (2, I)
    ld = Z.new_zeros(())
    for t, ma, mb in zip(tr, masks_a, masks_b):
        M = torch.stack((ma.reshape(-1), mb.reshape(-1)))
        if kind == "RNVP":
            Y = M * Z
            for li, idx in enumerate((0, 2, 4, 6)):
                Y = Y @ t["network.%d.weight" % idx].T + t["network.%d.bias" % idx]
                if li < 3:
                    Y = _leaky(Y)
            shift = Y @ t["t.weight"].T + t["t.bias"]
            gate = torch.sigmoid(Y @ t["s.weight"].T + t["s.bias"])
            Z = ((1 - M) * Z) * gate + (1 - gate) * shift + M * Z
            ld = ld + ((1 - M[1]) * gate[1].log()).sum()
        else:
            H = torch.tanh((M * Z) @ t["f.weight"].T + t["f.bias"])
            mu = H @ t["g.weight"].T + t["g.bias"]
            sig = torch.sigmoid(H @ t["k.weight"].T + t["k.bias"])
            Z = M * Z + (1 - M) * (Z * sig + (1 - sig) * mu)
            ld = ld + ((1 - M[1]) * sig[1].log()).sum()
    return Z[0], Z[1], ld


def _flow(z, spec, masks):
    kind, tr = spec
    if kind == "Planar":
        return _planar(z, tr)
    if kind in ("Radial", "Householder", "Sylvester", "mixed"):
        return _vector(z, tr)
    if z.is_cuda and _DENSE_HIP and len(tr) <= 8:
        return _dense_hip(z, kind, tr, masks)
    return _dense(z, kind, tr, masks)


def lrt_vector_graph(P, *, stochastic, want_kl, priors):
    """Bias-only part of the LRT layer; the (O,I) chain is lbbnn_weight_pass_backward's."""
    return {"z_k": None, "z2": None, "bmean": P["bias_mu"],
            "bvar": _sigma(P["bias_rho"]) ** 2 if stochastic else None,
            "kl": _kl_bias(P["bias_mu"], P["bias_rho"], priors) if want_kl else None}


def mnf_vector_graph(P, zf, rf, noise, act_mu, act_var, *, stochastic, want_kl, priors):
    """Vector-sized part of the MNF layer (LBBNN-GP-MF-MNF.py:182-187, 199-233): flows, q/r log-densities and
    the bias terms, as functions of the vector parameters and of the auxiliary activations act_mu / act_var
    (leaves here: their dependence on the weights is differentiated by lbbnn_weight_pass_backward)."""
    q0_std = P["q0_log_var"].exp().sqrt()
    pair = want_kl and zf[0] in ("RNVP", "MNF") and not _DENSE_HIP
    if pair:                                                   # both draws through the coupling MLPs as one 2-row batch
        z0 = P["q0_mean"] + q0_std * noise["eps_z2"]
        z_k, z2, log_det_q = _dense_pair(P["q0_mean"] + q0_std * noise["eps_z"], z0, zf[0], zf[1], noise["zmask"], noise["zmask2"])
    else:
        z_k, _ = _flow(P["q0_mean"] + q0_std * noise["eps_z"], zf, noise.get("zmask"))
    g = {"z_k": z_k, "z2": None, "bmean": P["bias_mu"],
         "bvar": _sigma(P["bias_rho"]) ** 2 if stochastic else None, "kl": None}
    if want_kl:
        if not pair:
            z0 = P["q0_mean"] + q0_std * noise["eps_z2"]
            z2, log_det_q = _flow(z0, zf, noise.get("zmask2"))
        log_q0 = (-0.5 * math.log(math.pi) - 0.5 * P["q0_log_var"]
                  - 0.5 * ((z0 - P["q0_mean"]) ** 2 / P["q0_log_var"].exp())).sum()
        act = torch.tanh(act_mu + act_var.sqrt() * noise["eps_act"])
        m = act.mean()
        mean_r, log_var_r = P["r0_b1"] * m, P["r0_b2"] * m
        z_b, log_det_r = _flow(z2, rf, noise.get("rmask"))
        log_rb = (-0.5 * math.log(math.pi) - 0.5 * log_var_r
                  - 0.5 * ((z_b[-1] - mean_r) ** 2 / log_var_r.exp())).sum()
        g["z2"] = z2
        g["kl"] = _kl_bias(P["bias_mu"], P["bias_rho"], priors) + (-log_det_q + log_q0) - (log_det_r + log_rb)
    return g
