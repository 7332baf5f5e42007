"""Normalizing-flow parameter containers with the reference's module/parameter names
(flows2.py:14-241), so ``state_dict`` keys and optimizer parameter groups carry over:
``z_flow.transforms.N.{u,w,bias}`` (planar), ``….network.{0,2,4,6}.*, t.*, s.*`` (RNVP),
``….{f,g,k}.*`` (MNF).

Inside the MNF layer the flows run fused with the layer's other vector work (``lbbnn_mnf_flow_planar``,
``lbbnn_mnf_flow_dense``, ``lbbnn_flow_chain``); stand-alone, ``PropagateFlow.forward(z)`` and each transform's
``forward(z)`` / ``log_det()`` (flows2.py:41-46) run the same arithmetic through ``lbbnn_flow_chain[_rows]`` and
``lbbnn_flow_dense_rows`` (MFMA).  Initialisation reproduces the reference's formulas and draw order, so a seeded
construction yields the reference's values.
"""
import torch
import torch.nn as nn

from . import ops


def parameter_init(low, high, size):
    # (low - high) * U[0,1) + high, as flows2.py:10-12
    return (low - high) * torch.rand(size) + high


class _Transform(nn.Module):
    """``z' = f(z)`` then ``f.log_det()`` -- the reference's two-call protocol (flows2.py:41-46: every transform caches
    what its log_det needs on ``self`` during forward).  Here forward runs the transform as a one-step flow through the
    same kernels as PropagateFlow.forward and caches the log-det it returned."""
    _kind = None
    _logdet = None

    def _as_flow(self):
        f = PropagateFlow.__new__(PropagateFlow)
        nn.Module.__init__(f)
        f.kind, f.dim = self._kind, self._dim()
        f.__dict__["_modules"]["transforms"] = nn.ModuleList([self])
        return f

    def _dim(self):
        raise NotImplementedError

    def forward(self, z):
        f = self._as_flow()
        f.masks = [self.mask_in] if getattr(self, "mask_in", None) is not None else None
        f.keep_masks = self._kind in ("RNVP", "MNF")
        out, self._logdet = f(z)
        if f.keep_masks and f.last_masks is not None:
            m = f.last_masks[0].reshape(z.shape)
            if self._kind == "RNVP":
                self.mask = m                      # flows2.py:209
            else:
                self.m = m                         # flows2.py:234
        return out

    def log_det(self):
        if self._logdet is None:
            raise RuntimeError("bnn_amd: log_det() before forward() (the reference reads attributes forward() sets)")
        return self._logdet


class PlanarTransform(_Transform):
    """Parameters of one planar transform (flows2.py:72-95): u, w (dim,), bias (1,)."""
    _kind = "Planar"

    def _dim(self):
        return self.u.shape[0]

    def __init__(self, dim):
        super().__init__()
        self.u = nn.Parameter(parameter_init(-0.01, 0.01, dim))
        self.w = nn.Parameter(parameter_init(-0.01, 0.01, dim))
        self.bias = nn.Parameter(parameter_init(-0.01, 0.01, 1))


class _MLP(nn.Sequential):
    def __init__(self, *layer_sizes, leaky_a=0.1):
        layers = []
        for s1, s2 in zip(layer_sizes, layer_sizes[1:]):
            layers.append(nn.Linear(s1, s2))
            layers.append(nn.LeakyReLU(leaky_a))
        super().__init__(*layers[:-1])


class RNVP(_Transform):
    """Parameters of one RNVP coupling transform (flows2.py:188-219).  ``mask_in``: explicit Bernoulli mask for the next
    forward (parity tests); after forward ``mask`` holds the mask used (flows2.py:209)."""
    _kind = "RNVP"
    mask_in = None

    def _dim(self):
        return self.t.out_features

    def __init__(self, dim, h_sizes=(75, 75, 75, 75)):
        super().__init__()
        self.network = _MLP(*([dim] + list(h_sizes)))
        self.t = nn.Linear(h_sizes[-1], dim)
        self.s = nn.Linear(h_sizes[-1], dim)


class MNF(_Transform):
    """Parameters of one MNF-type transform (flows2.py:225-241); ``mask_in`` / ``m`` as RNVP's ``mask_in`` / ``mask``."""
    _kind = "MNF"
    mask_in = None

    def _dim(self):
        return self.g.out_features

    def __init__(self, dim, hidden=100):
        super().__init__()
        self.f = nn.Linear(dim, hidden)
        self.g = nn.Linear(hidden, dim)
        self.k = nn.Linear(hidden, dim)


class RadialTransform(_Transform):
    """Parameters of one radial transform (flows2.py:48-69): z_0 (dim,), log_alpha (1,), beta (1,)."""
    _kind = "Radial"

    def _dim(self):
        return self.z_0.shape[0]

    def __init__(self, dim):
        super().__init__()
        self.z_0 = nn.Parameter(parameter_init(-0.1, 0.1, dim))
        self.log_alpha = nn.Parameter(parameter_init(-4, 5, 1))
        self.beta = nn.Parameter(parameter_init(-0.1, 0.1, 1))
        self.d = dim


class SylvesterTransform(_Transform):
    """Parameters of one Sylvester transform, M = 5 (flows2.py:98-120): A (dim,M), B (M,dim), b (M,)."""
    _kind = "Sylvester"

    def _dim(self):
        return self.A.shape[0]

    def __init__(self, dim):
        super().__init__()
        self.M = 5
        self.A = nn.Parameter(parameter_init(-0.01, 0.01, (dim, self.M)))
        self.B = nn.Parameter(parameter_init(-0.01, 0.01, (self.M, dim)))
        self.b = nn.Parameter(parameter_init(-0.01, 0.01, self.M))


class HouseholderTransform(_Transform):
    """Parameters of one Householder transform (flows2.py:122-135): v (dim,).  log_det() is 0 (:135)."""
    _kind = "Householder"

    def _dim(self):
        return self.v.shape[0]

    def log_det(self):
        return 0

    def __init__(self, dim):
        super().__init__()
        self.v = nn.Parameter(parameter_init(-0.01, 0.01, dim))


_KINDS = {"Planar": PlanarTransform, "RNVP": RNVP, "MNF": MNF, "Radial": RadialTransform,
          "Sylvester": SylvesterTransform, "Householder": HouseholderTransform}
VECTOR_KINDS = ("Planar", "Radial", "Sylvester", "Householder", "mixed")     # 1-D flows: lbbnn_flow_chain


class PropagateFlow(nn.Module):
    """flows2.PropagateFlow(transform, dim, num_transforms): same ctor, ``forward(z) -> (z, logdet)``."""

    def __init__(self, transform, dim, num_transforms):
        super().__init__()
        if transform != "mixed" and transform not in _KINDS:
            raise NotImplementedError("flow type %r (the reference only prints 'Transform not implemented', "
                                      "flows2.py:39-40); known: %s" % (transform, sorted(_KINDS) + ["mixed"]))
        self.kind = transform
        self.dim = dim
        if transform == "mixed":               # 5 x (Householder, Planar), num_transforms ignored (flows2.py:31-37)
            self.transforms = nn.ModuleList([cls(dim) for _ in range(5) for cls in (HouseholderTransform, PlanarTransform)])
        else:
            self.transforms = nn.ModuleList([_KINDS[transform](dim) for _ in range(num_transforms)])

    def chain_steps(self):
        """(type, M, p0, p1, p2) per transform for lbbnn_flow_chain (1-D flow kinds only)."""
        steps = []
        for t in self.transforms:
            if isinstance(t, PlanarTransform):
                steps.append((ops.FLOW_PLANAR, 0, t.u, t.w, t.bias))
            elif isinstance(t, RadialTransform):
                steps.append((ops.FLOW_RADIAL, 0, t.z_0, t.log_alpha, t.beta))
            elif isinstance(t, HouseholderTransform):
                steps.append((ops.FLOW_HOUSEHOLDER, 0, t.v, None, None))
            elif isinstance(t, SylvesterTransform):
                steps.append((ops.FLOW_SYLVESTER, t.M, t.A, t.B, t.b))
            else:
                raise NotImplementedError("%s is not a 1-D flow" % type(t).__name__)
        return steps

    def planar_params(self):
        return [(t.u, t.w, t.bias) for t in self.transforms]

    def dense_descs(self, masks_fwd, masks_kl):
        """ctypes array of lbbnn_dense_transform_t for an RNVP / MNF flow; masks: lists of (dim,) {0,1} tensors
        (or None) for the forward-draw call and the KL-branch call.  Returns (array, T, keepalive)."""
        from . import _lib
        T = len(self.transforms)
        arr = (_lib.DenseTransform * max(T, 1))()
        keep = []
        for t, tr in enumerate(self.transforms):
            d = arr[t]
            if self.kind == "RNVP":
                lin = [tr.network[0], tr.network[2], tr.network[4], tr.network[6]]
                d.kind, d.hidden = 0, lin[0].out_features
                d.w_in, d.b_in = lin[0].weight.data_ptr(), lin[0].bias.data_ptr()
                for l in range(3):
                    d.w_mid[l], d.b_mid[l] = lin[l + 1].weight.data_ptr(), lin[l + 1].bias.data_ptr()
                d.w_a, d.b_a, d.w_b, d.b_b = (tr.t.weight.data_ptr(), tr.t.bias.data_ptr(),
                                              tr.s.weight.data_ptr(), tr.s.bias.data_ptr())
            else:
                d.kind, d.hidden = 1, tr.f.out_features
                d.w_in, d.b_in = tr.f.weight.data_ptr(), tr.f.bias.data_ptr()
                d.w_a, d.b_a, d.w_b, d.b_b = (tr.g.weight.data_ptr(), tr.g.bias.data_ptr(),
                                              tr.k.weight.data_ptr(), tr.k.bias.data_ptr())
            for name, ms in (("mask_fwd", masks_fwd), ("mask_kl", masks_kl)):
                if ms is not None:
                    m = ms[t].reshape(-1).contiguous().float()
                    keep.append(m)
                    setattr(d, name, m.data_ptr())
        return arr, T, keep

    # -------------------------------------------------------------------------------------------- stand-alone forward
    masks = None          # dense kinds: list of T {0,1} tensors shaped like z (explicit Bernoulli draws, parity tests);
                          # None = Bernoulli(0.5) drawn in-kernel from the device Philox state
    keep_masks = False    # dense kinds: keep the masks of the last call in ``last_masks`` (T, R, I)
    last_masks = None

    def _needs_grad(self, z):
        return torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _dense_spec(self):
        return [{k: v for k, v in tr.named_parameters()} for tr in self.transforms]

    def _shape_logdet(self, ld_rows, one_d):
        """The shape flows2.PropagateFlow.forward's ``logdet`` has for this flow kind (flows2.py:41-46): RNVP sums over
        the last axis (:218-219 -> 0-d for a 1-D z, (R,) for (R,I)); the MNF type sums over everything (:240-241 -> 0-d);
        planar / Sylvester / Householder / mixed give a 0-d value per 1-D z (:94-95,117-120,135), Radial a (1,) one
        (:68: beta has shape (1,)); row-wise for a 2-D z."""
        if self.kind == "MNF":
            return ld_rows.sum()
        if one_d:
            return ld_rows.reshape(1) if self.kind == "Radial" else ld_rows.reshape(())
        return ld_rows

    def forward(self, z):
        """``(z, logdet) = flow(z)`` as flows2.PropagateFlow.forward (flows2.py:41-46), for a 1-D z (what ``r_flow(z2)``
        passes, LBBNN-GP-MF-MNF.py:222) or an (R,I) z (what ``z_flow(self.z)`` passes, :186): z comes back with the shape
        it went in with.  ONE launch for the whole chain:
          RNVP / MNF type  lbbnn_flow_dense_rows -- every row with its own Bernoulli masks, the affine steps on MFMA;
          1-D kinds        lbbnn_flow_chain (1-D z) / lbbnn_flow_chain_rows ((R,I) z, the row-wise restatement of SURVEY.md
                           8(a) F1: the reference itself raises there for all of them but Radial).
        When a gradient is required (grad mode on and z or a flow parameter requires grad) the rows go one by one through
        the differentiable 1-D forms (dense kinds: lbbnn_flow_dense_apply[_backward]; 1-D kinds: the vector-sized torch graph
        of ``_grad``) -- correct but R launches; wrap evaluation-only calls in ``torch.no_grad()`` for the one-launch form."""
        if not z.is_cuda:
            raise RuntimeError("bnn_amd: PropagateFlow.forward needs a HIP device tensor; there is no CPU path")
        if z.dim() not in (1, 2) or z.shape[-1] != self.dim:
            raise RuntimeError("bnn_amd: flow of dim %d got z of shape %s" % (self.dim, tuple(z.shape)))
        one_d = z.dim() == 1
        dense = self.kind not in VECTOR_KINDS
        if self._needs_grad(z):
            # differentiable path, row by row (the fast row-batched kernels below are forward-only): each row goes through
            # the 1-D autograd forms -- dense kinds lbbnn_flow_dense_apply[_backward], 1-D kinds the vector-sized torch graph
            from . import _grad
            rows = z.float().reshape(-1, self.dim)
            R = rows.shape[0]
            masks = None
            if dense:
                masks = self.masks if self.masks is not None else \
                    [torch.bernoulli(torch.full_like(rows, 0.5)) for _ in self.transforms]
                masks = [m.reshape(R, self.dim).to(rows.device).float() for m in masks]
                if self.keep_masks:
                    self.last_masks = torch.stack(masks)
            spec = self._dense_spec()
            outs, lds = [], []
            for r in range(R):
                if dense:
                    o, l = _grad._dense_hip(rows[r], self.kind, spec, [m[r].contiguous() for m in masks])
                else:
                    o, l = _grad._vector(rows[r], spec)
                outs.append(o)
                lds.append(l.reshape(()))
            out = outs[0] if one_d else torch.stack(outs)
            return out, self._shape_logdet(torch.stack(lds), one_d)
        with torch.no_grad():
            zz = z.detach().float()
            if dense:
                rows = zz.reshape(1, -1) if one_d else zz
                masks = None
                if self.masks is not None:
                    masks = torch.stack([m.reshape(rows.shape).to(rows.device).float() for m in self.masks])
                descs, T, keep = self.dense_descs(None, None)
                st = ops.RngState.get(zz.device) if masks is None else None
                out, ld_rows, used = ops.flow_dense_rows(descs, T, rows, masks=masks, rng=st.t if st is not None else None,
                                                         want_masks=self.keep_masks, keep=keep)
                if st is not None:
                    st.advance(1)
                if self.keep_masks:
                    self.last_masks = used
                return (out.reshape(-1) if one_d else out), self._shape_logdet(ld_rows, one_d)
            if one_d:
                out, ld = ops.flow_chain(self.chain_steps(), I=self.dim, z_in=zz.contiguous())
                return out, self._shape_logdet(ld, True)
            out, ld_rows = ops.flow_chain_rows(self.chain_steps(), zz)
            return out, ld_rows
