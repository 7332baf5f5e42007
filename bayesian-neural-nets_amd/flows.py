"""Normalizing-flow parameter containers with the reference's module/parameter names
(flows2.py:14-241), so ``state_dict`` keys and optimizer parameter groups carry over:
``z_flow.transforms.N.{u,w,bias}`` (planar), ``….network.{0,2,4,6}.*, t.*, s.*`` (RNVP),
``….{f,g,k}.*`` (MNF).

The transforms hold parameters only; the arithmetic runs in the HIP kernels
(``lbbnn_mnf_flow_planar`` for planar flows).  Initialisation reproduces the reference's
formulas and draw order, so a seeded construction yields the reference's values.
"""
import torch
import torch.nn as nn

from . import ops


def parameter_init(low, high, size):
    # (low - high) * U[0,1) + high, as flows2.py:10-12
    return (low - high) * torch.rand(size) + high


class PlanarTransform(nn.Module):
    """Parameters of one planar transform (flows2.py:72-95): u, w (dim,), bias (1,)."""

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


class RNVP(nn.Module):
    """Parameters of one RNVP coupling transform (flows2.py:188-219)."""

    def __init__(self, dim, h_sizes=(75, 75, 75, 75)):
        super().__init__()
        self.network = _MLP(*([dim] + list(h_sizes)))
        self.t = nn.Linear(h_sizes[-1], dim)
        self.s = nn.Linear(h_sizes[-1], dim)


class MNF(nn.Module):
    """Parameters of one MNF-type transform (flows2.py:225-241)."""

    def __init__(self, dim, hidden=100):
        super().__init__()
        self.f = nn.Linear(dim, hidden)
        self.g = nn.Linear(hidden, dim)
        self.k = nn.Linear(hidden, dim)


class RadialTransform(nn.Module):
    """Parameters of one radial transform (flows2.py:48-69): z_0 (dim,), log_alpha (1,), beta (1,)."""

    def __init__(self, dim):
        super().__init__()
        self.z_0 = nn.Parameter(parameter_init(-0.1, 0.1, dim))
        self.log_alpha = nn.Parameter(parameter_init(-4, 5, 1))
        self.beta = nn.Parameter(parameter_init(-0.1, 0.1, 1))
        self.d = dim


class SylvesterTransform(nn.Module):
    """Parameters of one Sylvester transform, M = 5 (flows2.py:98-120): A (dim,M), B (M,dim), b (M,)."""

    def __init__(self, dim):
        super().__init__()
        self.M = 5
        self.A = nn.Parameter(parameter_init(-0.01, 0.01, (dim, self.M)))
        self.B = nn.Parameter(parameter_init(-0.01, 0.01, (self.M, dim)))
        self.b = nn.Parameter(parameter_init(-0.01, 0.01, self.M))


class HouseholderTransform(nn.Module):
    """Parameters of one Householder transform (flows2.py:122-135): v (dim,)."""

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

    def forward(self, z):
        """Stand-alone flow on a 1-D z (what ``r_flow(z2)`` does at LBBNN-GP-MF-MNF.py:222): one
        lbbnn_flow_chain launch.  A 2-D z is taken row-wise and only its LAST row is computed (the row
        ``sample_z`` keeps, LBBNN-GP-MF-MNF.py:187).  Returns (z (dim,), logdet (1,)); no autograd."""
        if self.kind not in VECTOR_KINDS:
            raise NotImplementedError("stand-alone %s flow forward: dense flows run inside the MNF layer "
                                      "(lbbnn_mnf_flow_dense)" % self.kind)
        if not z.is_cuda:
            raise RuntimeError("bnn_amd: PropagateFlow.forward needs a HIP device tensor; there is no CPU path")
        if z.dim() != 1:
            z = z.reshape(-1, self.dim)[-1]
        with torch.no_grad():
            return ops.flow_chain(self.chain_steps(), I=self.dim, z_in=z.detach().float().contiguous())
