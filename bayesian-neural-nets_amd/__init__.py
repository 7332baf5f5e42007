"""bnn_amd -- MI355X-native (gfx950) Bayesian linear-layer hot path of LarsELund/Bayesian-Neural-Nets.

Hand-written HIP kernels behind a C ABI (include/lbbnn.h), exposed as drop-in ``torch.nn.Module``s
with the reference's constructor / forward / kl signatures.  See DESIGN.md.
"""
from . import ops  # noqa: F401
from ._lib import LIB_PATH, Priors  # noqa: F401
from .layers import (LRTBayesianLinear, LRTBayesianNetwork, MNFBayesianLinear,  # noqa: F401
                     MNFBayesianNetwork)
from .ops import get_precision, manual_seed, set_precision  # noqa: F401
from . import base, evaluate, flows, graphs, losses, lrt, mnf, optim, parallel, vd  # noqa: F401
from .losses import elbo_loss  # noqa: F401

__version__ = "0.1.0"
