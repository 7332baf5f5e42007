"""Names as in LBBNN-GP-MF-MNF.py: ``from bnn_amd.mnf import BayesianLinear, BayesianNetwork``."""
from .distributions import Bernoulli, Gaussian  # noqa: F401
from .flows import PropagateFlow  # noqa: F401
from .layers import MNFBayesianLinear as BayesianLinear  # noqa: F401
from .layers import MNFBayesianNetwork as BayesianNetwork  # noqa: F401
