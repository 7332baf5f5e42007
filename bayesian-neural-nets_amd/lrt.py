"""Names as in LBBNN-GP-MF-LRT.py: ``from bnn_amd.lrt import BayesianLinear, BayesianNetwork``."""
from .distributions import Bernoulli, Gaussian  # noqa: F401
from .layers import LRTBayesianLinear as BayesianLinear  # noqa: F401
from .layers import LRTBayesianNetwork as BayesianNetwork  # noqa: F401
