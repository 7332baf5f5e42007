"""Light helper objects the reference's training/eval code reaches into
(``layer.weight.sigma``, ``layer.gamma.rsample()``, ``layer.gamma.alpha`` ...).

They mirror ``Gaussian`` / ``Bernoulli`` of LBBNN-GP-MF-LRT.py:73-127 (same attribute and
method names) and run as plain torch ops on whatever device the parameters live on; they are
not on the hot path (the HIP kernels read the parameters directly).
"""
import math

import torch

TEMPER_PRIOR = 0.001   # LBBNN-GP-MF.py:44


class Gaussian(object):
    def __init__(self, mu, rho):
        self.mu = mu
        self.rho = rho

    @property
    def sigma(self):
        return torch.log1p(torch.exp(self.rho))

    def rsample(self):
        return self.mu + self.sigma * torch.randn_like(self.rho)

    def log_prob_iid(self, input):
        s = self.sigma
        return -math.log(math.sqrt(2 * math.pi)) - torch.log(s) - ((input - self.mu) ** 2) / (2 * s ** 2)

    def log_prob(self, input):
        return self.log_prob_iid(input).sum()

    def full_log_prob(self, input, gamma):
        return torch.log(gamma * torch.exp(self.log_prob_iid(input)) + (1 - gamma) + 1e-8).sum()


class Bernoulli(object):
    def __init__(self, alpha, exact=True):
        self._alpha = alpha
        self._provider = None
        self.exact = exact

    def bind(self, provider):
        """``alpha`` is then read from ``provider()`` (the owning layer's sigmoid(lambdal))."""
        self._provider = provider

    @property
    def alpha(self):
        return self._provider() if self._provider is not None else self._alpha

    @alpha.setter
    def alpha(self, value):
        self._alpha = value
        self._provider = None

    def rsample(self):
        from . import base                                 # (base.VALIDATE_ARGS: torch's argument checks cost a host sync per draw)
        if self.exact:
            return torch.distributions.Bernoulli(self.alpha, validate_args=base.VALIDATE_ARGS).sample()
        return torch.distributions.RelaxedBernoulli(probs=self.alpha, temperature=TEMPER_PRIOR,
                                                    validate_args=base.VALIDATE_ARGS).rsample()

    def log_prob(self, input):
        g = torch.round(input.detach()) if self.exact else input
        return (g * torch.log(self.alpha + 1e-8) + (1 - g) * torch.log(1 - self.alpha + 1e-8)).sum()
