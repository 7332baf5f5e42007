"""Ensemble evaluation of a Bayesian network -- what ``test_ensemble`` computes per test batch
(LBBNN-GP-MF-LRT.py:229-272, LBBNN-GP-MF-MNF.py:277-334), minus the script's file output:

* ``outputs[s] = net(data, sample=True)`` for s < samples (each a fused no-grad HIP forward with its own draws),
* ensemble prediction = argmax of the mean log-probability over the samples (``outputs[0:10].mean(0)``),
* posterior-mean prediction = argmax of ``net(data, sample=False)``,
* density[s] = mean of one Bernoulli draw of every layer's inclusion probabilities (``layer.gamma.rsample()``).

The reference's own ``test_ensemble`` also runs unchanged on these modules; this is the batched convenience form.
"""
from typing import Dict, Optional

import torch


@torch.no_grad()
def ensemble_forward(net, data: torch.Tensor, samples: int = 10) -> torch.Tensor:
    """(samples, B, classes) log-probabilities of ``samples`` stochastic forwards (net left in eval mode)."""
    net.eval()
    outs = [net(data, sample=True) for _ in range(samples)]
    return torch.stack(outs)


@torch.no_grad()
def ensemble_eval(net, data: torch.Tensor, target: Optional[torch.Tensor] = None, samples: int = 10) -> Dict[str, object]:
    outputs = ensemble_forward(net, data, samples)
    density = []
    for _ in range(samples):
        g = [l.gamma.rsample().flatten() for l in (net.l1, net.l2, net.l3)]
        density.append(torch.cat(g).mean())
    pred_ens = outputs.mean(0).argmax(1)
    pred_mean = net(data, sample=False).argmax(1)
    res = {"outputs": outputs, "pred_ensemble": pred_ens, "pred_posterior_mean": pred_mean,
           "density": torch.stack(density)}
    if target is not None:
        res["correct_ensemble"] = int(pred_ens.eq(target).sum())
        res["correct_posterior_mean"] = int(pred_mean.eq(target).sum())
    return res
