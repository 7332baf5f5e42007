#!/usr/bin/env python3
"""How a user of the reference's LBBNN-GP-MF-MNF.py switches to the HIP path: the script's own class definitions are
replaced by one import, its training iteration (:263-275: net(data, sample=True), nll + net.kl()/NUM_BATCHES, backward,
Adam step) and its ensemble test (:277-334: gamma.rsample(), net(data, sample=True) x TEST_SAMPLES, net(data,
sample=False)) run unchanged in meaning.  There is no dataset in this image, so MNIST-shaped synthetic data with learnable
labels stands in for the loaders.

    python examples/train_mnf_synthetic.py            # eager loop, the reference's default RNVP flows
    GRAPH=1 python examples/train_mnf_synthetic.py    # the same step captured once in a HIP graph and replayed
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import bnn_amd
from bnn_amd.mnf import BayesianNetwork            # was: class Gaussian / Bernoulli / BayesianLinear / BayesianNetwork inline
from bnn_amd.evaluate import ensemble_eval

DEVICE = torch.device("cuda:0")
BATCH_SIZE, NUM_BATCHES, EPOCHS, TEST_SAMPLES = 1000, 12, 6, 10
bnn_amd.set_precision("bf16x3")
torch.manual_seed(1)                                # the reference seeds per run (:409); also seeds the in-kernel noise

net = BayesianNetwork().to(DEVICE)                  # 784-400-600-10, RNVP flows, num_transforms=2 (:244-250)
optimizer = bnn_amd.optim.Adam(net.parameters(), lr=1e-3)      # torch.optim.Adam works too (:358)

g = torch.Generator(device=DEVICE).manual_seed(7)
proj = torch.randn(784, 10, device=DEVICE, generator=g)
train_x = torch.rand(NUM_BATCHES, BATCH_SIZE, 1, 28, 28, device=DEVICE, generator=g)
train_y = (train_x.view(NUM_BATCHES, BATCH_SIZE, 784) @ proj).argmax(-1)
test_x = torch.rand(BATCH_SIZE, 1, 28, 28, device=DEVICE, generator=g)
test_y = (test_x.view(BATCH_SIZE, 784) @ proj).argmax(-1)


def elbo(net, data, target):
    outputs = net(data, sample=True)
    return F.nll_loss(outputs, target, reduction="sum") + net.kl() / NUM_BATCHES


net.train()
if os.environ.get("GRAPH") == "1":
    step = bnn_amd.graphs.make_graphed_train_step(net, optimizer, elbo, train_x[0], train_y[0])
else:
    def step(data, target):
        net.zero_grad()
        loss = elbo(net, data, target)
        loss.backward()
        optimizer.step()
        return loss

for epoch in range(EPOCHS):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in range(NUM_BATCHES):
        loss = step(train_x[b], train_y[b])
    torch.cuda.synchronize()
    print("epoch %d  loss %.1f  (%.2f ms/iteration)" % (epoch, float(loss.detach()), (time.perf_counter() - t0) / NUM_BATCHES * 1e3))

res = ensemble_eval(net, test_x, test_y, samples=TEST_SAMPLES)
print("density %.3f | posterior mean %.3f | ensemble %.3f" % (float(res["density"].mean()),
      res["correct_posterior_mean"] / BATCH_SIZE, res["correct_ensemble"] / BATCH_SIZE))
