#!/usr/bin/env python3
"""Phase stamps of the dense-flow output stage (lab build: LBBNN_LIB_PATH=tools/lab/liblbbnn_dstamps.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bnn_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = bnn_amd.mnf.BayesianNetwork((784, 1200, 1200, 10), 2, z_flow_type="RNVP", r_flow_type="RNVP").to(dev).train()
x = torch.rand(4096, 784, device=dev)
with torch.no_grad():
    for _ in range(5):
        net(x, sample=True)
    torch.cuda.synchronize()
    for l in (net.l1, net.l2, net.l3):
        print("I=%d  loads landed, LDS filled %.2f us | chain done %.2f us | end %.2f us"
              % ((l.in_features,) + tuple(float(v) * 0.01 for v in l._workspace().scal[5:8])))
