import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # keep csrc/liblbbnn_hip.so in step with its sources (make is a no-op when up to date)
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "bayesian-neural-nets_amd", "csrc")
    if shutil.which("make") and os.path.exists("/opt/rocm/bin/hipcc"):
        r = subprocess.run(["make", "-C", csrc, "-j4"], capture_output=True, text=True)
        if r.returncode != 0 and not os.path.exists(os.path.join(csrc, "liblbbnn_hip.so")):
            raise RuntimeError("building the HIP extension failed:\n" + r.stdout + r.stderr)


class Golden:
    """Lazy view over one tests/golden/*.npz: ``g.case('c0')`` -> dict of torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def cases(self, prefix="c"):
        return sorted({k.split(".")[0] for k in self._z.files if k.startswith(prefix)})

    def case(self, case):
        pre = case + "."
        out = {}
        for k in self._z.files:
            if k.startswith(pre):
                a = self._z[k]
                out[k[len(pre):]] = a if a.dtype.kind in "US" else torch.from_numpy(np.array(a))
        return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def sub(d, prefix):
    """{'p.a': x, 'p.b': y} -> {'a': x, 'b': y}"""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel_err(a, b):
    """max|a-b| / max(|b|) -- the relative error used for every fp32 parity bar in this repo."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / denom)


def elementwise_violation(a, b, rtol=1e-4, atol_frac=1e-6):
    """Element-wise form of the parity bar: the worst value of |a-b| / (atol + rtol |b|) with
    atol = atol_frac * max|b| -- <= 1 means torch.allclose(a, b, rtol, atol) holds.  Unlike rel_err (a global norm) a
    small-magnitude element cannot hide behind the largest one beyond the stated atol."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    atol = atol_frac * float(b.abs().max().clamp_min(1e-30))
    return float(((a - b).abs() / (atol + rtol * b.abs())).max())


requires_gpu = pytest.mark.gpu
