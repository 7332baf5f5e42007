"""The noise generator the kernels carry (csrc/lbbnn_device.h: philox_normal4) against its numpy restatement
(tests/philox_ref.py), and that restatement against Random123's known-answer vectors for Philox4x32-10."""
import math

import numpy as np
import pytest
import torch

import philox_ref as P


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def bnn():
    import bnn_amd
    return bnn_amd


def test_philox4x32_10_known_answers():
    for ctr, key, want in P.KAT:
        got = P.philox4x32_10(*[np.array([v], dtype=np.uint64) for v in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_restatement_is_standard_normal():
    n = P.normal_matrix(seed=77, offset=3, stream=5, rows=256, cols=1000).reshape(-1)
    assert abs(n.mean()) < 8e-3 and abs(n.var() - 1) < 1.5e-2 and abs((n ** 4).mean() - 3) < 0.12
    a = P.normal_matrix(77, 3, 5, 8, 16)
    assert np.array_equal(a, P.normal_matrix(77, 3, 5, 8, 16))
    assert not np.array_equal(a, P.normal_matrix(77, 4, 5, 8, 16))        # the offset is part of the key
    assert not np.array_equal(a, P.normal_matrix(77, 3, 6, 8, 16))        # the stream is part of the counter
    assert np.array_equal(P.normal_matrix(77, 3, 5, 4, 16, row_base=4), a[4:])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,offset,stream,row_base", [(1234, 0, 5, 0), (2 ** 40 + 17, 9, 64 * 3 + 2, 100000),
                                                          (0, 2 ** 33 + 5, 1, 2 ** 33), (2 ** 63 + 1, 2 ** 32 - 1, 255, 7)])
def test_device_draws_are_the_restated_philox(bnn, dev, seed, offset, stream, row_base):
    """lbbnn_philox_normal (the very function the GEMM epilogues and flow kernels draw through) == Philox4x32-10 with
    the documented keying + Box-Muller, element for element: 2e-5 absolute (hardware log2 / sqrt / sin / cos at ~1 ulp of
    values up to ~6), over 64 x 1201 draws (a ragged last counter) and the 1-D form."""
    ops = bnn.ops
    st = torch.tensor([seed - 2 ** 64 if seed >= 2 ** 63 else seed, offset, 0, 0], dtype=torch.int64, device=dev)
    got = ops.philox_normal(st, stream, 64, 1201, row_base=row_base).cpu().numpy().astype(np.float64)
    ref = P.normal_matrix(seed, offset, stream, 64, 1201, row_base=row_base)
    assert np.abs(got - ref).max() < 2e-5
    got1 = ops.philox_normal(st, stream, 0, 1203).cpu().numpy().astype(np.float64)
    assert np.abs(got1 - P.normal_vector(seed, offset, stream, 1203)).max() < 2e-5


@pytest.mark.gpu
def test_device_draws_follow_the_normal_cdf(bnn, dev):
    """4.9 M draws (one headline layer's worth) against the normal CDF at 13 thresholds: each empirical tail count within
    4.5 standard deviations of its binomial expectation; extremes beyond 5 sigma occur at the expected order."""
    ops = bnn.ops
    bnn.manual_seed(99)
    n = ops.philox_normal(ops.RngState.get(dev).t, 7, 4096, 1200).double().reshape(-1)
    N = n.numel()
    for t in (-4.0, -3.0, -2.0, -1.0, -0.5, -0.1, 0.0, 0.1, 0.5, 1.0, 2.0, 3.0, 4.0):
        p = 0.5 * math.erfc(-t / math.sqrt(2.0))                         # P(X <= t)
        cnt = float((n <= t).sum())
        assert abs(cnt - N * p) < 4.5 * math.sqrt(N * p * (1 - p)) + 1, (t, cnt, N * p)
    far = float((n.abs() > 5.0).sum())                                    # expectation 2.8
    assert far <= 15 and float(n.abs().max()) < 6.7                      # 32-bit radius: |x| <= sqrt(2 * 32 ln 2) = 6.66
