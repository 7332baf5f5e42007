"""CPU tier: the C-ABI library loads and exports every symbol include/lbbnn.h declares
(no compute calls without a GPU); argument checks that need no launch return the documented codes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lbbnn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lbbnn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from bnn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from bnn_amd import _lib
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names          # ctypes table and header agree


def test_version_and_helpers(lib):
    assert lib.lbbnn_abi_version() == 1
    assert lib.lbbnn_operand_ld(784) == 800 and lib.lbbnn_operand_ld(1200) == 1216
    assert lib.lbbnn_operand_ld(32) == 32 and lib.lbbnn_operand_ld(1) == 32
    assert b"NULL" in lib.lbbnn_error_string(-1)
    assert lib.lbbnn_error_string(0) == b"ok"


def test_argument_checks_return_codes_without_launching(lib):
    from bnn_amd._lib import Priors
    pr = Priors()
    # NULL required pointers
    assert lib.lbbnn_weight_pass(None, None, None, None, None, None, None, ctypes.byref(pr),
                                 None, None, 32, None, None, None, None, 4, 4, 0, None) == -1
    fake = ctypes.c_void_p(4096)     # never dereferenced: the call must fail on shape before launching
    assert lib.lbbnn_weight_pass(fake, fake, fake, None, None, None, None, ctypes.byref(pr),
                                 None, None, 32, None, None, None, None, 0, 4, 0, None) == -2
    assert lib.lbbnn_weight_pass(fake, fake, fake, None, None, None, None, ctypes.byref(pr),
                                 fake, fake, 30, None, None, None, None, 4, 4, 0, None) == -3
    assert lib.lbbnn_lrt_gemm(None, 8, fake, fake, 32, None, None, None, None, None, 0, 0,
                              fake, 4, 4, 8, 4, 0, None) == -1
    assert lib.lbbnn_lrt_gemm(fake, 8, fake, fake, 32, None, None, None, None, None, 0, 0,
                              fake, 4, 4, 8, 4, 0, None) == -5          # no eps and no rng
    assert lib.lbbnn_lrt_gemm(fake, 8, fake, fake, 32, None, None, None, fake, None, 0, 0,
                              fake, 4, 4, 8, 4, 0x100, None) == -4      # unknown flag
    assert lib.lbbnn_log_softmax_rows(fake, 100, fake, 100, 4, 65, None) == -2
    assert lib.lbbnn_mnf_flow_planar(fake, fake, None, None, None, 0, None, None, None, 0, fake, fake,
                                     None, 0, fake, fake, fake, 20000, 1, None) == -2
