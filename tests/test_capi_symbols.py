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


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """Every struct of include/lbbnn.h that the Python side mirrors with ctypes has the same size and the same offset of
    its last member when compiled by gcc from the header itself (an ABI drift here would corrupt kernel arguments)."""
    import subprocess
    from bnn_amd import _lib
    pairs = [("lbbnn_priors_t", _lib.Priors, "bias_sigma_prior"), ("lbbnn_planar_flow_t", _lib.PlanarFlow, "T"),
             ("lbbnn_layer_desc_t", _lib.LayerDesc, "v_scale"), ("lbbnn_gemm_desc_t", _lib.GemmDesc, "head_flags"),
             ("lbbnn_dense_transform_t", _lib.DenseTransform, "mask_kl"),
             ("lbbnn_gate_args_t", _lib.GateArgs, None), ("lbbnn_wpb_args_t", _lib.WpbArgs, None),
             ("lbbnn_adam_list_t", _lib.AdamList, None), ("lbbnn_copy_list_t", _lib.CopyList, None),
             ("lbbnn_dense_layer_t", _lib.DenseLayer, "draw_masks"), ("lbbnn_dense_grad_t", _lib.DenseGrad, "b_b"),
             ("lbbnn_outgrad_args_t", _lib.OutGradArgs, "relu"), ("lbbnn_flow_step_t", _lib.FlowStep, "M"),
             ("lbbnn_flow_chain_t", _lib.FlowChain, "n"), ("lbbnn_planar_grad_t", _lib.PlanarGrad, "b"),
             ("lbbnn_flow_bwd_args_t", _lib.FlowBwdArgs, "layer_id"), ("lbbnn_dense_bwd_args_t", _lib.DenseBwdArgs, "layer_id"),
             ("lbbnn_reduce_job_t", _lib.ReduceJob, "nq"), ("lbbnn_aux_bwd_args_t", _lib.AuxBwdArgs, "layer_id")]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % os.path.join(ROOT, "include", "lbbnn.h"), "int main(void) {"]
    for cname, _, last in pairs:
        lines.append('printf("%s %%zu %%zu\\n", sizeof(%s), %s);' % (cname, cname, "offsetof(%s, %s)" % (cname, last) if last else "(size_t)0"))
    lines += ["return 0; }"]
    src = tmp_path / "sizes.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = dict((l.split()[0], (int(l.split()[1]), int(l.split()[2]))) for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls, last in pairs:
        assert ctypes.sizeof(cls) == out[cname][0], (cname, ctypes.sizeof(cls), out[cname][0])
        if last:
            assert getattr(cls, last).offset == out[cname][1], (cname, last)


def test_new_entry_points_argument_checks(lib):
    """Round-1 additions: early argument checks (nothing is launched, so this runs without a GPU) and size helpers."""
    from bnn_amd import _lib
    fake = ctypes.c_void_p(4096)
    assert lib.lbbnn_mnf_flow_planar_backward_batch(None, 1, None) == -1
    assert lib.lbbnn_mnf_flow_planar_backward_batch((_lib.FlowBwdArgs * 1)(), 0, None) == -2
    assert lib.lbbnn_mnf_flow_planar_backward_batch((_lib.FlowBwdArgs * 1)(), 5, None) == -2
    assert lib.lbbnn_mnf_flow_dense_backward_batch(None, 1, None) == -1
    assert lib.lbbnn_mnf_flow_dense_backward_batch((_lib.DenseBwdArgs * 1)(), 0, None) == -2
    assert lib.lbbnn_mnf_flow_dense_backward((_lib.DenseBwdArgs * 1)(), None) == -5          # neither eps nor rng
    assert lib.lbbnn_lrt_gemm_combine(fake, 8, fake, 32, None, 8, fake, 8, fake, 8, 4, 8, 32, 0, None) == -1
    assert lib.lbbnn_lrt_gemm_combine(fake, 8, fake, 32, fake, 8, fake, 8, fake, 8, 4, 8, 8, 0, None) == -2   # O <= 16
    assert lib.lbbnn_weight_operands_t(None, fake, fake, None, fake, fake, 32, 4, 4, 0, None) == -1
    assert lib.lbbnn_weight_operands_t(fake, fake, fake, None, fake, fake, 30, 4, 4, 0, None) == -3
    assert lib.lbbnn_weight_operands_t(fake, fake, fake, None, fake, fake, 32, 4, 4, 0x100, None) == -4
    assert lib.lbbnn_layers_operands_snap((_lib.LayerDesc * 1)(), 1, fake, None, 1, None) == -1      # rng without a snapshot slot
    assert lib.lbbnn_layers_operands_snap(None, 1, None, None, 0, None) == -1
    # kept intermediates of the dense flows: ZF | ZK ((Tz+1) x I each), ZR (Tr x I), 4 x 128 hidden floats per (transform, path)
    assert lib.lbbnn_flow_dense_save_size(1200, 2, 2) == 1200 * (2 * 3 + 2) + 4 * 2 * 4 * 128
    assert lib.lbbnn_flow_dense_save_size(0, 2, 2) == 0
    assert lib.lbbnn_mnf_flow_dense_backward_workspace(1200) == 2 * 1200 + 2 * 75 * 128 + 2 * 128
    # round 2: the dense flows in two phases (0 = all, 1 = draws + z flow, 2 = r flow + scalars)
    assert lib.lbbnn_layers_dense_flows_phase((_lib.DenseLayer * 1)(), 1, None, 3, None) == -4
    assert lib.lbbnn_layers_dense_flows_phase((_lib.DenseLayer * 1)(), 1, None, -1, None) == -4
    assert lib.lbbnn_layers_dense_flows_phase(None, 1, None, 1, None) == -1
    assert lib.lbbnn_layers_dense_flows_phase((_lib.DenseLayer * 1)(), 0, None, 2, None) == -2


def test_round3_entry_points_argument_checks(lib):
    """lbbnn_lrt_gemm_ex / lbbnn_format_x / lbbnn_weight_pass_f16: early argument checks (nothing is launched)."""
    from bnn_amd import _lib
    fake = ctypes.c_void_p(4096)
    assert lib.lbbnn_lrt_gemm_ex(None, None) == -1
    d = _lib.GemmDesc()
    d.flags = 0x4                                           # not the fp16 format: the descriptor form refuses
    assert lib.lbbnn_lrt_gemm_ex(ctypes.byref(d), None) == -4
    d.flags = 0x40
    d.B, d.I, d.O = 4, 64, 80
    assert lib.lbbnn_lrt_gemm_ex(ctypes.byref(d), None) == -1          # no operands
    for name in ("x", "e_w", "var_w", "mean_scale", "wvar_scale", "out", "eps"):
        setattr(d, name, 4096)
    d.ldx, d.ld, d.ldo = 64, 64, 80
    d.O = 10
    assert lib.lbbnn_lrt_gemm_ex(ctypes.byref(d), None) == -2          # the 10-class head is not this kernel's
    d.O, d.ld = 80, 60
    assert lib.lbbnn_lrt_gemm_ex(ctypes.byref(d), None) == -3
    d.ld, d.flags, d.ldx = 64, 0x40 | 0x100, 80                        # planes: ldx must be a multiple of 32
    assert lib.lbbnn_lrt_gemm_ex(ctypes.byref(d), None) == -3
    d.flags, d.ldx, d.eps = 0x40, 64, None
    assert lib.lbbnn_lrt_gemm_ex(ctypes.byref(d), None) == -5          # neither eps nor rng
    assert lib.lbbnn_format_x(None, 64, fake, 64, 4, 64, None) == -1
    assert lib.lbbnn_format_x(fake, 64, fake, 48, 4, 64, None) == -2
    assert lib.lbbnn_format_x(fake, 64, fake, 96, 4, 60, None) == -3   # I % 8
    assert lib.lbbnn_format_x(fake, 64, fake, 64, 0, 64, None) == 0    # empty batch
    pr = _lib.Priors()
    assert lib.lbbnn_weight_pass_f16(fake, fake, fake, None, None, None, None, ctypes.byref(pr), fake, fake, 64, None, fake,
                                     None, None, None, None, 4, 64, 0, None) == -1   # operands without their scale arrays
    assert lib.lbbnn_weight_pass_f16(fake, fake, fake, None, None, None, None, ctypes.byref(pr), fake, fake, 2048, fake, fake,
                                     None, None, None, None, 4, 2048, 0x80, None) == -2   # rows longer than one register batch
    assert lib.lbbnn_weight_pass_f16(fake, fake, fake, None, None, None, None, ctypes.byref(pr), fake, fake, 64, fake, fake,
                                     None, None, None, None, 4, 64, 0x4, None) == -4      # only LBBNN_F_VAR1 is a flag here
    # lbbnn_head_dw: NULL / paired-argument / shape checks
    assert lib.lbbnn_head_dw(None, None, 10, fake, 64, fake, None, 128, 10, 64, 16, None) == -1
    assert lib.lbbnn_head_dw(fake, fake, 10, fake, 64, fake, None, 128, 10, 64, 16, None) == -1       # gv without dWv
    assert lib.lbbnn_head_dw(fake, None, 10, fake, 64, fake, None, 128, 17, 64, 16, None) == -2       # more than 16 classes
    assert lib.lbbnn_head_dw(fake, None, 8, fake, 64, fake, None, 128, 10, 64, 16, None) == -2        # ldg < C
    assert lib.lbbnn_head_dw(fake, None, 10, fake, 64, fake, None, 8, 10, 64, 16, None) == -2         # more slabs than rows
    # lbbnn_reduce_partials_batch
    assert lib.lbbnn_reduce_partials_batch(None, 0, None) == 0
    assert lib.lbbnn_reduce_partials_batch(None, 1, None) == -1
    jobs = (_lib.ReduceJob * 9)()
    assert lib.lbbnn_reduce_partials_batch(jobs, 9, None) == -2                # more than LBBNN_MAX_REDUCE_JOBS
    assert lib.lbbnn_reduce_partials_batch(jobs, 1, None) == -1                # no workspace
    jobs[0].work, jobs[0].nblk, jobs[0].ncols, jobs[0].nq = 4096, 4, 64, 4
    jobs[0].block_stride = jobs[0].q_stride = 64
    assert lib.lbbnn_reduce_partials_batch(jobs, 1, None) == -2                # nq > 3
    # lbbnn_mnf_aux_backward_batch
    assert lib.lbbnn_mnf_aux_backward_batch(None, 1, None) == -1
    assert lib.lbbnn_mnf_aux_backward_batch((_lib.AuxBwdArgs * 1)(), 5, None) == -2
    assert lib.lbbnn_mnf_aux_backward_batch((_lib.AuxBwdArgs * 1)(), 1, None) == -1
    # lbbnn_elbo_loss_backward_logits
    assert lib.lbbnn_elbo_loss_backward_logits(fake, fake, None, 10, 4, 10, ctypes.c_float(0.1), fake, fake, None, None) == -1
    assert lib.lbbnn_elbo_loss_backward_logits(fake, fake, fake, 8, 4, 10, ctypes.c_float(0.1), fake, fake, None, None) == -2


def test_bias_backward_entry_points_argument_checks(lib):
    """lbbnn_bias_backward / lbbnn_bias_backward_partials (the LRT layer's vector-sized backward): early checks, nothing launched."""
    from bnn_amd import _lib
    fake = ctypes.c_void_p(4096)
    pr = _lib.Priors()
    assert lib.lbbnn_bias_backward(None, fake, fake, fake, fake, ctypes.byref(pr), fake, fake, 10, None) == -1
    assert lib.lbbnn_bias_backward(fake, fake, None, fake, fake, ctypes.byref(pr), fake, fake, 10, None) == -1    # g_sum is not optional
    assert lib.lbbnn_bias_backward(fake, fake, fake, None, None, None, fake, fake, 10, None) == -1               # priors
    assert lib.lbbnn_bias_backward(fake, fake, fake, None, None, ctypes.byref(pr), fake, fake, 0, None) == -2
    assert lib.lbbnn_bias_backward_partials(None, 64, 10, 1, fake, fake, None, ctypes.byref(pr), fake, fake, None) == -1
    assert lib.lbbnn_bias_backward_partials(fake, 64, 10, 1, fake, fake, None, ctypes.byref(pr), None, fake, None) == -1
    assert lib.lbbnn_bias_backward_partials(fake, 0, 10, 1, fake, fake, None, ctypes.byref(pr), fake, fake, None) == -2
    assert lib.lbbnn_bias_backward_partials(fake, 64, 0, 0, fake, fake, None, ctypes.byref(pr), fake, fake, None) == -2
