"""ctypes binding of the C ABI in include/lbbnn.h.

There is deliberately NO fallback: if ``csrc/liblbbnn_hip.so`` is missing, or a tensor is not on
a HIP device, the call raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is
never imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LBBNN_LIB_PATH: developer hook for the ablation builds of tools/lab (must exist; still no fallback)
LIB_PATH = os.environ.get("LBBNN_LIB_PATH") or os.path.join(_HERE, "csrc", "liblbbnn_hip.so")

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_u32 = ctypes.c_uint32
c_i64 = ctypes.c_int64
c_u64 = ctypes.c_uint64


class Priors(ctypes.Structure):
    """lbbnn_priors_t (include/lbbnn.h)."""
    _fields_ = [("mu_prior", ctypes.c_float), ("sigma_prior", ctypes.c_float),
                ("alpha_prior", ctypes.c_float), ("bias_mu_prior", ctypes.c_float),
                ("bias_sigma_prior", ctypes.c_float)]

    def __init__(self, mu_prior=0.0, sigma_prior=1.0, alpha_prior=0.05, bias_mu_prior=0.0,
                 bias_sigma_prior=1.0):
        super().__init__(mu_prior, sigma_prior, alpha_prior, bias_mu_prior, bias_sigma_prior)


MAX_FLOW_T = 16
MAX_LAYERS = 4


class PlanarFlow(ctypes.Structure):
    """lbbnn_planar_flow_t"""
    _fields_ = [("u", c_p * MAX_FLOW_T), ("w", c_p * MAX_FLOW_T), ("b", c_p * MAX_FLOW_T), ("T", c_i)]


class LayerDesc(ctypes.Structure):
    """lbbnn_layer_desc_t (field order must match include/lbbnn.h)"""
    _fields_ = [("weight_mu", c_p), ("weight_rho", c_p), ("lambdal", c_p), ("bias_mu", c_p), ("bias_rho", c_p),
                ("q0_mean", c_p), ("q0_log_var", c_p), ("r0_c", c_p), ("r0_b1", c_p), ("r0_b2", c_p),
                ("z_flow", PlanarFlow), ("r_flow", PlanarFlow), ("priors", Priors),
                ("O", c_i), ("I", c_i), ("layer_id", c_u32), ("stochastic", c_i), ("want_kl", c_i), ("split", c_i),
                ("eps_z", c_p), ("eps_z2", c_p), ("eps_act", c_p),
                ("z_fwd", c_p), ("z_kl", c_p), ("scal", c_p), ("e_w", c_p), ("var_w", c_p),
                ("kl_rows", c_p), ("act_mu", c_p), ("act_var", c_p), ("bias_var", c_p), ("kl_layer", c_p),
                ("flows_done", c_i), ("e_scale", c_p), ("v_scale", c_p)]


class GemmDesc(ctypes.Structure):
    """lbbnn_gemm_desc_t"""
    _fields_ = [("x", c_p), ("ldx", c_i), ("e_w", c_p), ("var_w", c_p), ("ld", c_i),
                ("mean_scale", c_p), ("wvar_scale", c_p), ("bias_mean", c_p), ("bias_var", c_p), ("var_scale", c_p),
                ("eps", c_p), ("rng", c_p), ("rng_stream", c_u32), ("row_offset", c_i64),
                ("out", c_p), ("ldo", c_i), ("out_planes", c_p), ("ldp", c_i), ("std_out", c_p),
                ("B", c_i), ("I", c_i), ("O", c_i), ("flags", c_i),
                ("layers", ctypes.POINTER(LayerDesc)), ("n_layers", c_i), ("fin_rng", c_p), ("kl_total", c_p),
                ("rng_live", c_p), ("advance", c_u64),
                ("head_e", c_p), ("head_v", c_p), ("head_ld", c_i), ("head_classes", c_i),
                ("head_bias_mean", c_p), ("head_bias_var", c_p), ("head_eps", c_p), ("head_rng_stream", c_u32),
                ("head_out", c_p), ("head_ldo", c_i), ("head_slab", c_p), ("head_flags", c_i)]


class DenseTransform(ctypes.Structure):
    """lbbnn_dense_transform_t"""
    _fields_ = [("kind", c_i), ("hidden", c_i), ("w_in", c_p), ("b_in", c_p),
                ("w_mid", c_p * 3), ("b_mid", c_p * 3), ("w_a", c_p), ("b_a", c_p), ("w_b", c_p), ("b_b", c_p),
                ("mask_fwd", c_p), ("mask_kl", c_p)]


class GateArgs(ctypes.Structure):
    """lbbnn_gate_args_t"""
    _fields_ = [(n, c_p) for n in ("mu", "rho", "gamma_alpha", "cgamma", "eps_w", "alpha_attr",
                                   "bias_mu", "bias_rho", "eps_b", "bias_a", "bias_b", "tau_b",
                                   "weight_a", "weight_b", "tau_w", "pa", "pb",
                                   "w_out", "bias_out", "rows", "log_prior", "log_q")] + \
               [(n, c_i) for n in ("O", "I", "ld", "mode", "exact", "want_lp", "flags")] + [("layer_id", c_u32)]


class ReduceJob(ctypes.Structure):
    """lbbnn_reduce_job_t (include/lbbnn.h)."""
    _fields_ = [("work", c_p), ("out", c_p * 3), ("block_stride", ctypes.c_int64), ("q_stride", ctypes.c_int64),
                ("nblk", c_i), ("ncols", c_i), ("nq", c_i)]


class AuxBwdArgs(ctypes.Structure):
    """lbbnn_aux_bwd_args_t (include/lbbnn.h)."""
    _fields_ = [(n, c_p) for n in ("act_mu", "act_var", "eps_act", "r0_b1", "r0_b2", "zb_last", "g_kl", "da_mu", "da_var", "aux",
                                   "rng")] + [("O", c_i), ("I", c_i), ("layer_id", ctypes.c_uint32)]


class WpbArgs(ctypes.Structure):
    """lbbnn_wpb_args_t"""
    _fields_ = [(n, c_p) for n in ("mu", "rho", "lambdal", "dWm", "dWv", "z_fwd", "z_kl", "r0_c",
                                   "da_mu", "da_var", "g_kl")] + [("priors", Priors)] + \
               [(n, c_p) for n in ("dmu", "drho", "dlambdal", "dz_fwd", "dz_kl", "dr0_c", "work")] + \
               [("O", c_i), ("I", c_i), ("nsplit", c_i), ("split_stride", c_i64)]


ADAM_MAX_TENSORS = 80


class AdamList(ctypes.Structure):
    """lbbnn_adam_list_t"""
    _fields_ = [("p", c_p * ADAM_MAX_TENSORS), ("g", c_p * ADAM_MAX_TENSORS), ("m", c_p * ADAM_MAX_TENSORS),
                ("v", c_p * ADAM_MAX_TENSORS), ("numel", c_i64 * ADAM_MAX_TENSORS), ("n", c_i)]


class CopyList(ctypes.Structure):
    """lbbnn_copy_list_t"""
    _fields_ = [("dst", c_p * ADAM_MAX_TENSORS), ("src", c_p * ADAM_MAX_TENSORS), ("numel", c_i64 * ADAM_MAX_TENSORS), ("n", c_i)]


class DenseLayer(ctypes.Structure):
    """lbbnn_dense_layer_t"""
    _fields_ = [("q0_mean", c_p), ("q0_log_var", c_p), ("zt", ctypes.POINTER(DenseTransform)), ("rt", ctypes.POINTER(DenseTransform)),
                ("eps_fwd", c_p), ("eps_kl", c_p), ("z_fwd", c_p), ("z_kl", c_p), ("scal", c_p), ("work", c_p),
                ("Tz", c_i), ("Tr", c_i), ("I", c_i), ("want_kl", c_i), ("layer_id", c_u32), ("save", c_p), ("draw_masks", c_i)]


class DenseGrad(ctypes.Structure):
    """lbbnn_dense_grad_t"""
    _fields_ = [("w_in", c_p), ("b_in", c_p), ("w_mid", c_p * 3), ("b_mid", c_p * 3),
                ("w_a", c_p), ("b_a", c_p), ("w_b", c_p), ("b_b", c_p)]


MAX_DENSE_T = 8


class GateBwdArgs(ctypes.Structure):
    """lbbnn_gate_bwd_args_t"""
    _fields_ = [(n, c_p) for n in ("mu", "rho", "gamma_alpha", "cgamma", "eps_w", "bias_mu", "bias_rho", "eps_b", "bias_a",
                                   "bias_b", "tau_b", "weight_a", "weight_b", "tau_w", "pa", "pb", "dW", "g_sum", "g_lp", "g_lq",
                                   "d_mu", "d_rho", "d_cgamma", "d_alpha", "w_out", "d_bias_mu", "d_bias_rho", "d_bias_a",
                                   "d_bias_b", "d_tau_b", "d_scalars", "rows")] + \
               [("O", c_i), ("I", c_i), ("exact", c_i), ("layer_id", c_u32)]


class OutGradArgs(ctypes.Structure):
    """lbbnn_outgrad_args_t"""
    _fields_ = [(n, c_p) for n in ("g_out", "out", "std", "eps", "rng", "gm", "gv", "gmT", "gvT", "g_sum", "gv_sum", "work")] + \
               [("row_offset", c_i64), ("rng_stream", c_u32)] + [(n, c_i) for n in ("B", "O", "ldg", "ldo", "relu")] + \
               [("gv_scale", c_p)]


class FlowStep(ctypes.Structure):
    """lbbnn_flow_step_t"""
    _fields_ = [("p0", c_p), ("p1", c_p), ("p2", c_p), ("type", c_i), ("M", c_i)]


class FlowChain(ctypes.Structure):
    """lbbnn_flow_chain_t"""
    _fields_ = [("step", FlowStep * MAX_FLOW_T), ("n", c_i)]


class PlanarGrad(ctypes.Structure):
    """lbbnn_planar_grad_t"""
    _fields_ = [("u", c_p * MAX_FLOW_T), ("w", c_p * MAX_FLOW_T), ("b", c_p * MAX_FLOW_T)]


class FlowBwdArgs(ctypes.Structure):
    """lbbnn_flow_bwd_args_t"""
    _fields_ = [(n, c_p) for n in ("q0_mean", "q0_log_var", "eps_fwd", "eps_kl", "r0_b1", "r0_b2", "aux",
                                   "dz_fwd", "dz_kl", "g_kl", "bias_mu", "bias_rho", "g_sum", "gv_sum")] + \
               [("z_flow", PlanarFlow), ("r_flow", PlanarFlow), ("priors", Priors)] + \
               [(n, c_p) for n in ("d_q0_mean", "d_q0_log_var", "d_r0_b1", "d_r0_b2", "d_bias_mu", "d_bias_rho")] + \
               [("d_z_flow", PlanarGrad), ("d_r_flow", PlanarGrad), ("work", c_p), ("O", c_i), ("I", c_i),
                ("rng", c_p), ("layer_id", c_u32)]


class DenseBwdArgs(ctypes.Structure):
    """lbbnn_dense_bwd_args_t"""
    _fields_ = [(n, c_p) for n in ("q0_mean", "q0_log_var", "eps_fwd", "eps_kl", "r0_b1", "r0_b2", "aux",
                                   "dz_fwd", "dz_kl", "g_kl", "bias_mu", "bias_rho", "g_sum", "gv_sum")] + \
               [("zt", ctypes.POINTER(DenseTransform)), ("rt", ctypes.POINTER(DenseTransform)),
                ("d_zt", ctypes.POINTER(DenseGrad)), ("d_rt", ctypes.POINTER(DenseGrad)), ("priors", Priors)] + \
               [(n, c_p) for n in ("d_q0_mean", "d_q0_log_var", "d_r0_b1", "d_r0_b2", "d_bias_mu", "d_bias_rho", "save", "work")] + \
               [("Tz", c_i), ("Tr", c_i), ("O", c_i), ("I", c_i), ("rng", c_p), ("layer_id", c_u32)]


# name -> (restype, argtypes); must list every symbol include/lbbnn.h declares
SIGNATURES = {
    "lbbnn_abi_version": (c_i, []),
    "lbbnn_error_string": (ctypes.c_char_p, [c_i]),
    "lbbnn_operand_ld": (c_i, [c_i]),
    "lbbnn_weight_pass": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.POINTER(Priors),
                                c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_weight_pass_f16": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.POINTER(Priors),
                                    c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_lrt_gemm_ex": (c_i, [ctypes.POINTER(GemmDesc), c_p]),
    "lbbnn_head_slab_floats": (c_i64, [c_i, c_i]),
    "lbbnn_format_x": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_lrt_gemm": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_u32, c_i64,
                             c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_lrt_gemm_train": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_u32, c_i64,
                                   c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_transpose_operand": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_weight_pass_backward_workspace": (c_i64, [c_i, c_i]),
    "lbbnn_weight_pass_backward": (c_i, [ctypes.POINTER(WpbArgs), c_p]),
    "lbbnn_layers_dense_flows": (c_i, [ctypes.POINTER(DenseLayer), c_i, c_p, c_p]),
    "lbbnn_layers_dense_flows_phase": (c_i, [ctypes.POINTER(DenseLayer), c_i, c_p, c_i, c_p]),
    "lbbnn_flow_dense_save_size": (c_i64, [c_i, c_i, c_i]),
    "lbbnn_mnf_flow_dense_backward_workspace": (c_i64, [c_i]),
    "lbbnn_mnf_flow_dense_backward": (c_i, [ctypes.POINTER(DenseBwdArgs), c_p]),
    "lbbnn_mnf_flow_dense_backward_batch": (c_i, [ctypes.POINTER(DenseBwdArgs), c_i, c_p]),
    "lbbnn_flow_dense_apply_workspace": (c_i64, [c_i, c_i]),
    "lbbnn_flow_dense_apply": (c_i, [ctypes.POINTER(DenseTransform), c_i, c_i, c_p, c_i, c_p, c_p, c_p]),
    "lbbnn_flow_dense_apply_backward": (c_i, [ctypes.POINTER(DenseTransform), ctypes.POINTER(DenseGrad), c_i, c_i, c_p, c_p,
                                              c_p, c_i, c_p, c_p, c_p]),
    "lbbnn_q0_rows": (c_i, [c_p, c_p, c_p, c_p, c_u32, c_i, c_i, c_p, c_p]),
    "lbbnn_flow_dense_rows_max_dim": (c_i, []),
    "lbbnn_flow_dense_rows": (c_i, [ctypes.POINTER(DenseTransform), c_i, c_p, c_p, c_p, c_u32, c_u64, c_p, c_i, c_i, c_i,
                                    c_p, c_i, c_p, c_p]),
    "lbbnn_gate_backward": (c_i, [ctypes.POINTER(GateBwdArgs), c_p, c_p]),
    "lbbnn_format_operand": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_multi_copy": (c_i, [ctypes.POINTER(CopyList), c_p]),
    "lbbnn_adam_step": (c_i, [ctypes.POINTER(AdamList), ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                              ctypes.c_float, c_p, c_i, c_p]),
    "lbbnn_matmul_splitk": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_lrt_gemm_combine": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_dx_combine": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_output_grad_workspace": (c_i64, [c_i, c_i]),
    "lbbnn_output_grad": (c_i, [ctypes.POINTER(OutGradArgs), c_p]),
    "lbbnn_head_dx": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_head_dw": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_reduce_partials_batch": (c_i, [c_p, c_i, c_p]),
    "lbbnn_flow_chain": (c_i, [ctypes.POINTER(FlowChain), c_p, c_p, c_p, c_p, c_p, c_u32, c_i, c_p, c_p, c_p, c_p, c_p]),
    "lbbnn_flow_chain_rows": (c_i, [ctypes.POINTER(FlowChain), c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    "lbbnn_mnf_aux_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_u32, c_p]),
    "lbbnn_mnf_aux_backward_batch": (c_i, [c_p, c_i, c_p]),
    "lbbnn_bias_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "lbbnn_bias_backward_partials": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "lbbnn_mnf_flow_backward_workspace": (c_i64, [c_i, c_i, c_i]),
    "lbbnn_mnf_flow_planar_backward": (c_i, [ctypes.POINTER(FlowBwdArgs), c_p]),
    "lbbnn_mnf_flow_planar_backward_batch": (c_i, [ctypes.POINTER(FlowBwdArgs), c_i, c_p]),
    "lbbnn_mnf_flow_planar": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p,
                                    c_p, c_u32, c_p, c_p, c_p, c_i, c_i, c_p]),
    "lbbnn_flow_dense_workspace": (c_i64, [c_i]),
    "lbbnn_mnf_flow_dense": (c_i, [c_p, c_p, ctypes.POINTER(DenseTransform), c_i, ctypes.POINTER(DenseTransform), c_i,
                                   c_p, c_p, c_p, c_u32, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "lbbnn_kl_finalize": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p,
                                ctypes.POINTER(Priors), c_p, c_u32, c_p, c_p, c_i, c_p]),
    "lbbnn_layers_prepare": (c_i, [ctypes.POINTER(LayerDesc), c_i, c_p, c_p]),
    "lbbnn_layers_operands_snap": (c_i, [ctypes.POINTER(LayerDesc), c_i, c_p, c_p, c_u64, c_p]),
    "lbbnn_layers_operands_x": (c_i, [ctypes.POINTER(LayerDesc), c_i, c_p, c_p, c_u64, c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_lrt_gemm_finalize": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_u32, c_i64,
                                      c_p, c_i, c_p, c_i, c_i, c_i, c_i, ctypes.POINTER(LayerDesc), c_i, c_p, c_p, c_p]),
    "lbbnn_lrt_gemm_finalize_adv": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_u32, c_i64,
                                          c_p, c_i, c_p, c_i, c_i, c_i, c_i, ctypes.POINTER(LayerDesc), c_i, c_p, c_p,
                                          c_p, c_u64, c_p]),
    "lbbnn_ensemble_operands": (c_i, [ctypes.POINTER(LayerDesc), c_i, c_i, c_p, c_u64, c_p]),
    "lbbnn_lrt_gemm_members": (c_i, [c_p, c_i, c_i64, c_p, c_i64, c_p, c_i, c_p, c_p, c_p, c_u32, c_i64, c_u64,
                                     c_p, c_i, c_i64, c_i, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_layers_operands": (c_i, [ctypes.POINTER(LayerDesc), c_i, c_p, c_p]),
    "lbbnn_layers_finalize": (c_i, [ctypes.POINTER(LayerDesc), c_i, c_p, ctypes.c_uint64, c_p, c_p]),
    "lbbnn_forward_finish": (c_i, [c_p, c_u64, c_p, c_i, c_p, c_p]),
    "lbbnn_gate_sample": (c_i, [ctypes.POINTER(GateArgs), c_p, c_p]),
    "lbbnn_vd_operands": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_weight_operands_t": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "lbbnn_rng_advance": (c_i, [c_p, c_u64, c_p]),
    "lbbnn_philox_normal": (c_i, [c_p, c_u32, c_i64, c_i64, c_i64, c_p, c_p]),
    "lbbnn_elbo_loss": (c_i, [c_p, c_i, c_p, c_i, c_i, c_p, ctypes.c_float, c_p, c_p]),
    "lbbnn_elbo_loss_backward": (c_i, [c_p, c_p, c_i, c_i, ctypes.c_float, c_p, c_p, c_p]),
    "lbbnn_elbo_loss_backward_logits": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, ctypes.c_float, c_p, c_p, c_p, c_p]),
    "lbbnn_log_softmax_backward": (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
    "lbbnn_log_softmax_rows": (c_i, [c_p, c_i, c_p, c_i, c_i, c_i, c_p]),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "bnn_amd: HIP extension not built: %s is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C bayesian-neural-nets_amd/csrc` (hipcc, --offload-arch=gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError => header/library out of sync: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = l
    if RECORD is not None:
        return _Recorder(_lib)
    return _lib


# graphs.LaunchPlan: while RECORD is a list, every C call made through lib() is appended to it as (function, arguments) -- the
# arguments are what ctypes was given (ints, None, ctypes arrays / structs: the latter stay alive with the record)
RECORD = None


class _Recorder:
    def __init__(self, l):
        self._l = l

    def __getattr__(self, name):
        fn = getattr(self._l, name)

        def call(*args):
            # only what ENQUEUES is recorded: every such entry point ends in `void* stream`; the size / version helpers do not
            if RECORD is not None and fn.argtypes and fn.argtypes[-1] is ctypes.c_void_p:
                RECORD.append((name, fn, args))
            return fn(*args)
        return call


def check(rc, what):
    if rc != 0:
        msg = lib().lbbnn_error_string(rc)
        raise RuntimeError("bnn_amd: %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
