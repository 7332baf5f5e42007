// lbbnn_layers_prepare -- the x-independent part of a network forward for all layers in three
// launches on one stream: K3 (flows of every MNF layer), K1 (one grid over the rows of every
// layer), K5 (KL finalize of every layer).  See include/lbbnn.h.
#include <cstdlib>
#include "lbbnn_internal.h"

using namespace lbbnn;

static int layers_prepare_impl(const lbbnn_layer_desc_t* L, int n, const uint64_t* rng, void* stream, bool with_k5,
                               uint64_t* rng_live = nullptr, uint64_t* rng_snap = nullptr, uint64_t advance = 0,
                               const FormatJob* fmt = nullptr) {
    if (!L) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS) return LBBNN_E_SHAPE;
    FlowArgs fa[LBBNN_MAX_LAYERS];
    WeightPassArgs wa[LBBNN_MAX_LAYERS];
    FinalizeArgs ka[LBBNN_MAX_LAYERS];
    int flow_of[LBBNN_MAX_LAYERS];               // index into fa of layer i's planar flows, or -1
    int nf = 0, nk = 0;
    for (int i = 0; i < n; ++i) {
        const lbbnn_layer_desc_t& d = L[i];
        const bool mnf = d.q0_mean != nullptr;
        flow_of[i] = -1;
        if (!d.weight_mu || !d.weight_rho || !d.lambdal || !d.bias_mu || !d.bias_rho || !d.e_w || !d.bias_var) return LBBNN_E_NULL;
        if (d.stochastic && !d.var_w) return LBBNN_E_NULL;
        if (d.want_kl && (!d.kl_rows || !d.kl_layer)) return LBBNN_E_NULL;
        if (mnf && d.flows_done) {
            if (!d.z_fwd || (d.want_kl && (!d.z_kl || !d.scal || !d.r0_c || !d.r0_b1 || !d.r0_b2 || !d.act_mu || !d.act_var))) return LBBNN_E_NULL;
        } else if (mnf) {
            if (!d.q0_log_var || !d.z_fwd) return LBBNN_E_NULL;
            if (d.I > LBBNN_MAX_FLOW_DIM || d.z_flow.T < 0 || d.z_flow.T > LBBNN_MAX_FLOW_T ||
                d.r_flow.T < 0 || d.r_flow.T > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
            if (d.want_kl && (!d.z_kl || !d.scal || !d.r0_c || !d.r0_b1 || !d.r0_b2 || !d.act_mu || !d.act_var)) return LBBNN_E_NULL;
            if ((!d.eps_z || (d.want_kl && (!d.eps_z2 || !d.eps_act))) && !rng) return LBBNN_E_NOISE;
            flow_of[i] = nf;
            FlowArgs& f = fa[nf++];
            f.q0_mean = d.q0_mean; f.q0_log_var = d.q0_log_var; f.eps_fwd = d.eps_z; f.eps_kl = d.eps_z2; f.rng = rng;
            f.z_fwd = d.z_fwd; f.z_kl = d.z_kl; f.scal = d.scal; f.zf = d.z_flow; f.rf = d.r_flow;
            if (!d.want_kl) f.rf.T = 0;
            for (int t = 0; t < f.zf.T; ++t) if (!f.zf.u[t] || !f.zf.w[t] || !f.zf.b[t]) return LBBNN_E_NULL;
            for (int t = 0; t < f.rf.T; ++t) if (!f.rf.u[t] || !f.rf.w[t] || !f.rf.b[t]) return LBBNN_E_NULL;
            f.I = d.I; f.want_kl = d.want_kl; f.layer = d.layer_id & 63u;
        }
        const int rc = make_weight_pass_args(wa[i], d.weight_mu, d.weight_rho, d.lambdal,
                                             mnf ? d.z_fwd : nullptr, (mnf && d.want_kl) ? d.z_kl : nullptr,
                                             (mnf && d.want_kl) ? d.r0_c : nullptr, d.bias_rho, &d.priors,
                                             d.e_w, d.stochastic ? d.var_w : nullptr, lbbnn_operand_ld(d.I),
                                             d.want_kl ? d.kl_rows : nullptr,
                                             (mnf && d.want_kl) ? d.act_mu : nullptr, (mnf && d.want_kl) ? d.act_var : nullptr,
                                             d.bias_var, d.O, d.I, d.split, d.split >= 2 ? d.e_scale : nullptr,
                                             d.split >= 2 ? d.v_scale : nullptr);
        if (rc) return rc;
        if (d.want_kl) {
            FinalizeArgs& k = ka[nk++];
            k.kl_rows = d.kl_rows; k.bias_mu = d.bias_mu; k.bias_rho = d.bias_rho;
            k.act_mu = mnf ? d.act_mu : nullptr; k.act_var = mnf ? d.act_var : nullptr; k.eps_act = d.eps_act;
            k.r0_b1 = d.r0_b1; k.r0_b2 = d.r0_b2; k.scal = mnf ? d.scal : nullptr; k.rng = rng;
            k.kl_out = nullptr; k.kl_layer = d.kl_layer; k.O = d.O; k.I = d.I; k.accum = 0; k.layer = d.layer_id & 63u;
            k.bias_mu_prior = d.priors.bias_mu_prior; k.bias_sigma_prior = d.priors.bias_sigma_prior;
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = 0;
    // Planar flows: EITHER the K3 launch ahead of the weight pass (default), OR computed inside the weight pass's own
    // workgroups (LBBNN_K1_INFLOW=1 in the environment, read once; every such layer must qualify: <= 4 transforms, rows the
    // row kernel takes, and this call must not advance the live RNG offset, which the workgroups read).  Measured on the
    // headline net (profiles/r02_k1_ab.txt): in-kernel 33.6 us for the one launch against 13.0 (K3) + 21.6 (K1) for two --
    // the 13 us are the chain's dependent latency, which every workgroup then pays before its row, so one launch boundary
    // is all that is saved; kept as the opt-in form (one launch fewer for small, launch-bound networks).
    InFlow inf[LBBNN_MAX_LAYERS] = {};
    static const bool inflow_allowed = [] { const char* e = getenv("LBBNN_K1_INFLOW"); return e && e[0] == '1'; }();
    bool in_kernel = nf > 0 && advance == 0 && inflow_allowed;
    for (int i = 0; i < n && in_kernel; ++i) {
        in_kernel = wa[i].vec && wa[i].ld <= 2048;
        if (in_kernel && flow_of[i] >= 0) in_kernel = in_flow_eligible(fa[flow_of[i]], wa[i]);
    }
    if (in_kernel) {
        for (int i = 0; i < n; ++i) if (flow_of[i] >= 0) make_in_flow(inf[i], fa[flow_of[i]]);
    } else if (nf) {
        bool fmt_done = false;
        rc = launch_flow_planar(fa, nf, s, 1, 0, 0, fmt, &fmt_done); if (rc) return rc;
        if (fmt_done) fmt = nullptr;
    }
    if (fmt) {          // no flow launch to ride in (LRT layers, in-kernel flows, shapes the register kernel does not take)
        rc = lbbnn_format_x(fmt->x, fmt->ldx, fmt->planes, fmt->ldp, fmt->B, fmt->I, stream); if (rc) return rc;
    }
    rc = launch_weight_pass(wa, n, s, rng_live, rng_snap, advance, in_kernel ? inf : nullptr); if (rc) return rc;
    if (nk && with_k5) { rc = launch_kl_finalize(ka, nk, s); if (rc) return rc; }
    return 0;
}

extern "C" int lbbnn_layers_prepare(const lbbnn_layer_desc_t* L, int n, const uint64_t* rng, void* stream) {
    return layers_prepare_impl(L, n, rng, stream, true);
}

extern "C" int lbbnn_layers_operands(const lbbnn_layer_desc_t* L, int n, const uint64_t* rng, void* stream) {
    return layers_prepare_impl(L, n, rng, stream, false);
}

extern "C" int lbbnn_layers_operands_snap(const lbbnn_layer_desc_t* L, int n, uint64_t* rng, uint64_t* rng_snap,
                                          uint64_t advance, void* stream) {
    if (rng && !rng_snap) return LBBNN_E_NULL;
    return layers_prepare_impl(L, n, rng, stream, false, rng, rng_snap, advance);
}

extern "C" int lbbnn_layers_operands_x(const lbbnn_layer_desc_t* L, int n, uint64_t* rng, uint64_t* rng_snap, uint64_t advance,
                                       const float* x, int ldx, void* planes, int ldp, int B, int I, void* stream) {
    if (rng && !rng_snap) return LBBNN_E_NULL;
    if (!x || !planes) return LBBNN_E_NULL;
    if (B <= 0 || I <= 0 || ldx < I || ldp < I) return LBBNN_E_SHAPE;
    if ((I & 7) || (ldx & 3) || (ldp & 31) || (reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(planes) & 15u))
        return LBBNN_E_ALIGN;
    const FormatJob j{x, static_cast<char*>(planes), ldx, ldp, B, I};
    return layers_prepare_impl(L, n, rng, stream, false, rng, rng_snap, advance, &j);
}

extern "C" int lbbnn_ensemble_operands(const lbbnn_layer_desc_t* L, int n, int members, const uint64_t* rng,
                                       uint64_t member_advance, void* stream) {
    if (!L) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS || members < 1 || members > 65535) return LBBNN_E_SHAPE;
    FlowArgs fa[LBBNN_MAX_LAYERS];
    WeightPassArgs wa[LBBNN_MAX_LAYERS];
    int nf = 0;
    long long z_ms = -1;
    for (int i = 0; i < n; ++i) {
        const lbbnn_layer_desc_t& d = L[i];
        const bool mnf = d.q0_mean != nullptr;
        if (!d.weight_mu || !d.weight_rho || !d.lambdal || !d.bias_mu || !d.bias_rho || !d.e_w || !d.var_w || !d.bias_var)
            return LBBNN_E_NULL;
        if (!d.stochastic || d.want_kl || d.flows_done || d.eps_z) return LBBNN_E_FLAGS;
        const int ld = lbbnn_operand_ld(d.I);
        if (mnf) {
            if (!d.q0_log_var || !d.z_fwd) return LBBNN_E_NULL;
            if (!rng) return LBBNN_E_NOISE;
            if (d.z_flow.T < 0 || d.z_flow.T > LBBNN_MAX_FLOW_T) return LBBNN_E_SHAPE;
            // one launch serves all layers: the member stride of z must be the same for all of them -- the caller lays every
            // layer's z block out with the stride of the WIDEST layer (checked here through equal strides ld_max)
            FlowArgs& f = fa[nf++];
            f = FlowArgs{};
            f.q0_mean = d.q0_mean; f.q0_log_var = d.q0_log_var; f.rng = rng; f.z_fwd = d.z_fwd; f.zf = d.z_flow; f.rf.T = 0;
            for (int t = 0; t < f.zf.T; ++t) if (!f.zf.u[t] || !f.zf.w[t] || !f.zf.b[t]) return LBBNN_E_NULL;
            f.I = d.I; f.want_kl = 0; f.layer = d.layer_id & 63u;
            if (z_ms == -1) z_ms = ld;
            else if (z_ms != ld) z_ms = -2;                            // (-2 sticks: layers of different widths)
        }
        const int rc = make_weight_pass_args(wa[i], d.weight_mu, d.weight_rho, d.lambdal, mnf ? d.z_fwd : nullptr, nullptr,
                                             nullptr, d.bias_rho, &d.priors, d.e_w, d.var_w, ld, nullptr, nullptr, nullptr,
                                             d.bias_var, d.O, d.I, d.split == 1 ? 1 : 0);
        if (d.split >= 2) return LBBNN_E_FLAGS;
        if (rc) return rc;
        if (!wa[i].vec || wa[i].ld > 2048) return LBBNN_E_ALIGN;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nf) {
        // layers of different widths: one K3 launch per distinct stride would be needed -- launch per layer instead (n <= 4)
        if (z_ms == -2) {
            for (int k = 0; k < nf; ++k) {
                const int rc = launch_flow_planar(&fa[k], 1, s, members, member_advance, lbbnn_operand_ld(fa[k].I));
                if (rc) return rc;
            }
        } else {
            const int rc = launch_flow_planar(fa, nf, s, members, member_advance, z_ms);
            if (rc) return rc;
        }
    }
    return launch_weight_pass(wa, n, s, nullptr, nullptr, 0, nullptr, members);
}

int lbbnn::fill_finalize_args(const lbbnn_layer_desc_t* L, int n, const uint64_t* rng, FinalizeArgs* ka, int* active) {
    if (!L) return LBBNN_E_NULL;
    if (n <= 0 || n > LBBNN_MAX_LAYERS) return LBBNN_E_SHAPE;
    for (int i = 0; i < n; ++i) {
        const lbbnn_layer_desc_t& d = L[i];
        const bool mnf = d.q0_mean != nullptr;
        active[i] = d.want_kl ? 1 : 0;
        FinalizeArgs& k = ka[i];
        k = FinalizeArgs{};
        if (!d.want_kl) continue;
        if (!d.kl_rows || !d.kl_layer || !d.bias_mu || !d.bias_rho) return LBBNN_E_NULL;
        if (mnf && (!d.scal || !d.r0_b1 || !d.r0_b2 || !d.act_mu || !d.act_var)) return LBBNN_E_NULL;
        if (mnf && !d.eps_act && !rng) return LBBNN_E_NOISE;
        k.kl_rows = d.kl_rows; k.bias_mu = d.bias_mu; k.bias_rho = d.bias_rho;
        k.act_mu = mnf ? d.act_mu : nullptr; k.act_var = mnf ? d.act_var : nullptr; k.eps_act = d.eps_act;
        k.r0_b1 = d.r0_b1; k.r0_b2 = d.r0_b2; k.scal = mnf ? d.scal : nullptr; k.rng = rng;
        k.kl_out = nullptr; k.kl_layer = d.kl_layer; k.O = d.O; k.I = d.I; k.accum = 0; k.layer = d.layer_id & 63u;
        k.bias_mu_prior = d.priors.bias_mu_prior; k.bias_sigma_prior = d.priors.bias_sigma_prior;
    }
    return 0;
}

extern "C" int lbbnn_layers_finalize(const lbbnn_layer_desc_t* L, int n, uint64_t* rng, uint64_t advance, float* kl_total,
                                     void* stream) {
    FinalizeArgs ka[LBBNN_MAX_LAYERS];
    int active[LBBNN_MAX_LAYERS];
    if (const int rc = fill_finalize_args(L, n, rng, ka, active)) return rc;
    if (kl_total) for (int i = 0; i < n; ++i) if (!active[i]) return LBBNN_E_NULL;   // a total needs every layer's KL
    return launch_kl_finalize_all(ka, active, n, rng, advance, kl_total, static_cast<hipStream_t>(stream));
}
