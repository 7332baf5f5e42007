// Internal (non-ABI) kernel argument structs and launchers shared by the translation units.
// Every kernel here is batched over up to LBBNN_MAX_LAYERS layers; the single-layer C entry points
// are n = 1 calls of the same launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lbbnn.h"
#include "lbbnn_device.h"

namespace lbbnn {

#define LBBNN_HIDDEN __attribute__((visibility("hidden")))

struct WeightPassArgs {
    const float* mu; const float* rho; const float* lambdal;
    const float* z_fwd; const float* z_kl; const float* r0_c; const float* bias_rho;
    float* e_w; float* var_w;
    float* kl_rows; float* act_mu; float* act_var; float* bias_var;
    int O, I, ld, vec;
    int split;      // 0: fp32 [O][ld]; 1: bf16 hi | lo units (lbbnn_device.h); 2: fp16 hi | lo units of the row-scaled values;
                    // 3: as 2 for e_w, var_w as plain fp16 rows of ld halves (the hi part alone: LBBNN_F_VAR1 operands)
    float* e_scale; float* v_scale;      // split == 2: (O) inverse row scales written by the row kernel
    float mu_prior, sigma_prior, alpha_prior;
    float log_sp, log_ap, log_1map, inv_2sp2;     // host-precomputed prior constants
};

struct FlowArgs {
    const float* q0_mean; const float* q0_log_var;
    const float* eps_fwd; const float* eps_kl;
    const uint64_t* rng;
    float* z_fwd; float* z_kl; float* scal;
    lbbnn_planar_flow_t zf, rf;
    int I; int want_kl; uint32_t layer;
};

// In-kernel planar flows of one layer for the row kernel of the weight pass (weight_pass.hip): Tz + Tr <= 4 transforms,
// z flow first.  on == 0: the layer's z vectors (if any) come from memory (WeightPassArgs::z_fwd / z_kl).
struct InFlow {
    const float* q0_mean; const float* q0_log_var; const float* eps_fwd; const float* eps_kl;
    const uint64_t* rng;
    float* z_fwd; float* z_kl; float* scal;
    const float* u[4]; const float* w[4]; const float* b[4];
    int Tz, Tr, want_kl, on;
    uint32_t layer;
};

struct FinalizeArgs {
    const float* kl_rows; const float* bias_mu; const float* bias_rho;
    const float* act_mu; const float* act_var; const float* eps_act;
    const float* r0_b1; const float* r0_b2; const float* scal;
    const uint64_t* rng;
    float* kl_out; float* kl_layer;
    int O, I, accum; uint32_t layer;
    float bias_mu_prior, bias_sigma_prior;
};

// K5 of up to LBBNN_MAX_LAYERS layers + their total, as carried by a GEMM launch (lbbnn_lrt_gemm_finalize)
struct FinalizePiggy { FinalizeArgs l[LBBNN_MAX_LAYERS]; int active[LBBNN_MAX_LAYERS]; int n; float* total;
                       uint64_t* rng_adv; uint64_t adv; };     // rng_adv != NULL: the piggy workgroup also does rng_adv[1] += adv

// Select element `idx` of a by-value kernel-argument array WITHOUT dynamic indexing: a runtime index into a
// kernarg struct array makes hipcc copy the array to scratch memory (measured: 64 B/lane of scratch in K1);
// a chain of uniform compares keeps every field a constant-offset scalar load.
#define LBBNN_SELECT_LAYER(dst, arr, idx)                      \
    do {                                                       \
        if ((idx) == 0) dst = (arr)[0];                        \
        else if ((idx) == 1) dst = (arr)[1];                   \
        else if ((idx) == 2) dst = (arr)[2];                   \
        else dst = (arr)[3];                                   \
    } while (0)
static_assert(LBBNN_MAX_LAYERS == 4, "LBBNN_SELECT_LAYER enumerates 4 layers");

// Fill / validate helpers (return LBBNN_E_* or 0); launchers return hipGetLastError().
LBBNN_HIDDEN int make_weight_pass_args(WeightPassArgs& a, const float* mu, const float* rho, const float* lambdal,
                                       const float* z_fwd, const float* z_kl, const float* r0_c, const float* bias_rho,
                                       const lbbnn_priors_t* priors, void* e_w, void* var_w, int ld,
                                       float* kl_rows, float* act_mu, float* act_var, float* bias_var, int O, int I,
                                       int split = 0, float* e_scale = nullptr, float* v_scale = nullptr);
LBBNN_HIDDEN int launch_weight_pass(const WeightPassArgs* a, int n, hipStream_t s, uint64_t* rng = nullptr,
                                    uint64_t* rng_snap = nullptr, uint64_t advance = 0, const InFlow* flows = nullptr,
                                    int members = 1);
// planar flows of a layer computed inside the weight pass's workgroups instead of by launch_flow_planar (advance must be 0)
LBBNN_HIDDEN bool in_flow_eligible(const FlowArgs& f, const WeightPassArgs& w);
LBBNN_HIDDEN void make_in_flow(InFlow& o, const FlowArgs& f);
// fmt (optional): an lbbnn_format_x job for the launch to carry on otherwise idle CUs; *fmt_done says whether it did
LBBNN_HIDDEN int launch_flow_planar(const FlowArgs* a, int n, hipStream_t s, int members = 1, unsigned long long m_adv = 0,
                                    long long z_ms = 0, const struct FormatJob* fmt = nullptr, bool* fmt_done = nullptr);
LBBNN_HIDDEN int launch_kl_finalize_all(const FinalizeArgs* a, const int* active, int n, uint64_t* rng, uint64_t advance,
                                        float* kl_total, hipStream_t s);
LBBNN_HIDDEN int launch_kl_finalize(const FinalizeArgs* a, int n, hipStream_t s);
// FinalizeArgs of every layer of a network from its descriptors (validation as lbbnn_layers_finalize)
LBBNN_HIDDEN int fill_finalize_args(const lbbnn_layer_desc_t* L, int n, const uint64_t* rng, FinalizeArgs* ka, int* active);

}  // namespace lbbnn
