/*
 * lbbnn.h -- C ABI of the MI355X (gfx950) Bayesian linear-layer hot path.
 *
 * Drop-in boundary for LarsELund/Bayesian-Neural-Nets' BayesianLinear.forward family.
 * The reference has no native layer: its "FFI" is the set of PyTorch aten calls the layer
 * issues.  Each entry point below replaces one group of those calls and
 * cites the reference lines it stands for (paths relative to the reference checkout).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer to contiguous row-major fp32 unless stated otherwise.
 *    The caller (PyTorch) owns all buffers, including workspaces; nothing here allocates.
 *  - `stream` is a hipStream_t passed as void*.  Calls only enqueue work and return; they
 *    never synchronise and are safe under HIP-graph capture.  The only process-wide state is an idempotent cache
 *    of "dynamic-LDS limit already raised" per kernel (hipFuncSetAttribute is a host-side attribute, not a stream
 *    operation); a few environment variables read once select measurement knobs (LBBNN_GEMM_RESIDENCY, LBBNN_GEMM_NO_DMA).
 *  - Return value: 0 on success; a negative LBBNN_E_* for argument errors (nothing was
 *    launched); a positive hipError_t if the launch failed.  No C++ exception crosses.
 *  - Noise: every stochastic entry takes either an explicit draw (`eps`, parity mode) or,
 *    when that pointer is NULL, generates N(0,1) in-kernel with Philox4x32-10 keyed by the
 *    two 64-bit words at `rng` = {seed, offset} (device memory, so a captured graph sees
 *    fresh noise on every replay after lbbnn_rng_advance).
 */
#ifndef LBBNN_H_
#define LBBNN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBBNN_ABI_VERSION 1

#define LBBNN_OK 0
#define LBBNN_E_NULL (-1)      /* a required pointer is NULL                    */
#define LBBNN_E_SHAPE (-2)     /* a dimension is <= 0 or exceeds a kernel limit */
#define LBBNN_E_ALIGN (-3)     /* a pointer / leading dimension is misaligned   */
#define LBBNN_E_FLAGS (-4)     /* unknown or inconsistent flag bits             */
#define LBBNN_E_NOISE (-5)     /* neither an explicit draw nor an rng state     */

#define LBBNN_MAX_FLOW_T 16    /* max transforms per planar flow ('mixed' uses 10)  */
#define LBBNN_MAX_FLOW_DIM 16384

/* GEMM flags */
#define LBBNN_F_RELU 0x1       /* fuse F.relu on the output (LBBNN-GP-MF-LRT.py:208-209)      */
#define LBBNN_F_MEAN_ONLY 0x2  /* posterior-mean branch: out = x.e_w^T + b (…LRT.py:178-180)  */
#define LBBNN_F_SPLIT16 0x4    /* split-precision MFMA path: bf16x3 products, operands hi + lo  */
#define LBBNN_F_LOG_SOFTMAX 0x8 /* fuse F.log_softmax(dim=1) on the output; O <= 16 only (…LRT.py:210) */
#define LBBNN_F_HALF16 0x20    /* with LBBNN_F_SPLIT16 | LBBNN_F_SINGLE16: the single product in FP16 (v_mfma_f32_16x16x32_f16; operands
                                  from lbbnn_vd_operands(LBBNN_F_HALF16): fp16 in the hi units) -- the "fp16 MFMA" of BASELINE
                                  configs[4]; unscaled, so only for operands inside fp16's range (the variational-dropout theta) */
#define LBBNN_F_SINGLE16 0x10  /* with LBBNN_F_SPLIT16, lbbnn_lrt_gemm only: ONE bf16 product per moment (the hi parts of the
                                  same operands, x rounded to bf16 in registers) -- the plain "bf16 MFMA" arithmetic that
                                  BASELINE configs[1] names; 2e-3 relative on the mean GEMM, outside the 1e-4 contract */

/* Philox stream ids (third counter word) -- one per kind of draw, per layer (stream = kind*64+layer) */
#define LBBNN_STREAM_EPS_OUT 0
#define LBBNN_STREAM_EPS_Z 1
#define LBBNN_STREAM_EPS_Z2 2
#define LBBNN_STREAM_EPS_ACT 3
#define LBBNN_STREAM_MASK 6    /* Bernoulli(0.5) masks of the dense flows (lbbnn_dense_layer_t::draw_masks) */
#define LBBNN_STREAM_ROW_MASK 7 /* per-row masks of lbbnn_flow_dense_rows (stream id = 7*64 + layer id by convention)  */

/* Prior constants of one layer.  The reference keeps them as constant tensors:
 * LBBNN-GP-MF-LRT.py:142-143,151,159-160; LBBNN-GP-MF-MNF.py:145-146,154,162-163. */
typedef struct lbbnn_priors {
    float mu_prior;          /* 0    */
    float sigma_prior;       /* 1    */
    float alpha_prior;       /* 0.05 */
    float bias_mu_prior;     /* 0    */
    float bias_sigma_prior;  /* 1    */
} lbbnn_priors_t;

int lbbnn_abi_version(void);
const char* lbbnn_error_string(int code);

/* Round a K extent up to the operand leading dimension the GEMM expects (multiple of 32). */
int lbbnn_operand_ld(int I);

/* ---------------------------------------------------------------------------------------------
 * K1  lbbnn_weight_pass -- one fused, coalesced pass over the (O,I) variational parameters.
 *
 * Replaces, per layer and per forward:
 *   alpha = 1/(1+exp(-lambdal))                       LBBNN-GP-MF-LRT.py:167, …MNF.py:191
 *   sigma = log1p(exp(rho))  (recomputed 3-4x there)  …LRT.py:81-82
 *   e_w = mu*alpha ; var_w = sigma^2*alpha^2           …LRT.py:170-171, …MNF.py:195-196
 *   kl_weight integrand + row sums                     …LRT.py:189-192, …MNF.py:230-233
 *   act_mu = r0_c @ (z2*mu*alpha)^T, act_var = r0_c^2 @ var_w^T     …MNF.py:211-212,216-217
 *   bias sigma^2                                       …LRT.py:173
 *
 * Inputs : mu, rho, lambdal (O,I).  z_fwd (I) or NULL: per-input multiplier z_k folded into
 *          the mean operand (…MNF.py:197).  z_kl (I) or NULL: z2 of the KL branch (…MNF.py:210).
 *          r0_c (I) or NULL.  bias_rho (O) or NULL.
 * Outputs: e_w, var_w: GEMM operands [O][ld] (ld = lbbnn_operand_ld(I), zero-filled tail),
 *          either may be NULL.  kl_rows (O): per-row sum of the KL integrand (NULL = skip).
 *          act_mu, act_var (O) (NULL = skip; need z_kl and r0_c).  bias_var (O) = softplus(bias_rho)^2.
 *          With LBBNN_F_SPLIT16 each operand row (the same 4*ld bytes) holds w = hi + lo in bf16: per 32-k chunk one
 *          128-B line of 16-B units, unit 2g = hi of k in [8g, 8g+8), unit 2g+1 = lo of the same k; see DESIGN.md
 *          "Data layout".  The format is produced and consumed only by this library.
 */
int lbbnn_weight_pass(const float* mu, const float* rho, const float* lambdal,
                      const float* z_fwd, const float* z_kl, const float* r0_c,
                      const float* bias_rho, const lbbnn_priors_t* priors,
                      void* e_w, void* var_w, int ld,
                      float* kl_rows, float* act_mu, float* act_var, float* bias_var,
                      int O, int I, int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2  lbbnn_lrt_gemm -- the two activation-moment GEMMs + local-reparameterisation epilogue.
 *
 *   mean = x . e_w^T + bias_mean          torch.mm  …LRT.py:172, …MNF.py:197
 *   var  = x^2 . var_w^T (* var_scale) + bias_var      …LRT.py:173, …MNF.py:198  (x^2 formed in registers)
 *   out  = mean + sqrt(var) * eps         randn + sample  …LRT.py:174-175, …MNF.py:199-200
 *   optional ReLU                         …LRT.py:208-209
 * LBBNN_F_MEAN_ONLY: out = mean (…LRT.py:178-180; F.linear of LBBNN-GP-MF.py:255).
 *
 * x (B,I) with row stride ldx floats; e_w/var_w operands from lbbnn_weight_pass (leading dim ld);
 * bias_mean/bias_var/var_scale (O) or NULL (0 / 0 / 1); eps (B,O) or NULL => Philox from `rng`
 * with counter = (row_offset + b, o/4), stream id `rng_stream`; out (B,O) row stride ldo.
  * B == 0 (empty batch) is a successful no-op.
 */
int lbbnn_lrt_gemm(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                   const float* bias_mean, const float* bias_var, const float* var_scale,
                   const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                   float* out, int ldo, int B, int I, int O, int flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3  lbbnn_mnf_flow_planar -- z sampling + planar flows of one MNF layer, one launch.
 *
 * Replaces sample_z (LBBNN-GP-MF-MNF.py:182-187) called twice per training forward, the planar
 * transforms (flows2.py:86-95, PropagateFlow.forward :41-46), log_q0 (…MNF.py:213-214) and
 * r_flow(z2) (…MNF.py:222).  Only the row the reference keeps (zs[-1], quirk 1 of SURVEY.md 3.2)
 * is computed.
 *
 *   z_fwd = z_flow(q0_mean + exp(q0_log_var)^.5 * eps_fwd)            (forward multiplier z_k)
 *   z0    = q0_mean + exp(q0_log_var)^.5 * eps_kl ; z_kl = z_flow(z0) (KL branch z2)
 *   scal[0] = log_det_q   scal[1] = log_q0 (uses -0.5*log(pi), quirk 3)
 *   scal[2] = log_det_r   scal[3] = r_flow(z_kl)[-1]  (last ELEMENT, quirk 2)
 *   scal[4] = log-det of the forward draw's z_flow (what sample_z(B) returns next to z_k)
 *
 * zu/zw/zb (ru/rw/rb): host arrays of T device pointers to the u (I), w (I), bias (1) of each
 * transform.  eps_fwd / eps_kl: (I) draws or NULL => Philox (streams EPS_Z / EPS_Z2, counter = i/4).
 * want_kl == 0 computes z_fwd (and scal[4] if scal != NULL) only.  z_fwd / z_kl (I) outputs; scal: 8 floats.
 */
int lbbnn_mnf_flow_planar(const float* q0_mean, const float* q0_log_var,
                          const float* const* zu, const float* const* zw, const float* const* zb, int Tz,
                          const float* const* ru, const float* const* rw, const float* const* rb, int Tr,
                          const float* eps_fwd, const float* eps_kl,
                          const uint64_t* rng, uint32_t layer_id,
                          float* z_fwd, float* z_kl, float* scal,
                          int I, int want_kl, void* stream);


/* ---------------------------------------------------------------------------------------------
 * K4  lbbnn_mnf_flow_dense -- z sampling + dense coupling flows (RNVP / MNF type) of one MNF layer.
 *
 * Same outputs as lbbnn_mnf_flow_planar (z_fwd, z_kl, scal[0..4]) for the reference's default flow
 * types: RNVP (flows2.py:188-219: mask, 4-layer LeakyReLU(0.1) MLP I->75->75->75->75, t/s heads,
 * sigmoid gate, Sum (1-m) log gate) and MNF (flows2.py:225-241: mask, tanh(f(m z)) 100 hidden units,
 * g/k heads, Sum (1-m) log sigma).  Only the kept row (zs[-1]) is computed, so every dense step is a
 * GEMV; per transform: one launch spreads the I-long dot products of the hidden units over
 * workgroups, one launch produces the I outputs (recomputing the tiny H x H chain per workgroup) and
 * per-workgroup log-det partials that are summed in a fixed order (deterministic).
 *
 * Bernoulli(0.5) masks (flows2.py:209,234) are explicit inputs (the host wrapper draws them):
 * mask_fwd = the forward draw's z_flow call, mask_kl = the KL branch's z_flow / r_flow call.
 * work: caller-owned scratch of lbbnn_flow_dense_workspace(I) floats.
 */
#define LBBNN_FLOW_RNVP 0
#define LBBNN_FLOW_MNF 1
#define LBBNN_MAX_HIDDEN 128

typedef struct lbbnn_dense_transform {
    int kind;                               /* LBBNN_FLOW_RNVP / LBBNN_FLOW_MNF                         */
    int hidden;                             /* 75 (RNVP) / 100 (MNF); <= LBBNN_MAX_HIDDEN               */
    const float *w_in, *b_in;               /* (H,I),(H): RNVP network.0 | MNF f                        */
    const float *w_mid[3], *b_mid[3];       /* (H,H),(H): RNVP network.2/4/6 | NULL for MNF             */
    const float *w_a, *b_a;                 /* (I,H),(I): RNVP t (shift) | MNF g (mu)                   */
    const float *w_b, *b_b;                 /* (I,H),(I): RNVP s (scale) | MNF k (sigma pre-activation) */
    const float *mask_fwd, *mask_kl;        /* (I) each, values in {0,1}                                */
} lbbnn_dense_transform_t;

int64_t lbbnn_flow_dense_workspace(int I);

int lbbnn_mnf_flow_dense(const float* q0_mean, const float* q0_log_var,
                         const lbbnn_dense_transform_t* zt, int Tz,
                         const lbbnn_dense_transform_t* rt, int Tr,
                         const float* eps_fwd, const float* eps_kl,
                         const uint64_t* rng, uint32_t layer_id,
                         float* z_fwd, float* z_kl, float* scal, float* work,
                         int I, int want_kl, void* stream);

/* The same for up to LBBNN_MAX_LAYERS layers at once (blockIdx.z = layer): 2 + 2*(Tz+Tr) launches for the whole network
 * instead of that many per layer.  The layers must agree on Tz, Tr and want_kl (LBBNN_E_SHAPE otherwise). */
typedef struct lbbnn_dense_layer {
    const float *q0_mean, *q0_log_var;
    const lbbnn_dense_transform_t *zt, *rt;
    const float *eps_fwd, *eps_kl;
    float *z_fwd, *z_kl, *scal, *work;          /* work: lbbnn_flow_dense_workspace(I) floats */
    int Tz, Tr, I, want_kl;
    uint32_t layer_id;
    float *save;                                 /* NULL, or lbbnn_flow_dense_save_size(I, Tz, Tr) floats: the forward keeps the
                                                    input of every transform and the hidden activations of the coupling MLPs
                                                    there for lbbnn_mnf_flow_dense_backward */
    int draw_masks;                              /* 1: the mask vectors of zt / rt are OUTPUTS of this call -- Bernoulli(0.5)
                                                    (flows2.py:209,234) from Philox stream LBBNN_STREAM_MASK of `rng`
                                                    (required), counter = row i: bit t of word 0 / 1 / 2 = transform t of the
                                                    z flow's forward call / its KL call / the r flow -- written by the first
                                                    launch; 0: they are inputs */
} lbbnn_dense_layer_t;

int64_t lbbnn_flow_dense_save_size(int I, int Tz, int Tr);
int lbbnn_layers_dense_flows(const lbbnn_dense_layer_t* layers, int n, const uint64_t* rng, void* stream);
/* The same launches in two parts, so that the part only the KL needs can leave the critical path of a forward
 * (LBBNN-GP-MF-MNF.py:190-200 needs z_k only; :208-235 need the rest): phase 1 = the draws and the z flow on both draws
 * (outputs z_fwd, z_kl), phase 2 = the r flow on the KL draw and the scalars (scal); phase 0 = both (the call above).
 * Phase 2 must follow phase 1 of the same call arguments (it continues in `work`); it may be enqueued on another stream
 * once phase 1 has completed there. */
int lbbnn_layers_dense_flows_phase(const lbbnn_dense_layer_t* layers, int n, const uint64_t* rng, int phase, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5  lbbnn_kl_finalize -- the O(O+I) tail of the KL and the final scalar.
 *
 *   kl_bias                                                    …LRT.py:185-186, …MNF.py:227-228
 *   act = tanh(act_mu + sqrt(act_var)*eps_act); mean_r, log_var_r; log_rb   …MNF.py:218-224
 *   kl = kl_bias + sum(kl_rows) + (-log_det_q + log_q0) - (log_det_r + log_rb)   …MNF.py:215,225,235
 * LRT (scal == NULL): kl = kl_bias + sum(kl_rows)               …LRT.py:194
 *
 * kl_accum: if nonzero, *kl_out += kl (BayesianNetwork.kl(), …LRT.py:213-214), else *kl_out = kl.
 */
int lbbnn_kl_finalize(const float* kl_rows, const float* bias_mu, const float* bias_rho, int O,
                      const float* act_mu, const float* act_var, const float* eps_act,
                      const float* r0_b1, const float* r0_b2, int I,
                      const float* scal, const lbbnn_priors_t* priors,
                      const uint64_t* rng, uint32_t layer_id,
                      float* kl_out, float* kl_layer, int kl_accum, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Batched "prepare": everything in a network forward that does NOT depend on the activations
 * (K3 flows, K1 weight pass, K5 KL finalize) for up to LBBNN_MAX_LAYERS layers in three launches
 * on one stream -- one launch per kernel kind covering all layers -- so the critical path of
 * BayesianNetwork.forward (LBBNN-GP-MF-MNF.py:252-257) is  prepare -> GEMM1 -> GEMM2 -> GEMM3.
 * Field meaning = the arguments of the single-layer entry points above.
 */
#define LBBNN_MAX_LAYERS 4

typedef struct lbbnn_planar_flow {
    const float* u[LBBNN_MAX_FLOW_T];
    const float* w[LBBNN_MAX_FLOW_T];
    const float* b[LBBNN_MAX_FLOW_T];
    int T;
} lbbnn_planar_flow_t;

typedef struct lbbnn_layer_desc {
    /* parameters (state_dict names of the reference) */
    const float *weight_mu, *weight_rho, *lambdal, *bias_mu, *bias_rho;
    const float *q0_mean, *q0_log_var, *r0_c, *r0_b1, *r0_b2;      /* all NULL for an LRT layer */
    lbbnn_planar_flow_t z_flow, r_flow;
    lbbnn_priors_t priors;
    int O, I;
    uint32_t layer_id;
    int stochastic;          /* produce var_w (training or sample)            */
    int want_kl;             /* training or calculate_log_probs               */
    int split;               /* operand format: 0 fp32; 1 bf16 hi + lo (LBBNN_F_SPLIT16); 2 fp16 hi + lo with per-row power-of-two
                                scales (LBBNN_F_F16S: needs e_scale / v_scale below, rows of at most 1280 weights); 3 the same
                                for e_w and var_w as plain fp16 rows, hi part only (the operands of LBBNN_F_F16S | LBBNN_F_VAR1) */
    /* explicit draws; NULL => Philox from rng */
    const float *eps_z, *eps_z2, *eps_act;
    /* caller-owned workspace */
    float *z_fwd, *z_kl, *scal;                  /* (I), (I), 8 floats; MNF only */
    void *e_w, *var_w;                           /* [O][lbbnn_operand_ld(I)]     */
    float *kl_rows, *act_mu, *act_var, *bias_var;/* (O) each                     */
    float* kl_layer;                             /* 1 float: this layer's KL     */
    int flows_done;          /* nonzero: z_fwd / z_kl / scal were already produced by the caller (lbbnn_layers_dense_flows,
                                lbbnn_flow_chain): lbbnn_layers_operands then runs K1 only for this layer */
    /* split == 2 (LBBNN_F_F16S operands): per-output-feature accumulator scales written by K1, (O) floats each, exact
       powers of two: e_w row o holds fp16 hi + lo of e_w[o,:] / e_scale[o], var_w row o of var_w[o,:] / v_scale[o] */
    float *e_scale, *v_scale;
} lbbnn_layer_desc_t;

int lbbnn_layers_prepare(const lbbnn_layer_desc_t* layers, int n, const uint64_t* rng, void* stream);
/* The same forward with K5 moved to the END (it feeds no GEMM):  lbbnn_layers_operands = K3 + K1 of all layers;
 * lbbnn_layers_finalize = K5 of every layer with want_kl, kl_total = sum of the layer KLs in layer order (NULL = no
 * total; needs want_kl on every layer otherwise) and rng offset += advance, all in ONE single-workgroup launch.
 * A network forward is then 6 launches: operands (2), three GEMMs, finalize. */
int lbbnn_layers_operands(const lbbnn_layer_desc_t* layers, int n, const uint64_t* rng, void* stream);
int lbbnn_layers_finalize(const lbbnn_layer_desc_t* layers, int n, uint64_t* rng, uint64_t advance, float* kl_total,
                          void* stream);

/* The same forward with two launches fewer (what BayesianNetwork.forward uses):
 *
 * lbbnn_layers_operands_snap = lbbnn_layers_operands, and the weight-pass launch also copies the live Philox state
 *   {seed, offset} to rng_snap (2 words) and adds `advance` to the live offset.  It follows the last kernel of the
 *   forward that reads the live state (the flow kernels), so every LATER kernel of this forward -- the GEMMs, the KL
 *   finalize -- must be given rng_snap as its `rng`.  rng == NULL: no RNG in use, nothing is copied.
 *
 * lbbnn_lrt_gemm_finalize = lbbnn_lrt_gemm_train (same arguments up to `flags`; std_out may be NULL) + the work of lbbnn_layers_finalize
 *   (every layer's KL tail from its descriptor, kl_layer outputs, *kl_total if non-NULL; no RNG advance) for the n
 *   layers, computed by ONE extra workgroup of the same launch while the tiles are computed: the KL tail depends on
 *   parameters only, so it needs no launch of its own after the last GEMM.  fin_rng: the state K5 draws eps_act from
 *   (rng_snap above).  When the GEMM kernel selected for this shape cannot host the extra workgroup (small-tile
 *   configuration, O <= 16, not enough dynamic LDS) the finalize runs as a launch of its own after the GEMM: same
 *   results either way.
 */
int lbbnn_layers_operands_snap(const lbbnn_layer_desc_t* layers, int n, uint64_t* rng, uint64_t* rng_snap,
                               uint64_t advance, void* stream);
int lbbnn_lrt_gemm_finalize(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                            const float* bias_mean, const float* bias_var, const float* var_scale,
                            const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                            float* out, int ldo, float* std_out, int B, int I, int O, int flags,
                            const lbbnn_layer_desc_t* layers, int n, const uint64_t* fin_rng, float* kl_total,
                            void* stream);

/* lbbnn_lrt_gemm_finalize_adv = lbbnn_lrt_gemm_finalize + `rng_live[1] += advance`, done by the same extra workgroup
 * (n == 0 allowed: nothing to finalize, the advance is then a tiny launch after the GEMM).  This is what lets
 * lbbnn_layers_operands_snap run with advance = 0 -- and only then do its workgroups compute the planar flows of the
 * layers THEMSELVES from the live {seed, offset} (no K3 launch ahead of the weight pass; weight_pass.hip): the advance
 * has to come from a later launch of the same forward, none of whose kernels reads the live offset (they are given
 * rng_snap). */
int lbbnn_lrt_gemm_finalize_adv(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                                const float* bias_mean, const float* bias_var, const float* var_scale,
                                const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                                float* out, int ldo, float* std_out, int B, int I, int O, int flags,
                                const lbbnn_layer_desc_t* layers, int n, const uint64_t* fin_rng, float* kl_total,
                                uint64_t* rng_live, uint64_t advance, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Ensemble evaluation (test_ensemble, LBBNN-GP-MF-MNF.py:286-294 / ...LRT.py:241-247: TEST_SAMPLES stochastic forwards of
 * the same batch): `members` forwards of a layer in ONE launch per kernel kind, member m bit-identical to the m-th of
 * `members` consecutive single forwards (whose shared Philox offset advances by member_advance each).
 *
 * lbbnn_ensemble_operands: K3 + K1 for all n layers and all members (2 launches): per layer, z_fwd points to
 *   [members][lbbnn_operand_ld(I)] floats (member m's z_k; MNF layers with planar flows of <= 4 transforms; ignored for
 *   LRT layers) and e_w to [members][O][ld] operands (member m's e_w = mu alpha z_m); var_w, bias_var are shared (one copy).
 *   Every layer must have stochastic = 1, want_kl = 0 (evaluation draws no KL, ...MNF.py:208), flows_done = 0, no explicit
 *   eps_z.  Member m draws from Philox offset rng[1] + m * member_advance.
 * lbbnn_lrt_gemm_members: the dual-moment GEMM of lbbnn_lrt_gemm for every member, gridDim.z = members:
 *   x + m * x_mstride (floats; 0 = the same input for every member, the first layer), e_w + m * w_mstride (floats),
 *   var_w shared, out + m * o_mstride; in-kernel noise only, at offset rng[1] + m * member_advance.
 * The caller advances rng by members * member_advance afterwards (lbbnn_rng_advance). */
int lbbnn_ensemble_operands(const lbbnn_layer_desc_t* layers, int n, int members, const uint64_t* rng,
                            uint64_t member_advance, void* stream);
int lbbnn_lrt_gemm_members(const float* x, int ldx, int64_t x_mstride, const void* e_w, int64_t w_mstride,
                           const void* var_w, int ld, const float* bias_mean, const float* bias_var,
                           const uint64_t* rng, uint32_t rng_stream, int64_t row_offset, uint64_t member_advance,
                           float* out, int ldo, int64_t o_mstride, int B, int I, int O, int flags, int members,
                           void* stream);

/* End of a network forward: *kl_total = sum_l *kl_layers[l] (fixed order; BayesianNetwork.kl(),
 * …LRT.py:213-214) and rng[1] += advance, one tiny launch.  kl_total may be NULL (n ignored). */
int lbbnn_forward_finish(uint64_t* rng, uint64_t advance, const float* const* kl_layers, int n,
                         float* kl_total, void* stream);


/* ---------------------------------------------------------------------------------------------
 * K6  lbbnn_gate_sample -- baseline LBBNN layer (explicit latent-binary gate x Gaussian weight sample).
 *
 * Replaces, for BayesianLinear.forward of LBBNN-GP-MF.py:228-255, ONE fused pass over (O,I):
 *   ws = mu + softplus(rho)*eps ; weight = cgamma*ws        Gaussian.rsample :85-87, :232-233   (mode 0)
 *   weight = cgamma*mu (mode 1, medimean :236-238) | alpha_attr*mu (mode 2, :240-242)
 *   bias = bias_mu + softplus(bias_rho)*eps_b  (mode 0) | bias_mu
 * and, when want_lp, the Monte-Carlo log-probabilities (:246-251):
 *   log_prior = GaussGamma(weight_a,weight_b).log_prob(weight,cgamma)        :140-151
 *             + GaussGamma(bias_a,bias_b).log_prob(bias,1) + BetaBinomial(pa,pb).log_prob(cgamma)  :162-173
 *   log_q     = Gaussian.full_log_prob(weight,cgamma) :99-101 + Bernoulli(gamma_alpha).log_prob(cgamma) :122-128
 *             + Gaussian.log_prob(bias) :89-92
 * The Gamma draws tau_w (1) / tau_b (O) of :141 are inputs (drawn by the host wrapper with torch).
 * exact bits (the reference's `.exact` switches, :559-627): 1 weight_prior, 2 bias_prior, 4 gamma_prior, 8 gamma.
 * Outputs: w_out = GEMM operand [O][ld] (fp32, or the split bf16 hi|lo layout with LBBNN_F_SPLIT16) for
 *          lbbnn_lrt_gemm(LBBNN_F_MEAN_ONLY) = F.linear (:255); bias_out (O); log_prior, log_q (1 float each);
 *          rows: workspace of 4*O floats.  eps_w (O,I) / eps_b (O) NULL => Philox (streams EPS_W / EPS_B).
 */
#define LBBNN_STREAM_EPS_W 4
#define LBBNN_STREAM_EPS_B 5
#define LBBNN_MODE_SAMPLE 0
#define LBBNN_MODE_MEDIMEAN 1
#define LBBNN_MODE_MEAN 2

typedef struct lbbnn_gate_args {
    const float *mu, *rho, *gamma_alpha, *cgamma, *eps_w, *alpha_attr;       /* (O,I) */
    const float *bias_mu, *bias_rho, *eps_b, *bias_a, *bias_b, *tau_b;       /* (O)   */
    const float *weight_a, *weight_b, *tau_w, *pa, *pb;                      /* (1)   */
    void* w_out; float* bias_out; float* rows; float* log_prior; float* log_q;
    int O, I, ld, mode, exact, want_lp, flags;
    uint32_t layer_id;
} lbbnn_gate_args_t;

int lbbnn_gate_sample(const lbbnn_gate_args_t* args, const uint64_t* rng, void* stream);

/* K6b  lbbnn_gate_backward -- backward of the SAMPLED baseline layer (mode LBBNN_MODE_SAMPLE with log-probabilities: what
 * net.sample_elbo(...)[0].backward() differentiates, LBBNN-GP-MF.py:331-337).  Inputs: the forward's arguments (eps_w /
 * eps_b explicit, or NULL + the forward's rng state: same Philox streams), dW (O,I) = G^T x of F.linear (NULL = 0),
 * g_sum (O) = column sums of G (NULL = 0), and the device scalars g_lp = dL/dlog_prior, g_lq = dL/dlog_q (NULL = 0).
 * Outputs: d_mu, d_rho, d_cgamma, d_alpha (O,I) [d_alpha: through Bernoulli.log_prob's alpha, :125-127]; d_bias_mu,
 * d_bias_rho, d_bias_a, d_bias_b, d_tau_b (O); d_scalars[5] = d weight_a, d weight_b, d tau_w, d pa, d pb; w_out (O,I,
 * nullable): the sampled weight as a dense fp32 matrix (the operand of dX = G W).  rows: 3*O floats of workspace.
 * `exact` bits as in the forward (a rounded, detached gate passes no gradient).  Two launches. */
typedef struct lbbnn_gate_bwd_args {
    const float *mu, *rho, *gamma_alpha, *cgamma, *eps_w;                    /* (O,I) */
    const float *bias_mu, *bias_rho, *eps_b, *bias_a, *bias_b, *tau_b;       /* (O)   */
    const float *weight_a, *weight_b, *tau_w, *pa, *pb;                      /* (1)   */
    const float *dW, *g_sum, *g_lp, *g_lq;
    float *d_mu, *d_rho, *d_cgamma, *d_alpha, *w_out;
    float *d_bias_mu, *d_bias_rho, *d_bias_a, *d_bias_b, *d_tau_b, *d_scalars, *rows;
    int O, I, exact;
    uint32_t layer_id;
} lbbnn_gate_bwd_args_t;

int lbbnn_gate_backward(const lbbnn_gate_bwd_args_t* args, const uint64_t* rng, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K7  lbbnn_vd_operands -- Gaussian variational-dropout layer (variational_dropout.py:55-68).
 *   phi = x.theta ; delta = (x^2).(theta^2) * alpha ; out = phi + sqrt(delta)*zeta        :64-67
 * theta is (I,O) row-major (NN layout).  This pass writes the GEMM operands theta^T and (theta^2)^T as
 * [O][ld] (fp32 or split planes) through a tiled LDS transpose, so theta^2 is never materialised in (I,O)
 * form and the same lbbnn_lrt_gemm runs with var_scale = alpha, bias = NULL.
 */
int lbbnn_vd_operands(const float* theta, void* e_w, void* var_w, int ld, int I, int O, int flags, void* stream);

/* Backward operands of a Bayesian layer from its parameters in one pass: e_t = (weight_mu * sigmoid(lambdal) * z)^T and
 * v_t = (softplus(weight_rho)^2 sigmoid(lambdal)^2)^T as GEMM operands [I][ld] with ld = lbbnn_operand_ld(O), for the
 * input-gradient products dX = G_m . W_m + 2 x (.) (G_v . W_v) (contraction over O).  z (I) NULL = LRT layer; v_t NULL =
 * posterior-mean forward.  flags: LBBNN_F_SPLIT16 as elsewhere.  Replaces lbbnn_weight_pass + 2 x lbbnn_transpose_operand. */
int lbbnn_weight_operands_t(const float* weight_mu, const float* weight_rho, const float* lambdal, const float* z,
                            void* e_t, void* v_t, int ld, int O, int I, int flags, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Training support (the backward of loss.backward(), LBBNN-GP-MF-LRT.py:225).
 *
 * lbbnn_lrt_gemm_train = lbbnn_lrt_gemm that additionally stores std_out[b,o] = sqrt(var[b,o])
 * (pre-ReLU), which the backward needs for  dL/dvar = g * eps / (2 * std).
 *
 * lbbnn_transpose_operand: dst[c][r] = src[r][c] (squared when `square`) for src (R,C) with row stride
 * lds_src, written as a GEMM operand [C][ld] (ld = lbbnn_operand_ld(R), zero tail; fp32 or, with
 * LBBNN_F_SPLIT16, the split bf16 hi|lo layout).  With it every backward product is an "x . operand^T" GEMM on the
 * same kernels as the forward:
 *   dX  = G_m . W_m + 2 x (.) (G_v . W_v)      operands W_m^T, W_v^T          (K = O)
 *   dW_m = G_m^T . x ,  dW_v = G_v^T . x^2     operands x^T, (x^2)^T          (K = B)
 */
int lbbnn_lrt_gemm_train(const float* x, int ldx, const void* e_w, const void* var_w, int ld,
                         const float* bias_mean, const float* bias_var, const float* var_scale,
                         const float* eps, const uint64_t* rng, uint32_t rng_stream, int64_t row_offset,
                         float* out, int ldo, float* std_out, int B, int I, int O, int flags, void* stream);

/* lbbnn_format_operand: dst[r][c] = src[r][c] (squared when `square`) as a GEMM operand [R][ld] (ld = lbbnn_operand_ld(C),
 * zero tail, fp32 or LBBNN_F_SPLIT16) -- no transpose: the matrix already has the contraction index contiguous.  Used by
 * the variational-dropout backward: theta (n,m) and theta^2 are the operands of dX = G . theta^T + 2 x (.) (G_v . (theta^2)^T). */
int lbbnn_format_operand(const float* src, int R, int C, int lds_src, void* dst, int ld, int square, int flags, void* stream);
int lbbnn_transpose_operand(const float* src, int R, int C, int lds_src, void* dst, int ld,
                            int square, int flags, void* stream);

/* lbbnn_weight_pass_backward (K1b): analytic backward of the fused weight pass -- one pass over (O,I).
 * Given the operand gradients from the backward GEMMs (dWm wrt e_w = mu*alpha*z_fwd, dWv wrt var_w =
 * sigma^2*alpha^2), the upstream KL gradient *g_kl (device scalar, NULL = none) and the gradients of the
 * auxiliary activations da_mu / da_var (O) (wrt act_mu / act_var, already scaled by g_kl; NULL = none), writes
 *   dmu, drho, dlambdal (O,I)              chain through alpha = sigmoid(lambda), sigma = softplus(rho)
 *   dz_fwd (I) = sum_o dWm*mu*alpha        dz_kl (I), dr0_c (I): column sums of the KL / auxiliary terms
 * Column sums are deterministic: per-row-block partials in `work` (lbbnn_weight_pass_backward_workspace floats)
 * reduced in a fixed order by a second tiny launch.  Formulas: derivative of LBBNN-GP-MF-MNF.py:195-196,211-217,230-233.
 */
typedef struct lbbnn_wpb_args {
    const float *mu, *rho, *lambdal, *dWm, *dWv;          /* (O,I); dWv may be NULL (posterior-mean forward) */
    const float *z_fwd, *z_kl, *r0_c;                     /* (I) or NULL                                     */
    const float *da_mu, *da_var;                          /* (O) or NULL                                     */
    const float *g_kl;                                    /* device scalar or NULL                           */
    lbbnn_priors_t priors;
    float *dmu, *drho, *dlambdal;                         /* (O,I) outputs                                   */
    float *dz_fwd, *dz_kl, *dr0_c;                        /* (I) outputs or NULL                             */
    float *work;
    int O, I;
    int nsplit;                                           /* dWm / dWv are nsplit slabs of (O,I), added here; 0 or 1 = plain */
    int64_t split_stride;                                 /* floats between slabs                                            */
} lbbnn_wpb_args_t;

int64_t lbbnn_weight_pass_backward_workspace(int O, int I);
int lbbnn_weight_pass_backward(const lbbnn_wpb_args_t* args, void* stream);


/* rng[1] += delta (device side, so graph replays draw fresh noise). */
int lbbnn_rng_advance(uint64_t* rng, uint64_t delta, void* stream);

/* Reproduce the in-kernel N(0,1) draws (tests, and the backward pass that must re-create eps):
 * rows > 0: out[r][c] = normal(counter (row_base + r, c/4))[c%4]  -- lbbnn_lrt_gemm's epilogue indexing;
 * rows == 0: out[i]   = normal(counter (i/4, 0))[i%4], i < cols   -- the 1-D draws of K3 / K5. */
int lbbnn_philox_normal(const uint64_t* rng, uint32_t rng_stream, int64_t row_base, int64_t rows,
                        int64_t cols, float* out, void* stream);

/* log_softmax over the last dim of a (B,O<=64) matrix, in place allowed (…LRT.py:210). */
int lbbnn_log_softmax_rows(const float* in, int ldi, float* out, int ldo, int B, int O, void* stream);

/* Vector-sized backward of one MNF layer with planar flows -- two single-workgroup launches around K1b:
 *
 * lbbnn_mnf_aux_backward: gradient of the KL wrt the auxiliary activations (LBBNN-GP-MF-MNF.py:211-233).
 *   m = mean_o tanh(act_mu + sqrt(act_var)*eps_act);  kl has -log_rb(m; r0_b1, r0_b2, zb_last);
 *   da_mu = g_kl * dkl/dm / O * (1 - act^2),  da_var = da_mu * eps_act / (2 sqrt(act_var));  aux[0] = m.
 *   zb_last = device pointer to scal[3] of the forward.  eps_act == NULL: re-created from the Philox state `rng`
 *   the forward used (stream EPS_ACT of layer_id), as are eps_fwd / eps_kl of the flow backward below.
 *
 * lbbnn_mnf_flow_planar_backward: given dz_fwd / dz_kl (I) from K1b, the upstream g_kl and the column sums of
 *   the output gradients g_sum = sum_b G_m, gv_sum = sum_b G_v (O; gv_sum NULL for a posterior-mean forward), re-runs
 *   the z flow (both draws) and the r flow forward keeping every intermediate z, then walks them backwards
 *   (flows2.py:168-179 planar step; LBBNN-GP-MF-MNF.py:182-187,199-208,224-233) and writes the gradients of
 *   q0_mean, q0_log_var, r0_b1, r0_b2, every flow's u/w/bias, bias_mu and bias_rho.  g_kl NULL = no KL branch
 *   (r-flow / r0_b gradients are then zero-filled).  work: lbbnn_mnf_flow_backward_workspace(I, Tz, Tr) floats.
 */
typedef struct lbbnn_planar_grad {
    float* u[LBBNN_MAX_FLOW_T];
    float* w[LBBNN_MAX_FLOW_T];
    float* b[LBBNN_MAX_FLOW_T];
} lbbnn_planar_grad_t;

typedef struct lbbnn_flow_bwd_args {
    const float *q0_mean, *q0_log_var, *eps_fwd, *eps_kl;   /* (I) */
    const float *r0_b1, *r0_b2;                              /* (I) */
    const float *aux;                                        /* aux[0] = m from lbbnn_mnf_aux_backward      */
    const float *dz_fwd, *dz_kl;                             /* (I) upstream, either may be NULL (= 0)       */
    const float *g_kl;                                       /* device scalar or NULL                        */
    const float *bias_mu, *bias_rho, *g_sum, *gv_sum;        /* (O)                                          */
    lbbnn_planar_flow_t z_flow, r_flow;
    lbbnn_priors_t priors;
    float *d_q0_mean, *d_q0_log_var, *d_r0_b1, *d_r0_b2;     /* (I) outputs                                  */
    float *d_bias_mu, *d_bias_rho;                           /* (O) outputs                                  */
    lbbnn_planar_grad_t d_z_flow, d_r_flow;
    float *work;
    int O, I;
    const uint64_t* rng;                                     /* eps_fwd / eps_kl == NULL: re-create the forward's draws  */
    uint32_t layer_id;                                       /*   (Philox streams EPS_Z / EPS_Z2 of this layer id)       */
} lbbnn_flow_bwd_args_t;

int lbbnn_mnf_aux_backward(const float* act_mu, const float* act_var, const float* eps_act, const float* r0_b1,
                           const float* r0_b2, const float* zb_last, const float* g_kl, int O, int I,
                           float* da_mu, float* da_var, float* aux, const uint64_t* rng, uint32_t layer_id, void* stream);
/* The same for n <= LBBNN_MAX_LAYERS layers in ONE launch (one workgroup per layer).  Every input is a by-product of the
 * forward pass, so a caller that knows d loss / d kl for all layers at once (bnn_amd: the backward of the network's KL sum)
 * runs all of them together. */
typedef struct {
    const float *act_mu, *act_var, *eps_act, *r0_b1, *r0_b2, *zb_last, *g_kl;
    float *da_mu, *da_var, *aux;
    const uint64_t* rng;
    int O, I;
    uint32_t layer_id;
} lbbnn_aux_bwd_args_t;
int lbbnn_mnf_aux_backward_batch(const lbbnn_aux_bwd_args_t* args, int n, void* stream);

/* Gradients of the bias parameters of a layer WITHOUT flows (the LRT layer: BayesianLinear.forward + kl of
 * LBBNN-GP-MF-LRT.py:171-173, 189-196 differentiated by loss.backward(), :223): activation mean + bias_mu, activation variance
 * + softplus(bias_rho)^2, KL bias term.  g_sum / gv_sum (O): column sums of the gradients with respect to the activation
 * mean / variance (lbbnn_output_grad; gv_sum NULL: posterior-mean forward), g_kl: device scalar, the gradient with respect
 * to the layer's KL (NULL: KL not part of the loss).  d_bias_mu = g_sum + g_kl (mu - mu_p) / s_p^2;
 * d_bias_rho = (2 sigma gv_sum + g_kl (sigma / s_p^2 - 1 / sigma)) sigmoid(rho).  Replaces the torch autograd graph over the
 * two bias vectors that the layer's backward built until round 3 (~25 small launches per layer). */
int lbbnn_bias_backward(const float* bias_mu, const float* bias_rho, const float* g_sum, const float* gv_sum,
                        const float* g_kl, const lbbnn_priors_t* priors, float* d_bias_mu, float* d_bias_rho, int O,
                        void* stream);
/* The same from the column-sum PARTIALS that lbbnn_output_grad leaves in its workspace when called with g_sum == NULL (B, O as
 * given there; has_gv: the call had std != NULL): the second level of the sums and the bias gradients in one launch, bitwise
 * what lbbnn_output_grad's own second level followed by lbbnn_bias_backward gives. */
int lbbnn_bias_backward_partials(const float* work, int B, int O, int has_gv, const float* bias_mu, const float* bias_rho,
                                 const float* g_kl, const lbbnn_priors_t* priors, float* d_bias_mu, float* d_bias_rho,
                                 void* stream);
int64_t lbbnn_mnf_flow_backward_workspace(int I, int Tz, int Tr);
int lbbnn_mnf_flow_planar_backward(const lbbnn_flow_bwd_args_t* args, void* stream);
/* The same for n <= LBBNN_MAX_LAYERS layers in ONE launch (one workgroup per layer): the chains are latency-bound and
 * independent of each other, so a network's worth takes the time of one.  Every flow must have at most 4 transforms
 * (LBBNN_E_SHAPE otherwise: use the single-layer entry point).  The caller must have all n layers' inputs ready -- i.e. defer
 * the vector-sized chains to the end of the backward pass (bnn_amd.layers.vector_backward_overlap). */
int lbbnn_mnf_flow_planar_backward_batch(const lbbnn_flow_bwd_args_t* args, int n, void* stream);

/* lbbnn_flow_chain -- a chain of 1-D (vector) flow transforms on one z (I): planar, radial, Householder,
 * Sylvester (flows2.py:72-95, 48-69, 122-135, 98-120) in any order ('mixed' = 5 x (Householder, Planar),
 * flows2.py:31-37).  One single-workgroup launch; fixed-order double-precision block reductions.
 *   z = z_in                                   if z_in != NULL
 *     = q0_mean + exp(q0_log_var)^.5 * eps     otherwise (eps (I) or NULL => Philox stream rng_stream, counter i/4);
 *       log_q0 (if != NULL) = sum(-0.5*log(pi) - 0.5*log_var - 0.5*eps^2-form of LBBNN-GP-MF-MNF.py:213-214)
 *   for each step: z = f(z); logdet += f.log_det()
 * Outputs: z_out (I) (also the working vector; may alias z_in), logdet (1), z_last (1) = z_out[I-1].
 * Step parameters: PLANAR p0=u p1=w p2=bias(1); RADIAL p0=z_0 (I) p1=log_alpha(1) p2=beta(1) -- as written the norm is
 * over the whole vector and H1+H2 is added to every element; HOUSEHOLDER p0=v; SYLVESTER p0=A (I x M row-major)
 * p1=B (M x I) p2=b (M), M <= LBBNN_MAX_SYLVESTER_M, log det by LU with partial pivoting (NaN if det < 0, as torch).
 */
#define LBBNN_FLOW_PLANAR 0
#define LBBNN_FLOW_RADIAL 1
#define LBBNN_FLOW_HOUSEHOLDER 2
#define LBBNN_FLOW_SYLVESTER 3
#define LBBNN_MAX_SYLVESTER_M 8

typedef struct lbbnn_flow_step {
    const float *p0, *p1, *p2;
    int type, M;
} lbbnn_flow_step_t;

typedef struct lbbnn_flow_chain {
    lbbnn_flow_step_t step[LBBNN_MAX_FLOW_T];
    int n;
} lbbnn_flow_chain_t;

int lbbnn_flow_chain(const lbbnn_flow_chain_t* chain, const float* z_in, const float* q0_mean,
                     const float* q0_log_var, const float* eps, const uint64_t* rng, uint32_t rng_stream, int I,
                     float* z_out, float* logdet, float* log_q0, float* z_last, void* stream);
/* The same chain applied to each of the R rows of z_in (R, row stride ldz) independently -- workgroup r carries row r:
 * z_out (R, row stride ldo; may alias z_in), logdet_rows (R, nullable).  This is the row-wise restatement SURVEY.md 8(a) F1
 * defines for a 2-D z (flows2.PropagateFlow itself raises on a 2-D z for planar / Householder / Sylvester / mixed flows:
 * torch.dot, flows2.py:87,129; its Radial transform takes ONE norm over all rows, flows2.py:61 -- not reproduced here). */
int lbbnn_flow_chain_rows(const lbbnn_flow_chain_t* chain, const float* z_in, int ldz, int R, int I,
                          float* z_out, int ldo, float* logdet_rows, void* stream);


/* lbbnn_output_grad -- the (B,O) elementwise head of the layer backward in one pass (LBBNN-GP-MF-LRT.py:172-175 read
 * backwards):   G_m = g_out (.) [out > 0 if relu],   G_v = G_m * eps / (2 std)     (std = sqrt(var_b) of the forward)
 * written both row-major (B,O) -- the A operands of dX = G_m.W_m + 2x(.)(G_v.W_v) -- and transposed (O,B) -- the A
 * operands of dW_m = G_m^T.x, dW_v = G_v^T.x^2 -- plus the column sums g_sum = sum_b G_m, gv_sum = sum_b G_v (the
 * bias gradients).  eps (B,O) explicit, or NULL => re-created in-kernel from the Philox state the forward used
 * (stream rng_stream, counter (row_offset + b, o/4)).  std == NULL => posterior-mean forward: only G_m / G_m^T / g_sum.
 * Column sums are deterministic (per-64-row partials in work, fixed-order second launch).
 */
typedef struct lbbnn_outgrad_args {
    const float *g_out, *out, *std, *eps;     /* (B,O), row strides ldg / ldo / ldo / O; out may be NULL when !relu   */
    const uint64_t* rng;
    float *gm, *gv, *gmT, *gvT;               /* (B,O) dense, (O,B) dense                                              */
    float *g_sum, *gv_sum, *work;             /* (O), (O), lbbnn_output_grad_workspace floats                          */
    int64_t row_offset;
    uint32_t rng_stream;
    int B, O, ldg, ldo, relu;
    const float* gv_scale;                    /* NULL, or (O): G_v[b,o] *= gv_scale[o] -- the variational-dropout alpha
                                                 (d delta / d(x^2 theta^2) = alpha, variational_dropout.py:65)          */
} lbbnn_outgrad_args_t;

int64_t lbbnn_output_grad_workspace(int B, int O);
int lbbnn_output_grad(const lbbnn_outgrad_args_t* args, void* stream);    /* gm == gv == NULL: only the transposes and the sums;
                                                                             g_sum == gv_sum == NULL: the column-sum partials
                                                                             stay in `work` for lbbnn_reduce_partials_batch   */
/* The second level of the two-level column sums of lbbnn_output_grad (Sum_b G_m, Sum_b G_v) and lbbnn_weight_pass_backward
 * (dz_fwd, dz_kl, dr0_c) for SEVERAL layers in one launch: both kernels leave per-row-block partial vectors in their `work`
 * arrays and normally finish them with a launch of their own (six ~5 us launches per training step of a three-layer net);
 * their results are read by the vector-sized backward chains only, so a caller that defers those chains can defer the sums
 * too.  Job: out[q][i] = Sum_b work[b * block_stride + q * q_stride + i], q < nq <= 3, i < ncols, b < nblk, in the order
 * the stand-alone launches use (16 interleaved partial sums over b, added in order): bit-identical results.
 *   lbbnn_output_grad(B, O):              nblk = ceil(B / 64), ncols = O, block_stride = O, q_stride = nblk * O, nq = 1 or 2
 *   lbbnn_weight_pass_backward(O, I):     nblk = ceil(O / 8), ncols = I, ld = (I + 3) & ~3, block_stride = 3 ld, q_stride = ld, nq = 3
 * A NULL out[q] skips that vector.  At most LBBNN_MAX_REDUCE_JOBS jobs per call. */
#define LBBNN_MAX_REDUCE_JOBS 8
typedef struct {
    const float* work;
    float* out[3];
    int64_t block_stride, q_stride;
    int nblk, ncols, nq;
} lbbnn_reduce_job_t;
int lbbnn_reduce_partials_batch(const lbbnn_reduce_job_t* jobs, int n, void* stream);
/* Input gradient of a <= 16-class head in ONE launch: out[b][i] = sum_c gm[b][c] wmT[i][c] + 2 x[b][i] sum_c gv[b][c] wvT[i][c]
 * (gv == wvT == NULL: the first sum alone).  wmT / wvT: [I][ldw] fp32, the (e_w z)^T / var_w^T operands of lbbnn_weight_operands_t. */
int lbbnn_head_dx(const float* gm, const float* gv, int ldg, const float* wmT, const float* wvT, int ldw,
                  const float* x, int ldx, float* out, int ldo, int B, int C, int I, void* stream);
/* Weight gradients of a <= 16-class head as nslabs split-K slabs (lbbnn_weight_pass_backward adds them in order):
 *   dWm[s][c][i] = sum_{b in slab s} gm[b][c] x[b][i],  dWv[s][c][i] = sum_{b in slab s} gv[b][c] x[b][i]^2   ([nslabs][C][I] fp32 each)
 * from the row-major gradients (B,C; row stride ldg) and the row-major layer input (B,I; row stride ldx) -- no transposed
 * operand of x is built (autograd of torch.mm at LBBNN-GP-MF-LRT.py:172-173).  gv == dWv == NULL: the mean product alone.
 * Slab s covers rows [s r, min((s + 1) r, B)), r = ceil(B / nslabs).  Deterministic (fixed order of every addition). */
int lbbnn_head_dw(const float* gm, const float* gv, int ldg, const float* x, int ldx, float* dWm, float* dWv,
                  int B, int C, int I, int nslabs, void* stream);
/* gx (B,I dense) += 2 * x (B,I; row stride ldx) * gxv (B,I dense): the input gradient of the variance GEMM folded
 * into dX = G_m.W_m + 2 x (.) (G_v.W_v)  (d/dx of (x^2).var_w^T, LBBNN-GP-MF-LRT.py:173). */
/* out = comb_add + 2 * comb_x (.) (x . w_op^T): the mean-only product of lbbnn_lrt_gemm with lbbnn_dx_combine fused into
 * its epilogue -- dX = G_m.W_m + 2 x (.) (G_v.W_v) is the second product's output directly (comb_add = G_m.W_m,
 * comb_x = the layer input).  comb_x / comb_add: (B,O) with row strides ld_cx / ld_ca; out may alias comb_add.
 * O > 16 (LBBNN_E_SHAPE otherwise); flags: LBBNN_F_SPLIT16 as for lbbnn_lrt_gemm. */
int lbbnn_lrt_gemm_combine(const float* x, int ldx, const void* w_op, int ld, const float* comb_x, int ld_cx,
                           const float* comb_add, int ld_ca, float* out, int ldo, int B, int I, int O, int flags,
                           void* stream);
int lbbnn_dx_combine(float* gx, const float* gxv, const float* x, int ldx, int B, int I, void* stream);

/* lbbnn_adam_step -- torch.optim.Adam's update (the optimizer of the reference's training scripts,
 * LBBNN-GP-MF-LRT.py:358, LBBNN-GP-MF-MNF.py:421) for a LIST of parameter tensors in one launch:
 *   g' = g + weight_decay * p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  t = *step + 1
 *   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * *step is a device-side float counter, read by every workgroup; advance != 0 adds 1 to it in a second one-thread
 * launch (set it on the last list of an optimizer step), so the whole step is HIP-graph capturable.
 */
#define LBBNN_ADAM_MAX_TENSORS 80
typedef struct lbbnn_adam_list {
    float* p[LBBNN_ADAM_MAX_TENSORS];
    const float* g[LBBNN_ADAM_MAX_TENSORS];
    float* m[LBBNN_ADAM_MAX_TENSORS];
    float* v[LBBNN_ADAM_MAX_TENSORS];
    int64_t numel[LBBNN_ADAM_MAX_TENSORS];
    int n;
} lbbnn_adam_list_t;

int lbbnn_adam_step(const lbbnn_adam_list_t* list, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float* step, int advance, void* stream);

/* lbbnn_matmul_splitk -- split-K form of the mean-only bf16x3 product for long contractions with few output tiles
 * (the weight gradients dW = G^T.x: K = batch):  out[z] (B,O; row stride ldo; slab stride B*ldo) =
 * x[:, Kz] . w[:, Kz]^T with Kz = [z*kchunk, min(I, (z+1)*kchunk)), z < ceil(I / kchunk); kchunk a multiple of 32.
 * w_op: LBBNN_F_SPLIT16 operand of lbbnn_transpose_operand / lbbnn_weight_pass.  The consumer adds the slabs
 * (lbbnn_weight_pass_backward does, in a fixed order => deterministic).
 */
int lbbnn_matmul_splitk(const float* x, int ldx, const void* w_op, int ld, float* out, int ldo,
                        int B, int I, int O, int kchunk, void* stream);

/* lbbnn_multi_copy -- dst[i][0..numel[i]) = src[i][...] for a LIST of fp32 tensors in one launch: packing the gradients
 * of all parameters into the flat bucket that is all-reduced over RCCL (and unpacking it), instead of one copy kernel
 * per parameter tensor. */
typedef struct lbbnn_copy_list {
    float* dst[LBBNN_ADAM_MAX_TENSORS];
    const float* src[LBBNN_ADAM_MAX_TENSORS];      /* NULL = fill dst with zeros (a parameter without a gradient) */
    int64_t numel[LBBNN_ADAM_MAX_TENSORS];
    int n;
} lbbnn_copy_list_t;

int lbbnn_multi_copy(const lbbnn_copy_list_t* list, void* stream);

/* Stand-alone application of a dense coupling flow (RNVP / MNF type, flows2.py:188-241) to ONE vector, and its analytic
 * backward -- the pieces the vector-sized backward of an MNF layer with the reference's default flows is made of:
 *   lbbnn_flow_dense_apply:           z_out = f_T(...f_1(z_in)),  *logdet = sum of the transforms' log-dets
 *   lbbnn_flow_dense_apply_backward:  given d_zout (I) and *d_logdet (device scalar, NULL = 0): dz_in (I) and the
 *                                     gradient of every parameter of every transform (written, not accumulated).
 * which_mask selects mask_fwd (0) or mask_kl (1) of each transform.  One single-workgroup launch each; the backward
 * re-runs the forward keeping each transform's input (work: lbbnn_flow_dense_apply_workspace floats).
 */
#define LBBNN_MAX_DENSE_T 8
typedef struct lbbnn_dense_grad {
    float *w_in, *b_in, *w_mid[3], *b_mid[3], *w_a, *b_a, *w_b, *b_b;      /* same shapes as lbbnn_dense_transform_t */
} lbbnn_dense_grad_t;

int64_t lbbnn_flow_dense_apply_workspace(int I, int T);
int lbbnn_flow_dense_apply(const lbbnn_dense_transform_t* tr, int T, int which_mask, const float* z_in, int I,
                           float* z_out, float* logdet, void* stream);
int lbbnn_flow_dense_apply_backward(const lbbnn_dense_transform_t* tr, const lbbnn_dense_grad_t* grads, int T,
                                    int which_mask, const float* z_in, const float* d_zout, const float* d_logdet, int I,
                                    float* dz_in, float* work, void* stream);

/* lbbnn_flow_dense_rows -- the same coupling flows applied to R ROWS at once (each row with its own Bernoulli mask per
 * transform), the dense affine steps on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).  This is the as-written form
 * of `sample_z(batch_size)` (LBBNN-GP-MF-MNF.py:182-187: z_flow on a (B,I) matrix of which only the last row is kept -- the
 * layer kernels above compute that row alone) and what the stand-alone `PropagateFlow('RNVP'|'MNF').forward(z)` of
 * flows2.py:41-46,206-219,233-241 computes for a 1-D or (R,I) z.
 *   z_out (R, row stride ldo)  = f_T(... f_1(z_in))  row by row; may alias z_in
 *   logdet_rows (R), nullable  = sum_t sum_i (1 - m) log gate   (RNVP's log_det is this per-row vector, flows2.py:218-219;
 *                                the MNF type's is its sum over ALL rows, flows2.py:240-241: the caller adds the R values)
 *   masks [T][R][I] in {0,1}, or NULL: Bernoulli(0.5) drawn in-kernel from Philox (rng = {seed, offset}, stream
 *   rng_stream, counter = (row_base + row, t, i/4)); mask_out (nullable, [T][R][I]) receives the masks used.
 * The mask_fwd / mask_kl fields of the transforms are ignored.  One launch: a 256-thread workgroup carries 16 rows through
 * the whole chain with z resident in LDS.  I <= lbbnn_flow_dense_rows_max_dim() (LDS capacity; LBBNN_E_SHAPE otherwise).
 * Deterministic (fixed-order sums, no atomics). */
/* lbbnn_q0_rows -- the R-row z0 of sample_z(batch_size = R) (LBBNN-GP-MF-MNF.py:183-185):
 *   z0[r][i] = q0_mean[i] + exp(q0_log_var[i])^(1/2) * eps[r][i],   z0 (R,I) dense rows
 * eps (R,I) explicit, or NULL: N(0,1) from Philox stream rng_stream with counter (i/4, R-1-r) -- the last row's draw is
 * the draw the fused layer kernels make for the kept row, so both forms see the same z there. */
int lbbnn_q0_rows(const float* q0_mean, const float* q0_log_var, const float* eps, const uint64_t* rng,
                  uint32_t rng_stream, int R, int I, float* z0, void* stream);
int lbbnn_flow_dense_rows_max_dim(void);
int lbbnn_flow_dense_rows(const lbbnn_dense_transform_t* tr, int T, const float* masks, float* mask_out,
                          const uint64_t* rng, uint32_t rng_stream, uint64_t row_base,
                          const float* z_in, int ldz, int R, int I,
                          float* z_out, int ldo, float* logdet_rows, void* stream);

/* lbbnn_mnf_flow_dense_backward -- the vector-sized backward (cf. lbbnn_mnf_flow_planar_backward) for a layer whose flows are dense coupling flows
 * (RNVP / MNF type, flows2.py:188-241; the reference's default).  Inputs as lbbnn_mnf_flow_planar_backward; the
 * transforms (with the masks of the forward call) in zt / rt, their gradient destinations in d_zt / d_rt (host arrays),
 * and `save` = what the forward kept (lbbnn_dense_layer_t::save).  2 + 3*(Tz+Tr) launches spread over workgroups: per
 * transform, walking backwards, the two I x H heads (outer-product gradients, share of dy), the H x H chain in one
 * workgroup, the H x I input layer.  Both draws of the z flow are handled together and the SUM of their parameter
 * gradients is written once; every reduction has a fixed order (deterministic).  g_kl NULL: no KL branch, r-flow and
 * r0_b gradients are zero-filled.  Tz / Tr must be the values the forward ran with (they fix the layout of `save`).
 * work: lbbnn_mnf_flow_dense_backward_workspace(I) floats.
 */
typedef struct lbbnn_dense_bwd_args {
    const float *q0_mean, *q0_log_var, *eps_fwd, *eps_kl;   /* (I); eps NULL: re-created from rng / layer_id          */
    const float *r0_b1, *r0_b2;                              /* (I)                                                   */
    const float *aux;                                        /* aux[0] = m from lbbnn_mnf_aux_backward                */
    const float *dz_fwd, *dz_kl;                             /* (I) upstream from K1b, either may be NULL (= 0)       */
    const float *g_kl;                                       /* device scalar or NULL                                 */
    const float *bias_mu, *bias_rho, *g_sum, *gv_sum;        /* (O)                                                   */
    const lbbnn_dense_transform_t *zt, *rt;
    const lbbnn_dense_grad_t *d_zt, *d_rt;
    lbbnn_priors_t priors;
    float *d_q0_mean, *d_q0_log_var, *d_r0_b1, *d_r0_b2;     /* (I) outputs                                           */
    float *d_bias_mu, *d_bias_rho;                           /* (O) outputs                                           */
    const float *save;
    float *work;
    int Tz, Tr, O, I;
    const uint64_t* rng;
    uint32_t layer_id;
} lbbnn_dense_bwd_args_t;

int64_t lbbnn_mnf_flow_dense_backward_workspace(int I);
int lbbnn_mnf_flow_dense_backward(const lbbnn_dense_bwd_args_t* args, void* stream);
/* n <= LBBNN_MAX_LAYERS layers in the SAME 2 + 3*(Tz+Tr) launches (blockIdx.z = layer): the launches are latency-bound, so a
 * network's worth takes about the time of one layer.  The layers must agree on Tz, Tr and on having a KL branch
 * (LBBNN_E_SHAPE otherwise), and all of their inputs must be ready: defer the chains to the end of the backward pass. */
int lbbnn_mnf_flow_dense_backward_batch(const lbbnn_dense_bwd_args_t* args, int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Round 3: the row-scaled FP16 operand format (LBBNN_F_F16S) and the descriptor form of the dual-moment GEMM.
 *
 * Arithmetic (replaces the same reference lines as lbbnn_lrt_gemm: LBBNN-GP-MF-LRT.py:172-175, LBBNN-GP-MF-MNF.py:197-200):
 *   every operand value w is held as hi + lo, both IEEE fp16, products on v_mfma_f32_16x16x32_f16, fp32 accumulate:
 *     mean = x.e_w^T    as xh.eh + xh.el + xl.eh                      (the dropped xl.el term is 2^-22 relative)
 *     var  = x^2.var_w^T as sh.vh + sh.vl + sl.vh, s = x^2 * 2^-8     (LBBNN_F_VAR1: sh.vh alone, sh = RNE16(s))
 *   fp16 has 5 exponent bits, so the weight operands carry one exact power-of-two scale per output row (e_scale[o],
 *   v_scale[o]: the row maximum lands in [2^13, 2^14)) which the epilogue takes out of the accumulators again, and x^2 is
 *   formed from x * 2^-4 (fp16 overflows at 65504: the format holds |x| < 4096; a producer that sees a larger activation
 *   poisons its output with NaN rather than saturating silently -- `range_guard`).
 *   Measured against fp64 on the headline layers (tools/format_error.py): 3 + 3 products 2.3-3.0e-8 of max|out| (an fp32
 *   accumulate of exact products, torch.mm, gives 1-3e-7), 3 + 1 products 1.4-1.8e-5 (contract: 1e-4).
 *
 * Layout: a weight row is the 128-B-line form of LBBNN_F_SPLIT16 (lbbnn_device.h) with fp16 values.  x may be given
 *   - as fp32 rows (the network input): split in registers by the consumer, or
 *   - as PLANES (LBBNN_F_XPLANES): rows of ldx fp32-sized slots, per 32-k chunk one 128-B line of units
 *     (xh k0-7 | xl k0-7 | xh k8-15 | xl k8-15 | ...), ldx a multiple of 32, tail k >= I zero -- what this GEMM writes
 *     to `out_planes` for the next layer (ReLU applied) and what lbbnn_format_x makes from fp32 rows.  A consumer of
 *     planes spends 4 packed-fp16 instructions per 2 k on x^2 and none on the split.
 */
#define LBBNN_F_F16S 0x40      /* operands in the row-scaled fp16 hi + lo format (mean_scale / wvar_scale required) */
#define LBBNN_F_VAR1 0x80      /* with LBBNN_F_F16S: ONE product for the variance GEMM (3 + 1 products per tile step); var_w is then a
                                  plain fp16 matrix [O][ld] halves (the hi part alone: 64 B per row and K step instead of 128)  */
#define LBBNN_F_XPLANES 0x100  /* with LBBNN_F_F16S: x is given as fp16 hi | lo planes                                */

typedef struct lbbnn_gemm_desc {
    const void* x; int ldx;                      /* (B, I): fp32 rows of ldx floats, or planes (ldx fp32-sized slots per row) */
    const void* e_w; const void* var_w; int ld;  /* [O][ld] operands from lbbnn_layers_operands (split == 2) / lbbnn_weight_pass_f16 */
    const float* mean_scale;                     /* (O) e_scale  */
    const float* wvar_scale;                     /* (O) v_scale  */
    const float* bias_mean;                      /* (O) or NULL  */
    const float* bias_var;                       /* (O) or NULL  */
    const float* var_scale;                      /* (O) or NULL: extra per-feature factor on the variance product */
    const float* eps;                            /* (B, O) explicit N(0,1) draws, or NULL => Philox(rng)          */
    const uint64_t* rng; uint32_t rng_stream; int64_t row_offset;
    float* out; int ldo;                         /* (B, O) fp32 output, or NULL when only planes are wanted       */
    void* out_planes; int ldp;                   /* (B, ldp) plane rows for the next layer, or NULL; needs O % 8 == 0, ldp % 32 == 0 */
    float* std_out;                              /* (B, O) sqrt(var) for the backward pass, or NULL               */
    int B, I, O, flags;                          /* LBBNN_F_RELU | LBBNN_F_F16S | LBBNN_F_VAR1 | LBBNN_F_XPLANES   */
    /* optional KL finalize of a network carried by one extra workgroup (as lbbnn_lrt_gemm_finalize_adv) */
    const lbbnn_layer_desc_t* layers; int n_layers; const uint64_t* fin_rng; float* kl_total;
    uint64_t* rng_live; uint64_t advance;
    /* optional HEAD FOLD (head_out != NULL; needs LBBNN_F_RELU): the layer that FOLLOWS this one has head_classes <= 16 outputs
     * (the 10-class head of BayesianNetwork, LBBNN-GP-MF-MNF.py:255-256).  Its two moment products are formed in this launch's
     * epilogue from the accumulators (fp16 hi + lo on v_mfma_f32_16x16x16_f16, per-o-tile partials in head_slab), a second small
     * launch adds the partials, the head's bias / variance bias / noise (head_eps or Philox(rng, head_rng_stream)) and, with
     * LBBNN_F_LOG_SOFTMAX in head_flags, log_softmax: head_out (B, head_classes).  head_e / head_v: the head's fp32 operands
     * [head_classes][head_ld] (lbbnn_layers_operands, split == 0).  `out` / `out_planes` may then both be NULL: the hidden
     * activation of this layer is never stored.  head_slab: lbbnn_head_slab_floats(B, O) floats. */
    const float* head_e; const float* head_v; int head_ld; int head_classes;
    const float* head_bias_mean; const float* head_bias_var; const float* head_eps; uint32_t head_rng_stream;
    float* head_out; int head_ldo; float* head_slab; int head_flags;
} lbbnn_gemm_desc_t;

int lbbnn_lrt_gemm_ex(const lbbnn_gemm_desc_t* d, void* stream);
int64_t lbbnn_head_slab_floats(int B, int O);

/* fp32 rows -> fp16 hi | lo planes (the x format above); I % 8 == 0, x 16-B aligned rows, ldp % 32 == 0, ldp >= I.
 * Writes the whole row of planes (tail slots zero). */
int lbbnn_format_x(const float* x, int ldx, void* planes, int ldp, int B, int I, void* stream);

/* The scalar head of the training step (train(), LBBNN-GP-MF-MNF.py:268-271), one launch each, deterministic:
 *   lbbnn_elbo_loss:            *loss = -sum_b logp[b][target[b]] + (kl ? *kl * kl_scale : 0)   (F.nll_loss(reduction='sum') + kl / N)
 *   lbbnn_elbo_loss_backward:   g_logp[b][c] = -(*g) (c == target[b]);  *g_kl = (*g) * kl_scale (g_kl may be NULL)
 *   lbbnn_log_softmax_backward: out = g - exp(logp) * rowsum(g)   (backward of F.log_softmax(dim=1), C <= 64)
 * target: int64 class indices (entries outside [0, C) contribute nothing, as ignore_index does). */
int lbbnn_elbo_loss(const float* logp, int ldp, const int64_t* target, int B, int C, const float* kl, float kl_scale,
                    float* loss, void* stream);
int lbbnn_elbo_loss_backward(const float* g, const int64_t* target, int B, int C, float kl_scale, float* g_logp, float* g_kl,
                             void* stream);
/* lbbnn_elbo_loss_backward and, in the same launch, the gradient with respect to the LOGITS when `logp` is the log_softmax of a
 * layer's logits (what lbbnn_log_softmax_backward would make of g_logp: bitwise the same): g_logits (B,C dense). */
int lbbnn_elbo_loss_backward_logits(const float* g, const int64_t* target, const float* logp, int ldp, int B, int C,
                                    float kl_scale, float* g_logp, float* g_logits, float* g_kl, void* stream);
int lbbnn_log_softmax_backward(const float* g, int ldg, const float* logp, int ldp, float* out, int ldo, int B, int C,
                               void* stream);

/* lbbnn_layers_operands_snap + lbbnn_format_x of the network input in the SAME launches: the planar flows of a network are two
 * latency-bound workgroups per layer, and the format job (one pass over x) runs as extra workgroups of that launch on the CUs
 * it leaves idle -- no launch of its own, nothing added to the critical path.  Falls back to a separate lbbnn_format_x launch
 * when the flow launch cannot carry it (no planar flows, shapes outside the register form of the flow kernel). */
int lbbnn_layers_operands_x(const lbbnn_layer_desc_t* layers, int n, uint64_t* rng, uint64_t* rng_snap, uint64_t advance,
                            const float* x, int ldx, void* planes, int ldp, int B, int I, void* stream);

/* lbbnn_weight_pass producing LBBNN_F_F16S operands + their row scales (single layer; the batched form is
 * lbbnn_layers_operands with split == 2). */
int lbbnn_weight_pass_f16(const float* mu, const float* rho, const float* lambdal,
                          const float* z_fwd, const float* z_kl, const float* r0_c,
                          const float* bias_rho, const lbbnn_priors_t* priors,
                          void* e_w, void* var_w, int ld, float* e_scale, float* v_scale,
                          float* kl_rows, float* act_mu, float* act_var, float* bias_var,
                          int O, int I, int flags /* 0 | LBBNN_F_VAR1 */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LBBNN_H_ */
