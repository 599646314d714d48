/*
 * psvo_hip.h -- C ABI of the MI355X-native PSVO hot path (libpsvo_hip.so).
 *
 * The reference (amoretti86/PSVO) has no FFI/plugin registry: the hot path sits behind a
 * Python object protocol (`Obj(model, FLAGS).get_log_ZSMC(obs, hidden)`, reference
 * src/trainer.py:108, src/runner.py:70-79).  This header is the boundary a maintainer would
 * bind from that protocol; each entry point names the reference code it replaces.
 * INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless it says "host"; float = fp32, indices int32;
 *   - the caller allocates every input, output and workspace buffer; the library never
 *     allocates device memory and keeps no pointer after a call returns;
 *   - every entry point is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 = ok, <0 = psvo_status error (see psvo_status_string);
 *   - HBM layout is particle-minor SoA so that a wavefront's lanes (particles) are contiguous:
 *         particles  X[t][b][d][n]      -> (T, B, Dx, N)
 *         weights    logW[t][b][n]      -> (T, B, N)
 *         per-seq    lse[t][b]          -> (T, B)
 *         features   mu2[t][b][d]       -> (T, B, Dx),  obs[t][b][e] -> (T, B, Dy)
 *         bsim noise eps_b[t][b][d][n][m] -> (T, B, Dx, N, M)
 *     (the reference's own internal layout is (T, N, B, D), src/SMC/SVO.py:176-178; the
 *     host mirror permutes to the reference's (B, T, N, Dx) at the Python boundary).
 *   - per-particle MLPs have ONE hidden layer of width H (reference default `*_layers=[32]`,
 *     src/runner_flag.py:53-57) or TWO of the same width (`*_layers=[64, 64]`, the example the reference's flag
 *     file gives, src/runner_flag.py:50-52; psvo_desc.layers = 2); kernels are instantiated for Dx in {2,3,4},
 *     Dy in {1,2}, H in {16,32,64} (two layers: {32,64}), M in {4,8,16,32}; N <= 512 (filter, PSVOwR reverse
 *     pass), N <= 1024 (backward simulation); any T, B <= 65535.  Anything else returns PSVO_ERR_UNSUPPORTED
 *     (never a silent fallback).
 */
#ifndef PSVO_HIP_H
#define PSVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSVO_ABI_VERSION 6

typedef enum {
    PSVO_OK = 0,
    PSVO_ERR_INVALID = -1,      /* null pointer / non-positive size / inconsistent desc       */
    PSVO_ERR_UNSUPPORTED = -2,  /* (Dx, Dy, H, M, N) outside the instantiated kernel set      */
    PSVO_ERR_HIP = -3           /* a HIP runtime call failed (launch error)                   */
} psvo_status;

/* Problem descriptor.  Mirrors the reference FLAGS that shape the path
 * (src/runner_flag.py:22-28,88-113; read at src/SMC/SVO.py:7-29, src/SMC/PSVO.py:9-19). */
typedef struct {
    int32_t B;          /* sequences in this call (FLAGS.batch_size, local shard)              */
    int32_t T;          /* time steps (FLAGS.time)                                             */
    int32_t N;          /* FLAGS.n_particles                                                   */
    int32_t M;          /* FLAGS.n_particles_for_BSim_proposal (bsim only)                     */
    int32_t Dx;         /* FLAGS.Dx                                                            */
    int32_t Dy;         /* FLAGS.Dy                                                            */
    int32_t H;          /* hidden width of ALL per-particle MLPs of the call (q1 / f / g / q1_inv): 16, 32 or
                           64.  Narrower or unequal widths (q1_layers=24, g_layers=16): the caller pads W1
                           columns, b1 and W2 rows with zeros up to H -- a padded unit contributes exactly 0 --
                           and drops the padded entries of the gradients (psvo_amd/SMC/SVO.py:_mlp_params)   */
    int32_t resample;   /* 1: multinomial resampling every step (SVO/AESMC/PSVO); 0: IWAE      */
    int32_t two_q;      /* FLAGS.use_2_q                                                       */
    int32_t bootstrap;  /* FLAGS.use_bootstrap (f shares q1's MLP and sigma)                   */
    int32_t emission;   /* 0: g = N(MLP_g(x), sig_g) (tf_mvn).  1: FLAGS.poisson_emission -- the
                           reference's tf_poisson (src/distribution/poisson.py:27-50) is a UNIT-scale
                           normal whose mean is softplus(MLP_g(x)) + 1e-6: pass sig_g = ones; the
                           dsig_g output is then meaningless                                    */
    int32_t layers;     /* hidden layers of ALL per-particle MLPs of the call: 0 or 1 = one (psvo_mlp.Wh / bh
                           ignored), 2 = two layers of width H each (FLAGS.q1_layers="64,64" ...,
                           src/transformation/MLP.py:24-38,50-54).  An MLP with fewer layers than the others
                           cannot be padded (an extra relu layer is not the identity): UNSUPPORTED upstream */
} psvo_desc;

/* Per-particle MLP, keras Dense layout (reference src/transformation/MLP.py:27-46):
 * hidden_0: W1 (Din, H) row-major, b1 (H); mu_layer: W2 (H, Dout) row-major, b2 (Dout);
 * with psvo_desc.layers == 2 also hidden_1 between them: Wh (H, H) row-major, bh (H)
 *     mu = relu(relu(x W1 + b1) Wh + bh) W2 + b2;
 * Wh / bh are NULL (ignored) for one hidden layer. */
typedef struct {
    const float* W1;
    const float* b1;
    const float* W2;
    const float* b2;
    const float* Wh;
    const float* bh;
} psvo_mlp;

int psvo_abi_version(void);
const char* psvo_status_string(int status);
/* Tuning switches (process-wide; for A/B measurements and tests -- results are the same to rounding under every value):
 *   PSVO_TUNE_BSIM_BWD: which reverse backward-simulation kernel psvo_bsim_backward launches.
 *     -1 default (measured best), 0 = lane = (chain, half, m) with per-j butterflies, 1 = "j on lanes" layout with the per-j
 *     sums on the VALU, 2 = the same with the per-j sums on v_mfma_f32_16x16x4_f32, 3 = the same as 1 with the pair exponents
 *     on v_mfma_f32_16x16x4_f32 (Dx = 2; otherwise as 1), 4 = the same as 3 on v_mfma_f32_16x16x32_bf16 with every f32
 *     operand split into three bf16 pieces (Dx = 2; otherwise as 1).  It changes psvo_bsim_blocks():
 *     set it before sizing workspaces. */
#define PSVO_TUNE_BSIM_BWD 1
/*   PSVO_TUNE_ROWS_BWD: rows per workgroup of psvo_rows_mlp_backward: 0 = chosen by the number of rows (default), 16, 64.
 *     It changes psvo_rows_mlp_blocks(): set it before sizing workspaces. */
#define PSVO_TUNE_ROWS_BWD 2
/*   PSVO_TUNE_L2_SPLIT: two-hidden-layer builds of psvo_bsim_forward / _backward: 0 (default) one lane per (chain, m);
 *   1 = a chain spread over 2 M lanes for small problems, as the one-layer builds do.  Set before sizing buffers. */
#define PSVO_TUNE_L2_SPLIT 3
/*   PSVO_TUNE_WGRAD2: psvo_mlp2_wgrad's H x H products: 0 (default) = v_mfma_f32_16x16x4_f32; 2 / 3 = bf16 matrix
 *   instructions with every f32 operand split into two / three bf16 pieces (products carried to 2^-17 / 2^-24 relative). */
#define PSVO_TUNE_WGRAD2 5
/*   PSVO_TUNE_FILTER_BWD: psvo_filter_backward in the bootstrap wiring with resampling (one hidden layer): 1 (default) = the
 *   affine scan (coefficients of every step in parallel, one or four waves per sequence for the recurrence, rows in
 *   parallel) where it pays -- no upstream gradient from a backward simulation, or N > 256 --, 2 = the scan wherever it
 *   applies, 0 = the persistent reverse kernel (one workgroup per sequence), which every other wiring uses. */
#define PSVO_TUNE_FILTER_BWD 6
/*   PSVO_TUNE_SKEW: start-up phase offset between the workgroups that share a CU in the backward-simulation kernels, in
 *   per cent of the kernel's estimate of its pair-phase length (default 0 = all workgroups start together: measured, it
 *   changes nothing -- profiles/r03_bsim_bwd_C5_ab.md -- and is kept as an A/B knob). */
#define PSVO_TUNE_SKEW 4
int psvo_set_tuning(int key, int value);          /* PSVO_OK or PSVO_ERR_INVALID */
int psvo_get_tuning(int key);

/* hipGetErrorString of the most recent failed launch on the calling thread (PSVO_ERR_HIP). */
const char* psvo_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------
 * Forward particle filter.  Replaces SVO.SMC (reference src/SMC/SVO.py:60-180) including
 * sample_from_2_dist (:182-232, diagonal branch), resample_X / get_resample_idx (:243-300)
 * and the per-step reduce_logsumexp of compute_log_ZSMC (:302-311).
 *
 * One persistent workgroup per sequence loops over t in-kernel.
 *
 *  q1, f, g       per-particle MLPs; when desc->bootstrap != 0, `f` is ignored (f == q1).
 *  sig_q1/sig_q2/sig_f (Dx), sig_g (Dy)  already-clipped scales max(softplus(raw), min)
 *                 (src/distribution/mvn.py:80-90); sig_q2 ignored when !two_q;
 *                 sig_f ignored when bootstrap.
 *  mu2  (T,B,Dx)  hoisted MLP_q2(e_t[b]) (proposal term that does not depend on particles);
 *                 ignored when !two_q.
 *  m0   (B,Dx), sig0 (Dx)   t=0 proposal term: MLP_q0(X0 feature), sigma_q0 (SVO.py:80-88).
 *  fm0  (B,Dx), fsig0 (Dx)  t=0 transition term: equal to (m0, sig0) when bootstrap&&two_q,
 *                 else (MLP_f(X0 feature), sigma_f) (SVO.py:91-92).
 *  obs  (T,B,Dy)
 *  eps  (T,B,Dx,N) standard-normal draws (injected; the reference draws them with mvn.sample)
 *  u    (T,B,N)   uniforms in [0,1) for the multinomial draw; used when idx_in == NULL
 *  idx_in (T,B,N) int32 teacher-forced ancestor indices or NULL
 *
 *  outputs: X (T,B,Dx,N) pre-resampling particles (Xs_ta), Xanc (T,B,Dx,N) resampled particles
 *  (X_ancestors_ta), Fm (T,B,Dx,N) = MLP_f(X_t) (transition means of every forward particle,
 *  consumed by the backward simulation), P1 (T,B,Dx,N) = MLP_q1(X_t) (optional, may be NULL; only
 *  needed by psvo_filter_backward when !bootstrap), logW (T,B,N) (log_Ws_ta), idx_out (T,B,N)
 *  ancestors, lse (T,B) = logsumexp_n logW[t,b,:].
 * ------------------------------------------------------------------------------------------- */
int psvo_filter_forward(const psvo_desc* desc,
                        const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g,
                        const float* sig_q1, const float* sig_q2, const float* sig_f, const float* sig_g,
                        const float* mu2,
                        const float* m0, const float* sig0, const float* fm0, const float* fsig0,
                        const float* obs, const float* eps, const float* u, const int32_t* idx_in,
                        float* X, float* Xanc, float* Fm, float* P1, float* logW, int32_t* idx_out,
                        float* lse, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Forward particle filter with STATE-DEPENDENT diagonal scales: FLAGS.output_cov and FLAGS.diag_cov
 * (reference src/runner_flag.py:67-70,221-222).  Every MLP then has a second output head (`sigma_layer`,
 * src/transformation/MLP.py:40-46) and every distribution's scale is
 *     sigma(input) = sigma_con + 0.1 * (exp(hidden(input) W_sigma + b_sigma) + 1e-6)
 * (MLP.py:58-61, src/distribution/mvn.py:66-71; sigma_con = the state-independent vector of get_sigma, mvn.py:80-90).
 * Same loop and outputs as psvo_filter_forward (SVO.SMC, src/SMC/SVO.py:60-180); differences:
 *  q1, f, g        W2 = [mu_layer | sigma_layer] (H, 2 Dout) row-major, b2 = [b_mu | b_sigma] (2 Dout); one hidden
 *                  layer (desc->layers <= 1; two layers: PSVO_ERR_UNSUPPORTED).  desc->emission = 1: the reference's
 *                  tf_poisson drops MLP_g's covariance head (src/distribution/poisson.py:33) -- unit scale, sigc_g unused
 *  sigc_q1/sigc_f (Dx), sigc_g (Dy)   the sigma_con parts (already clipped); sigc_f ignored when bootstrap
 *  mu2, sig2 (T,B,Dx)   hoisted q2 mean AND scale per step and sequence; ignored when !two_q
 *  m0, sig0, fm0, fsig0 (B,Dx)   t = 0 proposal / transition mean and scale per sequence (SVO.py:80-92)
 *  outputs         as psvo_filter_forward, plus Fs (T,B,Dx,N) = the scale of f given X_t (beside its mean Fm) and
 *                  P1s (T,B,Dx,N) beside P1 (both required when !bootstrap, ignored otherwise).
 * ------------------------------------------------------------------------------------------- */
int psvo_filter_forward_cov(const psvo_desc* desc,
                            const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g,
                            const float* sigc_q1, const float* sigc_f, const float* sigc_g,
                            const float* mu2, const float* sig2,
                            const float* m0, const float* sig0, const float* fm0, const float* fsig0,
                            const float* obs, const float* eps, const float* u, const int32_t* idx_in,
                            float* X, float* Xanc, float* Fm, float* Fs, float* P1, float* P1s, float* logW,
                            int32_t* idx_out, float* lse, void* stream);

/* Reverse mode of psvo_filter_forward_cov (TensorFlow autodiff of the loop in the reference, src/trainer.py:115-118).
 *  inputs  : the forward call's inputs and outputs; upstream gradients dlse (T,B), dFm_ext / dFs_ext (T,B,Dx,N),
 *            dlogW_ext (T,B,N), each optional (NULL = zero).
 *  outputs : rows for psvo_mlp_wgrad, one pair per MLP -- w.r.t. the mu_layer output and w.r.t. the RAW sigma_layer
 *            output (d sigma / d raw = 0.1 exp(raw) already applied): dP / dPs (T,B,Dx,N) for MLP_q1 (which is also f
 *            when bootstrap), dF / dFs for MLP_f (!bootstrap), dG / dGs (T,B,Dy,N) for MLP_g;
 *            dmu2, dsig2 (T,B,Dx); dm0, dsig0, dfm0, dfsig0 (B,Dx) (written separately even when the caller passed the
 *            same buffer as m0 and fm0: add them); dsigc_q1 / dsigc_f (Dx), dsigc_g (Dy) = d loss / d sigma_con.
 *  ws      : workspace, psvo_filter_cov_ws_floats(B, T, N, Dx, Dy) floats, 16-byte aligned. */
long long psvo_filter_cov_ws_floats(int B, int T, int N, int Dx, int Dy);
int psvo_filter_backward_cov(const psvo_desc* desc,
                             const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g,
                             const float* sigc_q1, const float* sigc_f, const float* sigc_g,
                             const float* mu2, const float* sig2,
                             const float* m0, const float* sig0, const float* fm0, const float* fsig0,
                             const float* obs, const float* eps,
                             const float* X, const float* Fm, const float* Fs, const float* P1, const float* P1s,
                             const float* logW, const float* lse, const int32_t* idx,
                             const float* dlse, const float* dFm_ext, const float* dFs_ext, const float* dlogW_ext,
                             float* dP, float* dPs, float* dF, float* dFs, float* dG, float* dGs,
                             float* dmu2, float* dsig2,
                             float* dm0, float* dsig0, float* dfm0, float* dfsig0,
                             float* dsigc_q1, float* dsigc_f, float* dsigc_g,
                             float* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward simulation with proposal, STATE-DEPENDENT diagonal scales (FLAGS.output_cov and FLAGS.diag_cov; see
 * psvo_filter_forward_cov).  Same loop and outputs as psvo_bsim_forward (PSVO.backward_simulation_w_proposal, reference
 * src/SMC/PSVO.py:69-203); differences:
 *  Fm, Fs (T,B,Dx,N)   mean AND scale of f given every forward particle (outputs of psvo_filter_forward_cov): the transition
 *                  tile (PSVO.py:128-133, never materialised) has a scale per forward particle
 *  f, g, q1_inv    W2 = [mu_layer | sigma_layer] (H, 2 Dout), b2 (2 Dout); one hidden layer; desc->emission = 1 drops MLP_g's
 *                  covariance head as the reference's tf_poisson does
 *  sigc_f, sigc_q1inv (Dx), sigc_g (Dy)   the sigma_con parts
 *  bmu2, bsig2 (T,B,Dx); minit, sinit, imean, isig (B,Dx)   hoisted means and scales per row (BSim_q2; BSim_q_init at
 *                  t = T-1; the t = 0 prior term f / q0 at mu_0, PSVO.py:169-175)
 *  saves (all four or none; required by psvo_bsim_backward_cov): lam_all (T,B,N,M) log2-domain filter term, om_all (T,B,N,M)
 *                  normalised sub-particle log-weights, mu1_all / s1_all (T,B,Dx,N) mean and scale of q1_inv given bwX[t+1].
 * ------------------------------------------------------------------------------------------- */
int psvo_bsim_forward_cov(const psvo_desc* desc,
                          const float* Fm, const float* Fs, const float* logW, const float* lse,
                          const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                          const float* sigc_f, const float* sigc_g, const float* sigc_q1inv,
                          const float* bmu2, const float* bsig2,
                          const float* minit, const float* sinit, const float* imean, const float* isig,
                          const float* obs, const float* eps_b, const float* u_b, const int32_t* sel_in,
                          float* bwX, float* flp, float* glp, float* Omega, int32_t* sel_out, float* score,
                          float* lam_all, float* om_all, float* mu1_all, float* s1_all,
                          void* stream);

/* Reverse mode of psvo_bsim_forward_cov (TensorFlow autodiff of the loop in the reference, src/trainer.py:115-118).
 *  inputs  : the forward call's inputs, its outputs bwX, sel, the four saves, and dscore (B,N) = d loss / d score.
 *  outputs : rows for psvo_mlp_wgrad, w.r.t. the mu_layer output and w.r.t. the RAW sigma_layer output of every MLP
 *            evaluation: xt (T,B,Dx,N,M) sub-particles, dFt / dFts (T,B,Dx,N,M) for MLP_f(x~), dGt / dGts (T,B,Dy,N,M)
 *            for MLP_g(x~), dmu1 / dmu1s (T,B,Dx,N) for MLP_q1inv(bwX[t+1]);
 *            ACCUMULATED with float atomics -- the caller zero-fills them: dFm, dFs (T,B,Dx,N), dlogW (T,B,N), dlse (T,B)
 *            (-> psvo_filter_backward_cov as dFm_ext, dFs_ext, dlogW_ext and added to dlse), dbmu2, dbsig2 (T,B,Dx),
 *            dminit, dsinit, dimean, disig (B,Dx), dsigc_f, dsigc_q1inv (Dx), dsigc_g (Dy).
 *  The summation order of the atomics is not fixed: results are reproducible to rounding, not bit for bit. */
int psvo_bsim_backward_cov(const psvo_desc* desc,
                           const float* Fm, const float* Fs, const float* logW, const float* lse,
                           const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                           const float* sigc_f, const float* sigc_g, const float* sigc_q1inv,
                           const float* bmu2, const float* bsig2,
                           const float* minit, const float* sinit, const float* imean, const float* isig,
                           const float* obs, const float* eps_b, const float* bwX, const int32_t* sel,
                           const float* lam_all, const float* om_all, const float* mu1_all, const float* s1_all,
                           const float* dscore,
                           float* xt, float* dFt, float* dFts, float* dGt, float* dGts, float* dmu1, float* dmu1s,
                           float* dFm, float* dFs, float* dlogW, float* dlse,
                           float* dbmu2, float* dbsig2,
                           float* dminit, float* dsinit, float* dimean, float* disig,
                           float* dsigc_f, float* dsigc_g, float* dsigc_q1inv,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * PSVOwR (backward simulation WITH RESAMPLING across the chains, reference src/SMC/PSVOwR.py:65-198) with state-dependent
 * diagonal scales.  Inputs as psvo_bsim_forward_cov plus u_r (T,B,N) uniforms of the cross-chain draw (or anc_in (T,B,N)
 * teacher-forced ancestors); outputs as psvo_bsimwr_forward: bwX, bwXanc (T,B,Dx,N), bwW (T,B,N) per-step chain
 * log-weights, lseW (T,B) = logsumexp_n bwW, sel_out, anc_out (T,B,N), plus omsel (T,B,N) (the drawn sub-particles'
 * normalised log-weights = the logits of the cross-chain draw; required) and the four saves.
 * The cross-chain draw couples every chain of a sequence once per step: the loop is T + 2 launches (one per time step,
 * one for the draw of step 0, one for lseW) issued by this call on `stream`; no cooperative launch, no workspace.
 * psvo_bsimwr_backward_cov: its reverse (T launches); outputs as psvo_bsim_backward_cov (rows; accumulated with float
 * atomics into zero-filled dFm, dFs, dlogW, dlse, dbmu2, dbsig2, dminit, dsinit, dimean, disig, dsigc_*); the q1_inv rows
 * dmu1 / dmu1s pair with the inputs bwXanc[t+1]; dXs (T,B,Dx,N): zero-filled workspace (d loss / d bwX).
 * ------------------------------------------------------------------------------------------- */
int psvo_bsimwr_forward_cov(const psvo_desc* desc,
                            const float* Fm, const float* Fs, const float* logW, const float* lse,
                            const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                            const float* sigc_f, const float* sigc_g, const float* sigc_q1inv,
                            const float* bmu2, const float* bsig2,
                            const float* minit, const float* sinit, const float* imean, const float* isig,
                            const float* obs, const float* eps_b, const float* u_b, const float* u_r,
                            const int32_t* sel_in, const int32_t* anc_in,
                            float* bwX, float* bwXanc, float* bwW, float* lseW, int32_t* sel_out, int32_t* anc_out,
                            float* omsel, float* lam_all, float* om_all, float* mu1_all, float* s1_all,
                            void* stream);
int psvo_bsimwr_backward_cov(const psvo_desc* desc,
                             const float* Fm, const float* Fs, const float* logW, const float* lse,
                             const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                             const float* sigc_f, const float* sigc_g, const float* sigc_q1inv,
                             const float* bmu2, const float* bsig2,
                             const float* minit, const float* sinit, const float* imean, const float* isig,
                             const float* obs, const float* eps_b,
                             const float* bwXanc, const float* bwW, const float* lseW,
                             const int32_t* sel, const int32_t* anc,
                             const float* lam_all, const float* om_all, const float* mu1_all, const float* s1_all,
                             const float* dlseW,
                             float* xt, float* dFt, float* dFts, float* dGt, float* dGts, float* dmu1, float* dmu1s,
                             float* dFm, float* dFs, float* dlogW, float* dlse,
                             float* dbmu2, float* dbsig2,
                             float* dminit, float* dsinit, float* dimean, float* disig,
                             float* dsigc_f, float* dsigc_g, float* dsigc_q1inv,
                             float* dXs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward simulation with proposal.  Replaces PSVO.backward_simulation_w_proposal
 * (reference src/SMC/PSVO.py:69-203): the (M, N, N, B) transition tile (:128-133) is never
 * materialised -- forward-particle means are staged in LDS and reduced with an online
 * log-sum-exp.
 *
 *  X, Fm, logW, lse      outputs of psvo_filter_forward (pre-resampling history, PSVO.py:39)
 *  f, g, q1_inv          per-particle MLPs (f == q1 when bootstrap: pass q1)
 *  sig_f, sig_g, sig_q1inv, sig_bq2 (Dx / Dy)
 *  bmu2 (T,B,Dx)         hoisted MLP_BSim_q2(enc_t[b])
 *  minit (B,Dx), sig_init (Dx)   MLP_BSim_q_init(enc_{T-1}[b]), its sigma (PSVO.py:86-87)
 *  imean (B,Dx), isig (Dx)       t=0 "filter" term: (MLP_q0(mu_0), sigma_q0) when
 *                        bootstrap&&two_q else (MLP_f(mu_0), sigma_f) (PSVO.py:169-173)
 *  obs (T,B,Dy); eps_b (T,B,Dx,N,M) normal draws; u_b (T,B,N) uniforms; sel_in (T,B,N) or NULL
 *
 *  outputs: bwX (T,B,Dx,N) (bw_Xs), flp (T,B,N) (f_log_probs), glp (T,B,N) (g_log_probs),
 *  Omega (T,B,N) (bw_log_Omegas), sel_out (T,B,N) chosen sub-particle, score (B,N) =
 *  sum_t(flp+glp-Omega) (psvo_elbo_bsim turns it into logsumexp_n score - log N, PSVO.py:52-67).
 *  lam2_all (T,B,N,M), om_all (T,B,N,M), mu1_all (T,B,Dx,N): optional (NULL) saves for
 *  psvo_bsim_backward -- the log2-domain filter term, the normalised sub-particle log-weights and
 *  MLP_q1inv(bwX[t+1]).
 * ------------------------------------------------------------------------------------------- */
int psvo_bsim_forward(const psvo_desc* desc,
                      const float* X, const float* Fm, const float* logW, const float* lse,
                      const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                      const float* sig_f, const float* sig_g, const float* sig_q1inv, const float* sig_bq2,
                      const float* bmu2, const float* minit, const float* sig_init,
                      const float* imean, const float* isig,
                      const float* obs, const float* eps_b, const float* u_b, const int32_t* sel_in,
                      float* bwX, float* flp, float* glp, float* Omega, int32_t* sel_out,
                      float* score,
                      float* lam2_all, float* om_all, float* mu1_all,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reverse mode of psvo_bsim_forward (TensorFlow autodiff of PSVO.backward_simulation_w_proposal in
 * the reference, src/trainer.py:115-118).  The N x N tile is recomputed, not stored.
 *
 *  inputs  : the forward call's inputs (eps_b as drawn), its outputs bwX, sel and the three optional
 *            saves lam2_all (T,B,N,M), om_all (T,B,N,M), mu1_all (T,B,Dx,N) (required here), and
 *            dscore (B,N) = d loss / d score.
 *  outputs : rows for psvo_mlp_wgrad: xt (T,B,Dx,N,M) sub-particles, dFt (T,B,Dx,N,M) w.r.t.
 *            MLP_f(x~), dGt (T,B,Dy,N,M) w.r.t. MLP_g(x~), dmu1 (T,B,Dx,N) w.r.t. MLP_q1inv(bwX[t+1]);
 *            per-workgroup partials (nblk = psvo_bsim_blocks(desc)) dFm_part (T,B,nblk,Dx,N), dlogW_part (T,B,nblk,N)
 *            and sacc_part (B * nblk * psvo_bsim_acc_size(Dx, Dy) floats) for psvo_bsim_backward_fold;
 *            per-chain rows (to be summed over N): dbmu2_rows (T,B,Dx,N), dminit_rows (B,Dx,N), dimean_rows (B,Dx,N).
 *  The gradient w.r.t. lse is zero analytically (the normalised weights' gradients sum to zero per chain) and of the
 *  size of its fp32 rounding numerically; it is returned all the same (dlse below), as autodiff of the reference's graph
 *  carries it: psvo_filter_backward then subtracts the softmax-weighted sum from d logW, which is what keeps that pass
 *  well-conditioned when the forward particles are far from the data (tests/test_gpu_notebook_curve.py).
 *
 * psvo_bsim_backward_fold (one launch, to be issued right behind): folds the partials in workgroup order into
 *            dFm (T,B,Dx,N), dlogW (T,B,N) = d loss / d (Fm, logW) of the forward filter -> psvo_filter_backward, and the
 *            scale gradients dsig_f, dsig_q1inv, dsig_bq2, dsig_init, disig (Dx), dsig_g (Dy);
 *            dlse (T,B) or NULL = -sum_n dlogW[t,b,n] = d loss / d lse of the forward filter (ABI 5).
 * ------------------------------------------------------------------------------------------- */
int psvo_bsim_blocks(const psvo_desc* desc);      /* nblk for this problem under the current PSVO_TUNE_BSIM_BWD setting */
int psvo_bsim_acc_size(int Dx, int Dy);
int psvo_bsim_backward(const psvo_desc* desc,
                       const float* Fm, const float* logW, const float* lse,
                       const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                       const float* sig_f, const float* sig_g, const float* sig_q1inv, const float* sig_bq2,
                       const float* bmu2, const float* minit, const float* sig_init,
                       const float* imean, const float* isig,
                       const float* obs, const float* eps_b,
                       const float* bwX, const int32_t* sel,
                       const float* lam2_all, const float* om_all, const float* mu1_all,
                       const float* dscore,
                       float* xt, float* dFt, float* dGt, float* dmu1,
                       float* dFm_part, float* dlogW_part,
                       float* dbmu2_rows, float* dminit_rows, float* dimean_rows,
                       float* sacc_part, void* stream);
int psvo_bsim_backward_fold(const psvo_desc* desc, const float* dFm_part, const float* dlogW_part, const float* sacc_part,
                            const float* sig_q1inv, const float* sig_bq2, float* dFm, float* dlogW,
                            float* dsig_f, float* dsig_g, float* dsig_q1inv, float* dsig_bq2, float* dsig_init,
                            float* disig, float* dlse, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PSVOwR: backward simulation with cross-chain resampling and a per-step ELBO.
 * Replaces PSVOwR.backward_simulation_w_resampling (reference src/SMC/PSVOwR.py:65-198), called from
 * PSVOwR.get_log_ZSMC (src/SMC/PSVOwR.py:38-62).  A sequence is owned by a cluster of
 * psvo_bsimwr_blocks(B, N, M) persistent workgroups (cooperative launch) that meet at one barrier per step.
 *
 *  inputs  : as psvo_bsim_forward (the filter's Fm, logW, lse; the hoisted backward-proposal inputs),
 *            plus  u_r (T,B,N) uniforms of the cross-chain multinomial draw (or NULL with anc_in)
 *                  anc_in (T,B,N) teacher-forced cross-chain ancestors (NULL = draw from u_r).
 *            The reference resamples after every step including t = 0 (PSVOwR.py:150-160, 187-196).
 *  outputs : bwX    (T,B,Dx,N) selected sub-particle of every chain before the cross-chain draw
 *            bwXanc (T,B,Dx,N) the resampled chains, bwXanc[t][k] = bwX[t][anc[t][k]]  (the returned
 *                              trajectories bw_Xs, PSVOwR.py:198)
 *            bwW    (T,B,N)    bw_log_W of the step (PSVOwR.py:135-142), lseW (T,B) = logsumexp_n bwW
 *                              (ELBO = sum_t lseW[t] - T log N, PSVOwR.py:144-148, 184-185)
 *            sel_out, anc_out (T,B,N) drawn sub-particle / cross-chain ancestor indices
 *            lam2_all (T,B,N,M), om_all (T,B,N,M), mu1_all (T,B,Dx,N): optional saves for
 *            psvo_bsimwr_backward.
 *  ws      : workspace, psvo_bsimwr_ws_floats(B, T, N) floats, 8-byte aligned (a two-step ring of tagged 64-bit
 *            words through which the workgroups of a sequence exchange the chains' selected states and
 *            log-weights; its last word is nonzero afterwards iff a poll timed out).
 * ------------------------------------------------------------------------------------------- */
int psvo_bsimwr_blocks(int B, int N, int M);
long long psvo_bsimwr_ws_floats(int B, int T, int N);
int psvo_bsimwr_forward(const psvo_desc* desc,
                        const float* Fm, const float* logW, const float* lse,
                        const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                        const float* sig_f, const float* sig_g, const float* sig_q1inv, const float* sig_bq2,
                        const float* bmu2, const float* minit, const float* sig_init,
                        const float* imean, const float* isig,
                        const float* obs, const float* eps_b, const float* u_b, const float* u_r,
                        const int32_t* sel_in, const int32_t* anc_in,
                        float* bwX, float* bwXanc, float* bwW, float* lseW,
                        int32_t* sel_out, int32_t* anc_out,
                        float* lam2_all, float* om_all, float* mu1_all,
                        float* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reverse mode of psvo_bsimwr_forward (TensorFlow autodiff of the loop above, src/trainer.py:115-118).
 *  inputs  : the forward call's inputs and outputs (all three saves required) and dlseW (T,B) =
 *            d loss / d lseW.
 *  outputs : rows for psvo_mlp_wgrad: xt, dFt (T,B,Dx,N,M), dGt (T,B,Dy,N,M), dmu1 (T,B,Dx,N) w.r.t.
 *            MLP_q1inv(bwXanc[t+1]);  per-workgroup partials (K = psvo_bsimwr_blocks(B, N, M), to be summed
 *            over that axis) dFm_part (T,B,K,Dx,N), dlogW_part (T,B,K,N), dlse_part (T,B,K) ->
 *            psvo_filter_backward (here the gradient w.r.t. the filter's lse is NOT zero);
 *            per-chain rows (to be summed over N) dbmu2_rows (T,B,Dx,N), dminit_rows (B,Dx,N),
 *            dimean_rows (B,Dx,N); scale gradients as psvo_bsim_backward.
 *  sacc    : workspace, B * K * psvo_bsim_acc_size(Dx, Dy) floats.
 *  ws      : workspace, psvo_bsimwr_bwd_ws_floats(B, T, N, Dx) floats, 8-byte aligned (d loss / d bwXanc exchanged
 *            between the workgroups of a sequence as tagged 64-bit words; last word nonzero iff a poll timed out).
 * ------------------------------------------------------------------------------------------- */
long long psvo_bsimwr_bwd_ws_floats(int B, int T, int N, int Dx);
int psvo_bsimwr_backward(const psvo_desc* desc,
                         const float* Fm, const float* logW, const float* lse,
                         const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                         const float* sig_f, const float* sig_g, const float* sig_q1inv, const float* sig_bq2,
                         const float* bmu2, const float* minit, const float* sig_init,
                         const float* imean, const float* isig,
                         const float* obs, const float* eps_b,
                         const float* bwXanc, const float* bwW, const float* lseW,
                         const int32_t* sel, const int32_t* anc,
                         const float* lam2_all, const float* om_all, const float* mu1_all,
                         const float* dlseW,
                         float* xt, float* dFt, float* dGt, float* dmu1,
                         float* dFm_part, float* dlogW_part, float* dlse_part,
                         float* dbmu2_rows, float* dminit_rows, float* dimean_rows,
                         float* dsig_f, float* dsig_g, float* dsig_q1inv, float* dsig_bq2, float* dsig_init,
                         float* disig, float* sacc, float* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Reverse mode of psvo_filter_forward.  The reference obtains these gradients from TensorFlow
 * autodiff of the tf.while_loop (reference src/trainer.py:115-118; no stop_gradient in src/,
 * SURVEY.md Appendix B).  One persistent workgroup per sequence walks t = T-1 .. 0; in the bootstrap wiring with
 * resampling (one hidden layer) the pass may instead run as an affine scan -- coefficients of every step in one parallel launch,
 * one or four waves per sequence for the recurrence, rows and sums in a third launch (PSVO_TUNE_FILTER_BWD; same outputs).
 *
 *  inputs  : everything psvo_filter_forward took (eps as drawn), its outputs X, Fm, P1 (NULL when
 *            bootstrap), logW, lse, idx, and the upstream gradients
 *              dlse      (T,B)              d loss / d lse[t,b]            (NULL = 0)
 *              dFm_ext   (T,B,nparts,Dx,N)  d loss / d Fm, summed over the `nparts` axis (NULL = 0)
 *              dlogW_ext (T,B,nparts,N)     d loss / d logW, likewise      (NULL = 0)
 *            (nparts = workgroups per sequence of psvo_bsim_backward, which writes partials).
 *  outputs : per-evaluation output gradients for psvo_mlp_wgrad
 *              dP (T,B,Dx,N) w.r.t. MLP_q1(X_t) (includes the transition share when bootstrap),
 *              dF (T,B,Dx,N) w.r.t. MLP_f(X_t) (only when !bootstrap), dG (T,B,Dy,N) w.r.t. MLP_g(X_t);
 *            hoisted-input gradients dmu2 (T,B,Dx), dm0 (B,Dx), dfm0 (B,Dx);
 *            scale gradients dsig_q1, dsig_q2, dsig_f, dsig0, dfsig0 (Dx), dsig_g (Dy).
 *  sacc    : workspace, psvo_filter_ws_floats(B, T, N, Dx, Dy) floats, 16-byte aligned: B * psvo_filter_acc_size(Dx, Dy)
 *            per-sequence sums, the (T,B,Dx,N) per-particle rows of d mu2 (which a parallel kernel sums over N after the
 *            time loop), the affine scan's step records and per-step partial sums.
 *  Aliasing rule: when fm0 and m0 are the SAME buffer (bootstrap and use_2_q: f_0 is q0's own density,
 *            SVO.py:86-92) d m0 receives the sum of both gradients and d fm0 is zero; likewise d sig0 / d fsig0 when
 *            fsig0 and sig0 are the same buffer.
 * ------------------------------------------------------------------------------------------- */
int psvo_filter_acc_size(int Dx, int Dy);
long long psvo_filter_ws_floats(int B, int T, int N, int Dx, int Dy);
int psvo_filter_backward(const psvo_desc* desc,
                         const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g,
                         const float* sig_q1, const float* sig_q2, const float* sig_f, const float* sig_g,
                         const float* mu2,
                         const float* m0, const float* sig0, const float* fm0, const float* fsig0,
                         const float* obs, const float* eps,
                         const float* X, const float* Fm, const float* P1, const float* logW, const float* lse,
                         const int32_t* idx,
                         const float* dlse, int nparts, const float* dFm_ext, const float* dlogW_ext,
                         float* dP, float* dF, float* dG, float* dmu2, float* dm0, float* dfm0,
                         float* dsig_q1, float* dsig_q2, float* dsig_f, float* dsig_g, float* dsig0, float* dfsig0,
                         float* sacc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One-hidden-layer MLP over plain rows (the hoisted per-(sequence, step) networks q0, q2, BSim_q2,
 * BSim_q_init): tf_mvn.mean of MLP_transformation, reference src/transformation/MLP.py:48-68 as called
 * from src/SMC/SVO.py:80-84,134-138 and src/SMC/PSVO.py:86-87,120-122.
 *   X (R,Din) row-major, out / dOut (R,Dout); Din <= 128, H in {16,32,64}, Dout <= 4.
 *   forward : out = relu(X W1 + b1) W2 + b2.
 *   backward: dX (R,Din) (NULL = not needed) and grad = [dW1 | db1 | dW2 | db2] (keras layout, flat;
 *             accumulate != 0 adds into it); partial: workspace, psvo_rows_mlp_blocks(R) * len(grad) floats.
 * ------------------------------------------------------------------------------------------- */
int psvo_rows_mlp_blocks(long long R);
int psvo_rows_mlp_forward(long long R, int Din, int H, int Dout, const float* X, const psvo_mlp* w,
                          float* out, void* stream);
int psvo_rows_mlp_backward(long long R, int Din, int H, int Dout, const float* X, const float* dOut,
                           const psvo_mlp* w, float* dX, float* partial, float* grad, int accumulate,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * Parameter gradients of a one-hidden-layer MLP from per-row output gradients:
 *   grad = [dW1 (Din,H) | db1 (H) | dW2 (H,Dout) | db2 (Dout)]  (keras layout, flat)
 *   X [S][Din][L], dOut [S][Dout][L]: S segments of L rows ((T,B,D,N) tensors: S = T*B, L = N).
 *   partial: workspace, psvo_mlp_wgrad_blocks(S*L) * len(grad) floats.  accumulate != 0 adds into grad.
 * Replaces the dense-layer gradient ops TensorFlow autodiff emits for MLP_transformation
 * (reference src/transformation/MLP.py:48-68).
 * ------------------------------------------------------------------------------------------- */
int psvo_mlp_wgrad_blocks(long long rows);
int psvo_mlp_wgrad(long long S, int L, int Din, int H, int Dout, const float* X, const float* dOut,
                   const psvo_mlp* w, float* partial, float* grad, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same for a per-particle MLP with TWO hidden layers of width H in {32, 64} (psvo_desc.layers == 2;
 * TensorFlow autodiff of MLP_transformation.transform with *_layers = "H,H", reference
 * src/transformation/MLP.py:24-38,50-54; w->Wh, w->bh required):
 *   grad = [dW1 (Din,H) | db1 (H) | dWh (H,H) | dbh (H) | dW2 (H,Dout) | db2 (Dout)]  (keras order hidden_0,
 *   hidden_1, mu_layer; flat).  Rows as psvo_mlp_wgrad.  The three H x H products per row (second-layer
 *   pre-activations, their input gradient, dWh) run on v_mfma_f32_16x16x4_f32 over LDS tiles of 64 rows.
 *   partial: workspace, psvo_mlp2_wgrad_blocks(S*L) * len(grad) floats.  accumulate != 0 adds into grad.
 * ------------------------------------------------------------------------------------------- */
int psvo_mlp2_wgrad_blocks(long long rows);
int psvo_mlp2_wgrad(long long S, int L, int Din, int H, int Dout, const float* X, const float* dOut,
                    const psvo_mlp* w, float* partial, float* grad, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One bidirectional LSTMBlockCell layer over a batch of sequences -- the observation encoder
 * upstream of the particle path.  Replaces one layer of stack_bidirectional_dynamic_rnn over
 * tf.contrib.rnn.LSTMBlockCell (reference src/SMC/SVO.py:337-341, src/model.py:164-176).
 *   x (B,T,Din); W_fw / W_bw (Din+Dh, 4Dh) TF kernels, gate order (i, j, f, o); b_* (4Dh);
 *   forget_bias = 1.  out (B,T,2Dh) = concat(forward h, backward h).
 *   cs (2,B,T,Dh) cell states and gates (2,B,T,4Dh) activated gates are optional (NULL) saves
 *   for back-propagation through time.  Dh in {8,16,32,64}, Din <= 128.
 * ------------------------------------------------------------------------------------------- */
int psvo_bilstm_forward(int B, int T, int Din, int Dh, const float* x,
                        const float* W_fw, const float* b_fw, const float* W_bw, const float* b_bw,
                        float* out, float* cs, float* gates, void* stream);

/* Back-propagation through time of psvo_bilstm_forward (needs its cs / gates saves).
 *   dout (B,T,2Dh) = d loss / d out.  Outputs: dx_part (2,B,T,Din) per-direction input gradients
 *   (sum over axis 0), dW_part (B,2,Din+Dh,4Dh) and db_part (B,2,4Dh) per-sequence weight-gradient
 *   partials (sum over axis 0; index 0 = forward direction, 1 = backward). */
int psvo_bilstm_backward(int B, int T, int Din, int Dh, const float* x,
                         const float* W_fw, const float* W_bw,
                         const float* out, const float* cs, const float* gates, const float* dout,
                         float* dx_part, float* dW_part, float* db_part, void* stream);

/* Folds the per-sequence partials of psvo_bilstm_backward into the two cells' gradients, one launch, fixed order:
 *   g_fw / g_bw = [kernel (Din+Dh, 4Dh) | bias (4Dh)] of the forward / backward cell; accumulate != 0 adds into them. */
int psvo_bilstm_wgrad_fold(int B, int Din, int Dh, const float* dW_part, const float* db_part,
                           float* g_fw, float* g_bw, int accumulate, void* stream);

/* Fused Adam update of one flat fp32 parameter vector: tf.train.AdamOptimizer(lr).minimize(-log_ZSMC)
 * of the reference (src/trainer.py:115-118), TF 1.12 epsilon-hat form.  `step` counts from 1;
 * `grad_scale` multiplies the gradient first (-1/world_size for a summed all-reduce of d log_ZSMC). */
int psvo_adam_step(float* params, const float* grads, float* m, float* v, long long n, float lr,
                   float beta1, float beta2, float eps, long long step, float grad_scale, void* stream);

/* out[p] (+)= sum_{r < nrows} part[r * stride + p], p < n: folds per-sequence / per-workgroup partial
 * gradients straight into the flat gradient buffer (deterministic, no atomics). */
int psvo_reduce_rows(const float* part, int nrows, long long stride, int n, float* out, int accumulate,
                     void* stream);

/* Scale vectors of all distributions at once: sigma = max(softplus(raw), min), NaN -> 0 first
 * (tf_mvn.get_sigma, reference src/distribution/mvn.py:80-90) and its gradient
 * graw (+)= dsig * sigmoid(raw) * [softplus(raw) >= min]. */
int psvo_sigma_forward(const float* raw, const float* mins, float* sig, int n, void* stream);
int psvo_sigma_backward(const float* raw, const float* mins, const float* dsig, float* graw, int n,
                        int accumulate, void* stream);

/* Diagnostic: runs the DPP / permlane cross-lane primitives the kernels are built on over one
 * wavefront of input (64 floats) and writes 9 x 64 results (xor 1..32, inclusive scan, sum, max). */
/* ---------------------------------------------------------------------------------------------
 * Dense layers over plain rows on v_mfma_f32_16x16x4_f32 (exact f32): the hoisted networks q0, q2, BSim_q2, BSim_q_init
 * (reference src/transformation/MLP.py:24-68) when they have more than one hidden layer or exceed psvo_rows_mlp_*'s
 * widths (Din <= 4096, Dout <= 4096 here).  Rows X (R, Din) row-major, W (Din, Dout) keras layout, b (Dout).
 *   forward : Y = relu ? max(X W + b, 0) : X W + b
 *   backward: dZ = relu ? dY * [Y > 0] : dY;  dX = dZ W^T (R, Din; may be null);  grad = [dW (Din, Dout) | db (Dout)],
 *             accumulated into `grad` when `accumulate`.  partial: psvo_dense_wgrad_slices(R) * (Din + 1) * Dout floats.
 * ------------------------------------------------------------------------------------------- */
int psvo_dense_wgrad_slices(long long R);
int psvo_dense_forward(long long R, int Din, int Dout, const float* X, const float* W, const float* b, int relu,
                       float* Y, void* stream);
int psvo_dense_backward(long long R, int Din, int Dout, const float* X, const float* Y, const float* dY, const float* W,
                        int relu, float* dX, float* partial, float* grad, int accumulate, void* stream);

/* Measurement aid: a one-thread launch that stores the device's constant-rate wall clock (100 MHz) into *slot when the
 * stream reaches it.  Captured into a hipGraph beside the real launches it gives the timeline of a REPLAYED step, which
 * HIP events cannot (event records inside a capture cannot be timed on ROCm) -- tools/replay_timeline.py. */
int psvo_debug_stamp(unsigned long long* slot, void* stream);

int psvo_selftest_lanes(const float* in64, float* out576, void* stream);
/* second self-test: swap-add stages over lane bits 5 / 4, 16-lane row sum, and the operand / accumulator layout of
 * v_mfma_f32_16x16x4_f32 as bsim_bwd2 uses it.  in: 3 x 64 floats (lo, hi, extra); out: 5 x 64 floats
 * (swap_add32, swap_add16, row_sum16, mfma D register 0, mfma D register 1). */
int psvo_selftest_lanes2(const float* in192, float* out320, void* stream);

/* Per-sequence ELBO reductions (no batch mean: the caller averages, so a batch shard can be
 * all-reduced).  filter: out[b] = sum_t lse[t,b] (SVO.compute_log_ZSMC, SVO.py:302-311);
 * bsim: out[b] = logsumexp_n score[b,n] - log N (PSVO.compute_log_ZSMC, PSVO.py:52-67). */
int psvo_elbo_filter(const psvo_desc* desc, const float* lse, float* out, void* stream);
int psvo_elbo_bsim(const psvo_desc* desc, const float* score, float* out, void* stream);

/* The PSVO training objective in one launch and its reverse: out[0] = mean_b psvo_elbo_bsim[b]
 * (PSVO.py:52-67 including the reduce_mean), dscore[b,n] = dz[0] / B * softmax_n(score[b,:]).
 * B is the local batch: a batch shard differentiates its own mean and the gradient all-reduce averages the shards. */
int psvo_elbo_bsim_mean(const psvo_desc* desc, const float* score, float* out, void* stream);
int psvo_elbo_bsim_mean_backward(const psvo_desc* desc, const float* score, const float* dz, float* dscore,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSVO_HIP_H */
