// Reverse-mode pass of the forward particle filter (gradient of anything computed from the
// filter's logW / lse / Fm outputs w.r.t. its inputs).  The reference gets this from TensorFlow
// autodiff through tf.while_loop (reference src/trainer.py:115-118, no stop_gradient anywhere,
// SURVEY.md Appendix B); here it is one persistent workgroup per sequence walking t = T-1 .. 0.
//
// What flows (Appendix B): samples are reparameterised (x = mu + c*eps), ancestor indices are
// constants, the gather of resampled particles back-propagates as a scatter-add into the
// pre-resampling particles.  In this formulation a resampled particle only enters step t+1
// through the gathered MLP outputs P1_t[a] (proposal mean) and Fm_t[a] (transition mean), so the
// scatter-add targets are d P1_t / d Fm_t, kept in LDS (float atomics), double-buffered over t.
//
// Nothing is stored by the forward pass beyond its own outputs; hidden activations are
// recomputed.  MLP *weight* gradients are not formed here: the kernel writes, for every MLP
// evaluation, the gradient w.r.t. that evaluation's output (rows dP, dF, dG) and psvo_mlp_wgrad
// reduces rows to weights in a fully parallel launch.
#include "common.h"

namespace psvo {
inline namespace PSVO_LNS {   // l1 / l2: hidden layers of the per-particle MLPs (common.h)
PSVO_TIMERS_DEFINE(filter_bwd)


struct FilterBwdArgs {
    int B, T, N;
    int resample, two_q, bootstrap, emission;
    psvo_mlp q1, f, g;
    const float *sig_q1, *sig_q2, *sig_f, *sig_g;
    const float *mu2, *m0, *sig0, *fm0, *fsig0, *obs, *eps;
    const float *X, *Fm, *P1, *logW, *lse;
    const int32_t* idx;
    const float* dlse;       // (T,B) or null
    int nparts;              // leading "parts" dimension of the external gradients (bsim blocks per sequence)
    const float* dFm_ext;    // (T,B,nparts,Dx,N) or null
    const float* dlogW_ext;  // (T,B,nparts,N) or null
    float *dP, *dF, *dG, *dmu2, *dm0, *dfm0;
    float* sacc;             // (B, NACC) per-sequence scalar accumulators (see finalize)
    float* dm2_rows;         // (T,B,Dx,N) per-particle d mu2 rows, summed over N by row_sum_kernel afterwards
    int wave_copies;         // filter_bwd_kernel: one scatter-target copy per wave (set by the launcher when LDS has room)
    float* scanAB;           // affine-scan path: (T,B,N,REC) records (16-byte aligned); part: (T,B,NACC) per-step partial sums
    float* scanPart;
};

template <int DX, int DY>
struct FAcc {
    // per-dimension sums, one set for t >= 1 and one for t = 0
    static constexpr int kSc = 0;            // direct d c
    static constexpr int kSmm1 = DX;         // sum dmu * mean1
    static constexpr int kSmb = 2 * DX;      // sum dmu * mu2
    static constexpr int kSmm = 3 * DX;      // sum dmu * mu
    static constexpr int kSfs = 4 * DX;      // direct d (transition scale)
    static constexpr int kSet = 5 * DX;
    static constexpr int kSg = 2 * kSet;     // d sigma_g (DY)
    static constexpr int kN = 2 * kSet + DY;
};

template <int DX>
struct BStepK {
    float c[DX], ic[DX], i1[DX], i2[DX], ifs[DX];
};

template <int DX>
__device__ __forceinline__ BStepK<DX> make_bstepk(const float* s1, const float* s2, const float* fs, bool two_q) {
    BStepK<DX> K;
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        K.i1[d] = 1.f / s1[d];
        if (two_q) {
            K.i2[d] = 1.f / s2[d];
            K.ic[d] = K.i1[d] + K.i2[d];
            K.c[d] = 1.f / K.ic[d];
        } else {
            K.i2[d] = 0.f;
            K.ic[d] = K.i1[d];
            K.c[d] = s1[d];
        }
        K.ifs[d] = 1.f / fs[d];
    }
    return K;
}

// sum of `v` over the whole workgroup, result in every lane; `red` has 16 floats
__device__ __forceinline__ float block_sum(float v, float* red, int wave, int lane, int nw) {
    v = wave_sum(v);
    if (nw == 1) return v;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    __syncthreads();
    return s;
}

template <int DX, int DY, int H, int MAXT>
__global__ void __launch_bounds__(MAXT) filter_bwd_kernel(const FilterBwdArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    using AC = FAcc<DX, DY>;
    // one wave per SIMD (<= 256 lanes): let the compiler keep the loop-invariant MLP weights in VGPRs
    constexpr bool kRolled = (MAXT > 256) || (MlpLds<DX, H, DX, PSVO_L>::kSize + MlpLds<DX, H, DY, PSVO_L>::kSize > 330);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, nw = NT >> 6;
    const int b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const bool valid = tid < N;
    const int n = valid ? tid : N - 1;

    float* wq1 = smem;
    float* wf = wq1 + MQ::kSize;
    float* wg = wf + MQ::kSize;
    // scatter targets: one copy per wave when the launcher found room for them (a.wave_copies; see filter_bwd_lpp_kernel) --
    // fixed summation order --, else one shared copy and float atomics across the waves
    const int nc = a.wave_copies ? nw : 1;
    const int CP = DX * NT;               // floats of one copy
    float* accP = wg + MG::kSize;         // [2][nc][DX][NT] scatter targets d P1 (+ d Fm when bootstrap)
    float* accF = accP + 2 * nc * CP;     // [2][nc][DX][NT] d Fm (allocated only when !bootstrap)
    float* red = accF + (a.bootstrap ? 0 : 2 * nc * CP);      // 16 floats

    MQ::load(wq1, a.q1, tid, NT);
    if (!a.bootstrap) MQ::load(wf, a.f, tid, NT);
    MG::load(wg, a.g, tid, NT);
    const float* wfm = a.bootstrap ? wq1 : wf;
    for (int i = tid; i < (a.bootstrap ? 2 : 4) * nc * CP; i += NT) accP[i] = 0.f;

    float sq1[DX], sq2[DX], sfv[DX], s0[DX], fs0[DX], isg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        sq1[d] = a.sig_q1[d];
        sq2[d] = a.two_q ? a.sig_q2[d] : 1.f;
        sfv[d] = a.bootstrap ? a.sig_q1[d] : a.sig_f[d];
        s0[d] = a.sig0[d];
        fs0[d] = a.fsig0[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) isg[e] = 1.f / a.sig_g[e];
    const BStepK<DX> K0 = make_bstepk<DX>(s0, sq2, fs0, a.two_q != 0);
    const BStepK<DX> K1 = make_bstepk<DX>(sq1, sq2, sfv, a.two_q != 0);

    float acc[AC::kN];
#pragma unroll
    for (int i = 0; i < AC::kN; ++i) acc[i] = 0.f;
    float dlnw = 0.f;  // IWAE: gradient w.r.t. the normalised log-weight carried into step t+1

    // Per-step inputs are software-prefetched: everything of step t-1 is requested at the top of step t,
    // and the ancestor index it gathers through (idx[t-2]) one step earlier still, so the dependent
    // gather Fm[t-2][idx] never sits on the critical path (two HBM round trips per step otherwise).
    // (plain arrays + scalars rather than a struct: a struct of arrays copied per step ends up in scratch)
    float m0r[DX], fm0r[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        m0r[d] = a.m0[b * DX + d];
        fm0r[d] = a.fm0[b * DX + d];
    }
    auto load_anc = [&](int t) -> int {  // ancestor of particle n at step t (its parent lives at t-1)
        return (t >= 1 && a.resample) ? a.idx[((size_t)(t - 1) * B + b) * N + n] : n;
    };
    // load_step only ISSUES loads into raw registers: every select / add on the loaded values happens when the
    // step consumes them, one whole step later, so no s_waitcnt lands in the prefetch.  (Upstream partials with
    // nparts > 1 are summed here with dependent adds -- the host wrapper passes them already folded, nparts = 1.)
    const bool one_part = (a.nparts == 1);
    auto load_step = [&](int t, int anc, float (&sx)[DX], float (&se)[DX], float (&sm2)[DX], float (&sy)[DY],
                         float (&smean1)[DX], float (&sfmean)[DX], float (&sdfm)[DX], float (&ssc)[4]) {
        const size_t tb = (size_t)t * B + b;
        const size_t tp = (t == 0) ? tb : tb - B;              // (t = 0 has no parent: value replaced at use)
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            sx[d] = a.X[(tb * DX + d) * N + n];
            se[d] = a.eps[(tb * DX + d) * N + n];
            sm2[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
            sfmean[d] = a.Fm[(tp * DX + d) * N + anc];
            smean1[d] = a.bootstrap ? 0.f : a.P1[(tp * DX + d) * N + anc];
            if (a.dFm_ext) {
                if (one_part) {
                    sdfm[d] = a.dFm_ext[(tb * DX + d) * N + n];
                } else {
                    float ext = 0.f;
                    for (int p = 0; p < a.nparts; ++p) ext += a.dFm_ext[((tb * a.nparts + p) * DX + d) * N + n];
                    sdfm[d] = ext;
                }
            } else {
                sdfm[d] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) sy[k] = a.obs[tb * DY + k];
        ssc[0] = a.logW[tb * N + n];
        ssc[1] = a.lse[tb];
        ssc[2] = a.dlse ? a.dlse[tb] : 0.f;
        if (a.dlogW_ext) {
            if (one_part) {
                ssc[3] = a.dlogW_ext[tb * N + n];
            } else {
                float ext = 0.f;
                for (int p = 0; p < a.nparts; ++p) ext += a.dlogW_ext[(tb * a.nparts + p) * N + n];
                ssc[3] = ext;
            }
        } else {
            ssc[3] = 0.f;
        }
    };
    float c_x[DX], c_e[DX], c_m2[DX], c_y[DY], c_mean1[DX], c_fmean[DX], c_dfm[DX], c_sc[4];
    int c_anc = load_anc(T - 1);
    int anc_next = load_anc(T - 2);            // ancestor used by step T-2
    load_step(T - 1, c_anc, c_x, c_e, c_m2, c_y, c_mean1, c_fmean, c_dfm, c_sc);
    __syncthreads();

    SEC_INIT(filter_bwd)
    for (int t = T - 1; t >= 0; --t) {
        SEC(0);
        const size_t tb = (size_t)t * B + b;
        const bool first = (t == 0);
        const BStepK<DX> K = first ? K0 : K1;
        float inc[AC::kSet];  // this step's contribution to the per-dimension sums (set chosen below)
#pragma unroll
        for (int i = 0; i < AC::kSet; ++i) inc[i] = 0.f;
        const int mine = a.wave_copies ? wave * CP : 0;
        float* curP = accP + (t & 1) * nc * CP;                     // all copies of this step's targets
        float* nxtP = accP + ((t + 1) & 1) * nc * CP + mine;        // this wave's copy (or the shared one) for step t - 1
        float* curF = accF + (t & 1) * nc * CP;
        float* nxtF = accF + ((t + 1) & 1) * nc * CP + mine;

        // ---- forward quantities of this step (prefetched) ----------------------------------------------
        float n_x[DX], n_e[DX], n_m2[DX], n_y[DY], n_mean1[DX], n_fmean[DX], n_dfm[DX], n_sc[4];
        int n_anc = n;
        if (t >= 1) {
            n_anc = anc_next;
            load_step(t - 1, n_anc, n_x, n_e, n_m2, n_y, n_mean1, n_fmean, n_dfm, n_sc);
            anc_next = load_anc(t - 2);
        }
        SEC(1);   // issue of the prefetch loads
        // this step's scatter targets (complete since the barrier that ended step t + 1): requested here, ahead of their use
        float sp[DX], sf[DX];
        if (nc == 1) {           // (one shared copy, float atomics: N > 256)
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                sp[d] = curP[d * NT + tid];
                sf[d] = a.bootstrap ? 0.f : curF[d * NT + tid];
            }
        } else {
            float cp[8][DX];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int wc = w < nc ? w : nc - 1;       // (unconditional reads of a clamped copy + select)
#pragma unroll
                for (int d = 0; d < DX; ++d) cp[w][d] = curP[wc * CP + d * NT + tid];
            }
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                sp[d] = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) sp[d] += (w < nc) ? cp[w][d] : 0.f;       // in wave order
                sf[d] = 0.f;
            }
        }
        if (nc > 1 && !a.bootstrap) {
            float cf[8][DX];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int wc = w < nc ? w : nc - 1;
#pragma unroll
                for (int d = 0; d < DX; ++d) cf[w][d] = curF[wc * CP + d * NT + tid];
            }
#pragma unroll
            for (int d = 0; d < DX; ++d)
#pragma unroll
                for (int w = 0; w < 8; ++w) sf[d] += (w < nc) ? cf[w][d] : 0.f;
        }
        float x[DX], e[DX], m2[DX], y[DY], mean1[DX], fmean[DX];
        const int anc = c_anc;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            x[d] = c_x[d];
            e[d] = c_e[d];
            m2[d] = c_m2[d];
            // parents: gathered MLP outputs of step t-1 (bootstrap: MLP_q1 == MLP_f), or the t = 0 features
            fmean[d] = first ? fm0r[d] : c_fmean[d];
            mean1[d] = first ? m0r[d] : (a.bootstrap ? c_fmean[d] : c_mean1[d]);
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = c_y[k];
        float mu[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d)
            mu[d] = a.two_q ? K.c[d] * fmaf(K.i1[d], mean1[d], K.i2[d] * m2[d]) : mean1[d];

        // ---- gradient w.r.t. logW_t[n] ------------------------------------------------------------
        const float sm = valid ? exp2_fast((c_sc[0] - c_sc[1]) * kLog2e) : 0.f;
        float dlw = c_sc[2] * sm + c_sc[3];
        if (!a.resample) {
            const float tot = block_sum(dlnw, red, wave, lane, nw);
            dlw += dlnw - sm * tot;
        }
        if (!valid) dlw = 0.f;
        dlnw = first ? 0.f : dlw;

        SEC(2);   // d logW
        // ---- emission ---------------------------------------------------------------------------
        float dx[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dx[d] = 0.f;
        {
            float gm[DY], dgm[DY];
            MG::template eval<kRolled>(wg, x, gm);
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                float dmean = 1.f;
                if (a.emission) { dmean = emis_dmean(gm[k]); gm[k] = emis_mean(gm[k]); }
                const float z = (y[k] - gm[k]) * isg[k];
                dgm[k] = dlw * z * isg[k] * dmean;
                acc[AC::kSg + k] += dlw * (z * z - 1.f) * isg[k];
                if (valid) a.dG[(tb * DY + k) * N + n] = dgm[k];
            }
            MG::template bwd_input<kRolled>(wg, x, dgm, dx);
        }
        SEC(3);   // MLP_g forward + input gradient
        // ---- transition and proposal densities ------------------------------------------------------
        float dfmean[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float z = (x[d] - fmean[d]) * K.ifs[d];
            const float tf = dlw * z * K.ifs[d];
            dx[d] -= tf;
            dfmean[d] = tf;
            inc[AC::kSfs + d] += dlw * (z * z - 1.f) * K.ifs[d];
            inc[AC::kSc + d] += dlw * K.ic[d];  // -dq/dc = +dlw / c
        }
        // ---- MLP_q1(x_t), MLP_f(x_t): gradients scattered here by step t+1 (+ backward simulation) --------
        float dPn[DX], dFn[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            dPn[d] = sp[d];
            for (int w = 0; w < nc; ++w) curP[w * CP + d * NT + tid] = 0.f;
            const float ext = c_dfm[d];
            if (a.bootstrap) {
                dPn[d] += ext;
                dFn[d] = 0.f;
            } else {
                dFn[d] = sf[d] + ext;
                for (int w = 0; w < nc; ++w) curF[w * CP + d * NT + tid] = 0.f;
            }
            if (!valid) {
                dPn[d] = 0.f;
                dFn[d] = 0.f;
            }
        }
        if (valid) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                a.dP[(tb * DX + d) * N + n] = dPn[d];
                if (!a.bootstrap) a.dF[(tb * DX + d) * N + n] = dFn[d];
            }
        }
        SEC(4);   // densities, scatter targets from LDS, row stores
        MQ::template bwd_input<kRolled>(wq1, x, dPn, dx);
        if (!a.bootstrap) MQ::template bwd_input<kRolled>(wfm, x, dFn, dx);
        SEC(5);   // MLP_q1 (/ MLP_f) input gradient

        // ---- x = mu + c eps, mu = c (mean1/s1 + mu2/s2) ----------------------------------------------------
        float dmean1[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float dmu = dx[d];
            inc[AC::kSc + d] += dmu * e[d];
            float dm2 = 0.f;
            if (a.two_q) {
                dmean1[d] = dmu * K.c[d] * K.i1[d];
                dm2 = dmu * K.c[d] * K.i2[d];
                inc[AC::kSmm1 + d] += dmu * mean1[d];
                inc[AC::kSmb + d] += dmu * m2[d];
                inc[AC::kSmm + d] += dmu * mu[d];
                // summed over the particles by a parallel kernel after the time loop: a block sum here would put
                // a wave reduction, an LDS round trip and a barrier into every step of the serial chain
                if (valid) a.dm2_rows[(tb * DX + d) * N + n] = dm2;
            } else {
                dmean1[d] = dmu;
            }
        }
        SEC(6);   // PoG gradient, block sums of d mu2
        // ---- gather backward: scatter-add into the parents (SVO.py:255-257) ------------------------------
        if (!first) {
            if (valid) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    if (a.bootstrap) {
                        atomicAdd(&nxtP[d * NT + anc], dmean1[d] + dfmean[d]);
                    } else {
                        atomicAdd(&nxtP[d * NT + anc], dmean1[d]);
                        atomicAdd(&nxtF[d * NT + anc], dfmean[d]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float s1 = block_sum(dmean1[d], red, wave, lane, nw);
                const float s2 = block_sum(dfmean[d], red, wave, lane, nw);
                if (tid == 0) {   // (fm0 aliasing m0: ONE tensor feeds both, its gradient is the sum -- see psvo_hip.h)
                    const bool same0 = (a.fm0 == a.m0);
                    a.dm0[b * DX + d] = same0 ? s1 + s2 : s1;
                    a.dfm0[b * DX + d] = same0 ? 0.f : s2;
                }
            }
        }
        SEC(7);   // LDS scatter-add
        if (t >= 1) {
            c_anc = n_anc;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                c_x[d] = n_x[d]; c_e[d] = n_e[d]; c_m2[d] = n_m2[d];
                c_mean1[d] = n_mean1[d]; c_fmean[d] = n_fmean[d]; c_dfm[d] = n_dfm[d];
            }
#pragma unroll
            for (int k = 0; k < DY; ++k) c_y[k] = n_y[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) c_sc[k] = n_sc[k];
        }
#pragma unroll
        for (int i = 0; i < AC::kSet; ++i) {  // set 0: t = 0, set 1: t >= 1 (static register indices)
            acc[i] += first ? inc[i] : 0.f;
            acc[AC::kSet + i] += first ? 0.f : inc[i];
        }
        __syncthreads();
        SEC(8);   // register rotation (waits for the prefetch) + barrier
    }

    // ---- per-sequence scalar accumulators ---------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < AC::kN; ++i) {
        const float s = block_sum(acc[i], red, wave, lane, nw);
        if (tid == 0) a.sacc[(size_t)b * AC::kN + i] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Four lanes per particle (N <= 128), mirror of filter_fwd_lpp_kernel: lane p of the quad owns hidden units
// [p H/4, (p+1) H/4) of every MLP (forward recompute and input gradient), the partial input gradients are summed
// over the quad with two DPP adds; everything that is not an MLP is computed redundantly in the four lanes and
// counted once (lane p == 0) in the sums.  The reverse filter is on the critical path of a training step and its
// step is a dependent chain, so shortening the two MLP input-gradient chains is what matters.
// ---------------------------------------------------------------------------------------------
template <int DX, int DY, int H>
__global__ void __launch_bounds__(512) filter_bwd_lpp_kernel(const FilterBwdArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    using AC = FAcc<DX, DY>;
    constexpr int P = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, nw = NT >> 6;
    const int NPT = NT / P;
    const int b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const int pn = tid >> 2, p = tid & 3;
    const bool valid = pn < N;
    const bool one = valid && p == 0;          // the lane of the quad that counts in sums over particles
    const int n = valid ? pn : N - 1;

    float* wq1 = smem;
    float* wf = wq1 + MQ::kSize;
    float* wg = wf + MQ::kSize;
    // Scatter targets of the resampling gather's reverse (children add into their parent): ONE COPY PER WAVE.  Float atomics
    // from several waves on one LDS word are applied in arrival order, so the sums -- and with them every gradient of the
    // step -- differed in the last bits from run to run (round-2 review).  Within a wave the atomics of one instruction are
    // resolved in lane order and instructions in program order; with a copy per wave and the parent adding the copies in
    // wave order, the order of every sum is fixed.  Cost: nw reads + clears per value where there was one (~1 % of the
    // kernel); LDS 2 x nw x DX x NPT floats (16 KB at Dx = 2, N = 128).
    const int CP = DX * NPT;              // floats of one copy
    float* accP = wg + MG::kSize;         // [2][nw][DX][NPT] d P1 (+ d Fm when bootstrap)
    float* accF = accP + 2 * nw * CP;     // [2][nw][DX][NPT] d Fm (only when !bootstrap)
    float* red = accF + 2 * nw * CP;      // 16 floats

    MQ::load(wq1, a.q1, tid, NT);
    if (!a.bootstrap) MQ::load(wf, a.f, tid, NT);
    MG::load(wg, a.g, tid, NT);
    const float* wfm = a.bootstrap ? wq1 : wf;
    for (int i = tid; i < 4 * nw * CP; i += NT) accP[i] = 0.f;

    float sq1[DX], sq2[DX], sfv[DX], s0[DX], fs0[DX], isg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        sq1[d] = a.sig_q1[d];
        sq2[d] = a.two_q ? a.sig_q2[d] : 1.f;
        sfv[d] = a.bootstrap ? a.sig_q1[d] : a.sig_f[d];
        s0[d] = a.sig0[d];
        fs0[d] = a.fsig0[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) isg[e] = 1.f / a.sig_g[e];
    const BStepK<DX> K0 = make_bstepk<DX>(s0, sq2, fs0, a.two_q != 0);
    const BStepK<DX> K1 = make_bstepk<DX>(sq1, sq2, sfv, a.two_q != 0);

    float acc[AC::kN];
#pragma unroll
    for (int i = 0; i < AC::kN; ++i) acc[i] = 0.f;
    float dlnw = 0.f;

    float m0r[DX], fm0r[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        m0r[d] = a.m0[b * DX + d];
        fm0r[d] = a.fm0[b * DX + d];
    }
    auto load_anc = [&](int t) -> int {
        return (t >= 1 && a.resample) ? a.idx[((size_t)(t - 1) * B + b) * N + n] : n;
    };
    const bool one_part = (a.nparts == 1);
    // issue-only prefetch (see filter_bwd_kernel)
    auto load_step = [&](int t, int anc, float (&sx)[DX], float (&se)[DX], float (&sm2)[DX], float (&sy)[DY],
                         float (&smean1)[DX], float (&sfmean)[DX], float (&sdfm)[DX], float (&ssc)[4]) {
        const size_t tb = (size_t)t * B + b;
        const size_t tp = (t == 0) ? tb : tb - B;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            sx[d] = a.X[(tb * DX + d) * N + n];
            se[d] = a.eps[(tb * DX + d) * N + n];
            sm2[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
            sfmean[d] = a.Fm[(tp * DX + d) * N + anc];
            smean1[d] = a.bootstrap ? 0.f : a.P1[(tp * DX + d) * N + anc];
            if (a.dFm_ext) {
                if (one_part) {
                    sdfm[d] = a.dFm_ext[(tb * DX + d) * N + n];
                } else {
                    float ext = 0.f;
                    for (int q = 0; q < a.nparts; ++q) ext += a.dFm_ext[((tb * a.nparts + q) * DX + d) * N + n];
                    sdfm[d] = ext;
                }
            } else {
                sdfm[d] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) sy[k] = a.obs[tb * DY + k];
        ssc[0] = a.logW[tb * N + n];
        ssc[1] = a.lse[tb];
        ssc[2] = a.dlse ? a.dlse[tb] : 0.f;
        if (a.dlogW_ext) {
            if (one_part) {
                ssc[3] = a.dlogW_ext[tb * N + n];
            } else {
                float ext = 0.f;
                for (int q = 0; q < a.nparts; ++q) ext += a.dlogW_ext[(tb * a.nparts + q) * N + n];
                ssc[3] = ext;
            }
        } else {
            ssc[3] = 0.f;
        }
    };
    float c_x[DX], c_e[DX], c_m2[DX], c_y[DY], c_mean1[DX], c_fmean[DX], c_dfm[DX], c_sc[4];
    int c_anc = load_anc(T - 1);
    int anc_next = load_anc(T - 2);
    load_step(T - 1, c_anc, c_x, c_e, c_m2, c_y, c_mean1, c_fmean, c_dfm, c_sc);
    __syncthreads();

    SEC_INIT(filter_bwd)
    // t = 0 (own step constants, block sums instead of the scatter) is peeled out of the loop: see filter_fwd_lpp_kernel
    auto step = [&](auto first_tag, const int t) {
        SEC(0);   // (lpp) loop overhead
        const size_t tb = (size_t)t * B + b;
        constexpr bool first = decltype(first_tag)::value;
        const BStepK<DX> K = first ? K0 : K1;
        float inc[AC::kSet];
#pragma unroll
        for (int i = 0; i < AC::kSet; ++i) inc[i] = 0.f;
        float* curP = accP + (t & 1) * nw * CP;                      // all waves' copies of this step's targets
        float* nxtP = accP + ((t + 1) & 1) * nw * CP + wave * CP;    // this wave's copy for step t - 1
        float* curF = accF + (t & 1) * nw * CP;
        float* nxtF = accF + ((t + 1) & 1) * nw * CP + wave * CP;

        float n_x[DX], n_e[DX], n_m2[DX], n_y[DY], n_mean1[DX], n_fmean[DX], n_dfm[DX], n_sc[4];
        int n_anc = n;
        if (t >= 1) {
            n_anc = anc_next;
            load_step(t - 1, n_anc, n_x, n_e, n_m2, n_y, n_mean1, n_fmean, n_dfm, n_sc);
            anc_next = load_anc(t - 2);
        }
        SEC(1);   // (lpp) issue of the prefetch loads
        // the scatter targets of this step (complete since the barrier that ended step t + 1) are requested HERE, a whole
        // MLP pass ahead of their use, so that the nw reads per value cost issue slots only
        float sp[DX], sf[DX];
        {
            float cp[8][DX];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                // (unconditional reads of a clamped copy + select: a guarded load is a branch, and a wait, per copy)
                const int wc = w < nw ? w : nw - 1;
#pragma unroll
                for (int d = 0; d < DX; ++d) cp[w][d] = curP[wc * CP + d * NPT + pn];
            }
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                sp[d] = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) sp[d] += (w < nw) ? cp[w][d] : 0.f;        // the waves' copies, in wave order
                sf[d] = 0.f;
            }
        }
        if (!a.bootstrap) {
            float cf[8][DX];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int wc = w < nw ? w : nw - 1;
#pragma unroll
                for (int d = 0; d < DX; ++d) cf[w][d] = curF[wc * CP + d * NPT + pn];
            }
#pragma unroll
            for (int d = 0; d < DX; ++d)
#pragma unroll
                for (int w = 0; w < 8; ++w) sf[d] += (w < nw) ? cf[w][d] : 0.f;
        }
        float x[DX], e[DX], m2[DX], y[DY], mean1[DX], fmean[DX];
        const int anc = c_anc;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            x[d] = c_x[d];
            e[d] = c_e[d];
            m2[d] = c_m2[d];
            fmean[d] = first ? fm0r[d] : c_fmean[d];
            mean1[d] = first ? m0r[d] : (a.bootstrap ? c_fmean[d] : c_mean1[d]);
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = c_y[k];
        float mu[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d)
            mu[d] = a.two_q ? K.c[d] * fmaf(K.i1[d], mean1[d], K.i2[d] * m2[d]) : mean1[d];

        // ---- gradient w.r.t. logW_t[n] (the same in the four lanes of the particle) ---------------------------------
        const float sm = valid ? exp2_fast((c_sc[0] - c_sc[1]) * kLog2e) : 0.f;
        float dlw = c_sc[2] * sm + c_sc[3];
        if (!a.resample) {
            const float tot = block_sum(one ? dlnw : 0.f, red, wave, lane, nw);
            dlw += dlnw - sm * tot;
        }
        if (!valid) dlw = 0.f;
        dlnw = first ? 0.f : dlw;
        const float cnt = (p == 0) ? 1.f : 0.f;     // sums over particles take the quad's first lane

        SEC(2);   // (lpp) proposal mean, d logW
        // ---- emission: hidden units of MLP_g split over the quad ----------------------------------------------------
        float dxp[DX];                              // this lane's PARTIAL of d x (summed over the quad below)
#pragma unroll
        for (int d = 0; d < DX; ++d) dxp[d] = 0.f;
        {
            float gm[DY], dgm[DY];
            MG::template eval_part<P>(wg, p, x, gm);
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                gm[k] = group_sum<P>(gm[k]);
                float dmean = 1.f;
                if (a.emission) { dmean = emis_dmean(gm[k]); gm[k] = emis_mean(gm[k]); }
                const float z = (y[k] - gm[k]) * isg[k];
                dgm[k] = dlw * z * isg[k] * dmean;
                acc[AC::kSg + k] += cnt * dlw * (z * z - 1.f) * isg[k];
                if (valid && p == 2) a.dG[(tb * DY + k) * N + n] = dgm[k];
            }
            MG::template bwd_input_part<P>(wg, p, x, dgm, dxp);
        }
        SEC(3);   // (lpp) MLP_g forward, quad sum, input gradient, dG row store
        // ---- transition and proposal densities ------------------------------------------------------------------------
        float dfmean[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float z = (x[d] - fmean[d]) * K.ifs[d];
            const float tf = dlw * z * K.ifs[d];
            dxp[d] -= cnt * tf;
            dfmean[d] = tf;
            inc[AC::kSfs + d] += dlw * (z * z - 1.f) * K.ifs[d];
            inc[AC::kSc + d] += dlw * K.ic[d];
        }
        // ---- MLP_q1(x_t), MLP_f(x_t): gradients scattered here by step t+1 (+ backward simulation) -----------------------
        float dPn[DX], dFn[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            dPn[d] = sp[d];
            const float ext = c_dfm[d];
            if (a.bootstrap) {
                dPn[d] += ext;
                dFn[d] = 0.f;
            } else {
                dFn[d] = sf[d] + ext;
            }
            if (!valid) {
                dPn[d] = 0.f;
                dFn[d] = 0.f;
            }
        }
        // (only this quad reads these entries, in the wave instructions above: clear them for step t-2; lane p takes the
        //  copies w = p, p + 4)
        for (int w = p; w < nw; w += 4) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                curP[w * CP + d * NPT + pn] = 0.f;
                if (!a.bootstrap) curF[w * CP + d * NPT + pn] = 0.f;
            }
        }
        if (valid) {
            if (p == 0) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.dP[(tb * DX + d) * N + n] = dPn[d];
            } else if (p == 1 && !a.bootstrap) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.dF[(tb * DX + d) * N + n] = dFn[d];
            }
        }
        SEC(4);   // (lpp) densities, scatter targets from LDS, dP / dF row stores
        MQ::template bwd_input_part<P>(wq1, p, x, dPn, dxp);
        if (!a.bootstrap) MQ::template bwd_input_part<P>(wfm, p, x, dFn, dxp);
        SEC(5);   // (lpp) MLP_q1 (/ MLP_f) input gradient

        // ---- x = mu + c eps, mu = c (mean1/s1 + mu2/s2) -------------------------------------------------------------------
        float dmean1[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float dmu = group_sum<P>(dxp[d]);
            inc[AC::kSc + d] += dmu * e[d];
            if (a.two_q) {
                dmean1[d] = dmu * K.c[d] * K.i1[d];
                const float dm2 = dmu * K.c[d] * K.i2[d];
                inc[AC::kSmm1 + d] += dmu * mean1[d];
                inc[AC::kSmb + d] += dmu * m2[d];
                inc[AC::kSmm + d] += dmu * mu[d];
                if (valid && p == 3) a.dm2_rows[(tb * DX + d) * N + n] = dm2;
            } else {
                dmean1[d] = dmu;
            }
        }
        SEC(6);   // (lpp) quad sums of d x, product-of-Gaussians gradient, d mu2 row store
        // ---- gather backward: scatter-add into the parents (SVO.py:255-257), one dimension per lane of the quad --------
        if (!first) {
            if (valid) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    if ((d & 3) == p) {
                        if (a.bootstrap) {
                            atomicAdd(&nxtP[d * NPT + anc], dmean1[d] + dfmean[d]);
                        } else {
                            atomicAdd(&nxtP[d * NPT + anc], dmean1[d]);
                            atomicAdd(&nxtF[d * NPT + anc], dfmean[d]);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float s1 = block_sum(one ? dmean1[d] : 0.f, red, wave, lane, nw);
                const float s2 = block_sum(one ? dfmean[d] : 0.f, red, wave, lane, nw);
                if (tid == 0) {   // (fm0 aliasing m0: ONE tensor feeds both, its gradient is the sum -- see psvo_hip.h)
                    const bool same0 = (a.fm0 == a.m0);
                    a.dm0[b * DX + d] = same0 ? s1 + s2 : s1;
                    a.dfm0[b * DX + d] = same0 ? 0.f : s2;
                }
            }
        }
        SEC(7);   // (lpp) LDS scatter-add into the parents
        if (t >= 1) {
            c_anc = n_anc;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                c_x[d] = n_x[d]; c_e[d] = n_e[d]; c_m2[d] = n_m2[d];
                c_mean1[d] = n_mean1[d]; c_fmean[d] = n_fmean[d]; c_dfm[d] = n_dfm[d];
            }
#pragma unroll
            for (int k = 0; k < DY; ++k) c_y[k] = n_y[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) c_sc[k] = n_sc[k];
        }
#pragma unroll
        for (int i = 0; i < AC::kSet; ++i) {
            acc[i] += (first && p == 0) ? inc[i] : 0.f;
            acc[AC::kSet + i] += (!first && p == 0) ? inc[i] : 0.f;
        }
        __syncthreads();
        SEC(8);   // (lpp) register rotation (waits for the prefetch) + barrier
    };
    for (int t = T - 1; t >= 1; --t) step(std::false_type{}, t);
    step(std::true_type{}, 0);

#pragma unroll
    for (int i = 0; i < AC::kN; ++i) {
        const float s = block_sum(valid ? acc[i] : 0.f, red, wave, lane, nw);
        if (tid == 0) a.sacc[(size_t)b * AC::kN + i] = s;
    }
}

// Finalize: fold the (B, NACC) sums into gradients of the scale vectors.
//   two_q: c = 1/(1/s1 + 1/s2);  d c = Sc + Smm / c;  d(1/s1) = c*Smm1 - c^2 dc;  d(1/s2) = c*Smb - c^2 dc
//   else : d s1 = Sc
template <int DX, int DY>
__global__ void filter_bwd_finalize(const float* __restrict__ sacc, int B, int two_q, int bootstrap,
                                    const float* sig_q1, const float* sig_q2, const float* sig0, int same0,
                                    float* dsig_q1, float* dsig_q2, float* dsig_f, float* dsig_g, float* dsig0,
                                    float* dfsig0) {
    using AC = FAcc<DX, DY>;
    const int d = threadIdx.x;
    if (d < DX) {
        float S[2 * AC::kSet];
        for (int i = 0; i < 2 * AC::kSet; ++i) S[i] = 0.f;
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < 2 * AC::kSet; ++i) S[i] += sacc[(size_t)b * AC::kN + i];
        float ds2 = 0.f;
        for (int set = 0; set < 2; ++set) {  // set 0: t = 0 (sig0), set 1: t >= 1 (sig_q1)
            const float* P = S + set * AC::kSet;
            const float s1 = set ? sig_q1[d] : sig0[d];
            float ds1;
            if (two_q) {
                const float i1 = 1.f / s1, i2 = 1.f / sig_q2[d];
                const float c = 1.f / (i1 + i2);
                const float dc = P[AC::kSc + d] + P[AC::kSmm + d] / c;
                const float di1 = c * P[AC::kSmm1 + d] - c * c * dc;
                const float di2 = c * P[AC::kSmb + d] - c * c * dc;
                ds1 = -i1 * i1 * di1;
                ds2 += -i2 * i2 * di2;
            } else {
                ds1 = P[AC::kSc + d];
            }
            const float dfs = P[AC::kSfs + d];
            if (set) {
                dsig_q1[d] = ds1 + (bootstrap ? dfs : 0.f);
                dsig_f[d] = bootstrap ? 0.f : dfs;
            } else {
                dsig0[d] = same0 ? ds1 + dfs : ds1;      // (fsig0 aliasing sig0: the sum, as for m0 / fm0)
                dfsig0[d] = same0 ? 0.f : dfs;
            }
        }
        dsig_q2[d] = ds2;
    }
    if (d < DY) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sacc[(size_t)b * AC::kN + AC::kSg + d];
        dsig_g[d] = s;
    }
}

// out[r] = sum_l in[r * L + l]: one wave per row
__global__ void __launch_bounds__(256) row_sum_kernel(const float* __restrict__ in, long long rows, int L,
                                                      float* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int l = lane; l < L; l += 64) s += in[r * L + l];
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

#if PSVO_L == 1
// ---------------------------------------------------------------------------------------------
// The reverse filter as an AFFINE SCAN (round 3; bootstrap wiring with resampling, one hidden layer).
//
// With resampling the gradient w.r.t. logW_t[n] does not depend on later steps, and the only thing the reverse pass carries
// from step t+1 to step t is D_t[n] = d loss / d MLP_q1(x_t[n]) (Dx values per particle): the children of particle n scatter
//     kappa (dx0_c + J_c^T D_{t+1}[c]) + tf_c        into D_t[n]   (c: particles of step t+1 with ancestor n)
// where J_c = d MLP_q1 / d x at x_{t+1}[c], dx0_c and tf_c the parts of the step that do not depend on D, kappa = c / s1 the
// product-of-Gaussians factor.  The recurrence is AFFINE in D:  D_t[n] = ext_t[n] + sum_c (A_c D_{t+1}[c] + b_c)  with a
// Dx x Dx matrix A_c = diag(kappa) J_c^T and a vector b_c per particle and step.  So the time loop splits into
//   A. fbs_coeff_kernel : A and b of EVERY (t, b, n) -- the MLP_g forward + input gradient, the densities and Dx input-gradient
//      passes of MLP_q1 on unit vectors -- one workgroup per (t, sequence): T B workgroups on all CUs, no dependence between them;
//   B. fbs_scan_kernel  : ONE WAVE per sequence walks t = T-1 .. 0 doing nothing but D <- ext + scatter(A D + b): a Dx x Dx
//      matrix-vector product and Dx LDS adds per particle and step, no MLP, no barrier (a single wave: LDS operations are in
//      order), coefficients requested a step ahead; it writes the rows D_t[n] (= the dP rows of psvo_mlp_wgrad);
//   C. fbs_rows_kernel  : everything else the persistent kernel produces, from D -- dG rows, d mu2 rows, the scale sums, d m0 --
//      again one workgroup per (t, sequence).
// The persistent kernel spends 7 800 cycles per step on ONE workgroup per sequence (32 of 256 CUs at C*, 8 at C5) and only a
// quarter of that is MLP arithmetic; here the serial part is ~10 instructions per particle and step.
// Sums are fixed-order (per-step partials folded over t by fbs_fold_kernel; LDS adds of one wave are applied in lane order).
// ---------------------------------------------------------------------------------------------
// One record per (t, sequence, particle), stored as float4 planes [(t, sequence), float4 index, particle] so that the scan reads it
// with 16-byte loads that are contiguous over the lanes:
//   [ A (Dx x Dx, row-major) | b (Dx) | ext (Dx) = the upstream gradient w.r.t. Fm_t[n] | ancestor (int bits) ], padded to float4s.
// (The first layout was (T,B,coefficient,N) read with one dword load per value: 25 loads per particle and step at Dx = 4 --
//  a wave can have 64 vector-memory operations outstanding, so the scan ran ONE step ahead whatever its ring depth.)
template <int DX>
struct ScanRec {
    static constexpr int NC = DX * DX + DX;
    static constexpr int kExt = NC, kAnc = NC + DX;
    static constexpr int REC = (NC + DX + 1 + 3) & ~3;
};

template <int DX, int DY, int H>
__global__ void __launch_bounds__(512) fbs_coeff_kernel(const FilterBwdArgs a) {
    using MQ = MlpLds<DX, H, DX, 1>;
    using MG = MlpLds<DX, H, DY, 1>;
    using SR = ScanRec<DX>;
    constexpr bool kRolled = true;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int t = blockIdx.x, b = blockIdx.y, B = a.B, N = a.N;
    const bool valid = tid < N;
    const int n = valid ? tid : N - 1;
    const size_t tb = (size_t)t * B + b;
    float rec[SR::REC];
#pragma unroll
    for (int i = 0; i < SR::REC; ++i) rec[i] = 0.f;
#pragma unroll
    for (int d = 0; d < DX; ++d) rec[SR::kExt + d] = a.dFm_ext ? a.dFm_ext[(tb * DX + d) * N + n] : 0.f;
    if (t >= 1) {       // (t = 0 has no parent: only the upstream gradient)
        float* wq1 = smem;
        float* wg = wq1 + MQ::kSize;
        MQ::load(wq1, a.q1, tid, NT);
        MG::load(wg, a.g, tid, NT);
        float x[DX], fmean[DX], y[DY];
        const int anc = a.idx[(tb - B) * N + n];
        rec[SR::kAnc] = __int_as_float(anc);
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            x[d] = a.X[(tb * DX + d) * N + n];
            fmean[d] = a.Fm[((tb - B) * DX + d) * N + anc];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = a.obs[tb * DY + k];
        const float sm = exp2_fast((a.logW[tb * N + n] - a.lse[tb]) * kLog2e);
        const float dlw = (a.dlse ? a.dlse[tb] : 0.f) * sm + (a.dlogW_ext ? a.dlogW_ext[tb * N + n] : 0.f);
        float kap[DX], ifs[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float i1 = 1.f / a.sig_q1[d];
            kap[d] = a.two_q ? i1 / (i1 + 1.f / a.sig_q2[d]) : 1.f;     // c / s1
            ifs[d] = i1;                                                // bootstrap: the transition scale is sigma_q1
        }
        __syncthreads();
        float dxg[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dxg[d] = 0.f;
        {
            float gm[DY], dgm[DY];
            MG::template eval<kRolled>(wg, x, gm);
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                float dmean = 1.f;
                if (a.emission) { dmean = emis_dmean(gm[k]); gm[k] = emis_mean(gm[k]); }
                const float isg = 1.f / a.sig_g[k];
                dgm[k] = dlw * (y[k] - gm[k]) * isg * isg * dmean;
            }
            MG::template bwd_input<kRolled>(wg, x, dgm, dxg);
        }
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float tf = dlw * (x[d] - fmean[d]) * ifs[d] * ifs[d];
            rec[DX * DX + d] = fmaf(kap[d], dxg[d] - tf, tf);
        }
#pragma unroll
        for (int k = 0; k < DX; ++k) {
            float ek[DX], col[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) { ek[d] = (d == k) ? 1.f : 0.f; col[d] = 0.f; }
            MQ::template bwd_input<kRolled>(wq1, x, ek, col);          // J^T e_k
#pragma unroll
            for (int d = 0; d < DX; ++d) rec[d * DX + k] = kap[d] * col[d];
        }
    }
    if (valid) {
        // float4 i of the record at [(tb R4 + i) N + n]: a wave's accesses are 16 bytes per lane AND contiguous over the lanes
        // (particle-major records -- 112 bytes apart at Dx = 4 -- cost 64 cache lines per access: C5 67.0 -> 69.1 ms)
        float4* dst = reinterpret_cast<float4*>(a.scanAB) + (tb * (SR::REC / 4)) * N + n;
#pragma unroll
        for (int i = 0; i < SR::REC / 4; ++i) dst[(size_t)i * N] = make_float4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
    }
}

// NWV waves per sequence (1: no barrier at all -- LDS operations of one wave are in order; 4: one barrier per step), PPL particles
// per lane (N <= 64 NWV PPL), DEPTH steps of coefficients in flight: a step takes ~0.2 us and HBM ~2 us, so the loads of step
// t - DEPTH are issued when step t is consumed (a ring of DEPTH register sets, the step loop unrolled DEPTH times so that every
// index into it is static).
template <int DX, int NWV, int PPL, int DEPTH, int NCP = NWV>
__global__ void __launch_bounds__(64 * NWV) fbs_scan_kernel(const FilterBwdArgs a) {
    using SR = ScanRec<DX>;
    constexpr int NTS = 64 * NWV, R4 = SR::REC / 4;
    // NCP = NWV: one scatter copy per wave (fixed summation order; the launcher's choice at every N).  NCP = 1: one shared copy
    // and LDS float atomics across the waves -- measured at C5 (N = 512, Dx = 4): 68.4 against 68.3 ms, the copies are not what
    // the step waits for
    // scatter targets: one copy per wave, summed by the parent in wave order (LDS adds of ONE wave are applied in lane order,
    // adds of several waves in arrival order: with the copies every sum has a fixed order, as in filter_bwd_lpp_kernel)
    extern __shared__ __attribute__((aligned(16))) float acc[];      // [2][NWV][DX][N]
    const int tid = threadIdx.x, b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const int CPY = DX * N, mine = (NCP > 1 ? (tid >> 6) : 0) * CPY;
    for (int i = tid; i < 2 * NCP * CPY; i += NTS) acc[i] = 0.f;
    float4 rec[DEPTH][PPL][R4];
    auto load = [&](int t, float4 (&r)[PPL][R4]) {
        const size_t tb = (size_t)t * B + b;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            const int n = min(tid + NTS * p, N - 1);
            const float4* src = reinterpret_cast<const float4*>(a.scanAB) + (tb * R4) * N + n;
#pragma unroll
            for (int i = 0; i < R4; ++i) r[p][i] = src[(size_t)i * N];
        }
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s)
        if (T - 1 - s >= 0) load(T - 1 - s, rec[s]);
    if (NWV > 1) __syncthreads();
    for (int t0 = T - 1; t0 >= 0; t0 -= DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int t = t0 - s;
            if (t < 0) break;
            const size_t tb = (size_t)t * B + b;
            float* cur = acc + (t & 1) * NCP * CPY;
            float* nxt = acc + ((t + 1) & 1) * NCP * CPY + mine;
            float D[PPL][DX];
            float cf[PPL][SR::REC];
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
#pragma unroll
                for (int i = 0; i < R4; ++i) {
                    cf[p][4 * i] = rec[s][p][i].x; cf[p][4 * i + 1] = rec[s][p][i].y;
                    cf[p][4 * i + 2] = rec[s][p][i].z; cf[p][4 * i + 3] = rec[s][p][i].w;
                }
            }
            if (t - DEPTH >= 0) load(t - DEPTH, rec[s]);      // refill the slot just consumed
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const int n = tid + NTS * p;
                if (n < N) {
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < NCP; ++w) {
                            v += cur[w * CPY + d * N + n];
                            cur[w * CPY + d * N + n] = 0.f;
                        }
                        D[p][d] = v + cf[p][SR::kExt + d];
                        a.dP[(tb * DX + d) * N + n] = D[p][d];
                    }
                }
            }
            if (t >= 1) {
#pragma unroll
                for (int p = 0; p < PPL; ++p) {
                    const int n = tid + NTS * p;
                    if (n < N) {
                        const int an = __float_as_int(cf[p][SR::kAnc]);
#pragma unroll
                        for (int d = 0; d < DX; ++d) {
                            float v = cf[p][DX * DX + d];
#pragma unroll
                            for (int k = 0; k < DX; ++k) v = fmaf(cf[p][d * DX + k], D[p][k], v);
                            atomicAdd(&nxt[d * N + an], v);
                        }
                    }
                }
            }
            if (NWV > 1) {
                // a workgroup barrier that orders the LDS traffic of the step ONLY: __syncthreads() also waits for every
                // outstanding global access (s_waitcnt vmcnt(0)), i.e. for the record loads just issued for DEPTH steps ahead
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}

template <int DX, int DY, int H>
__global__ void __launch_bounds__(512) fbs_rows_kernel(const FilterBwdArgs a) {
    using MQ = MlpLds<DX, H, DX, 1>;
    using MG = MlpLds<DX, H, DY, 1>;
    using AC = FAcc<DX, DY>;
    constexpr bool kRolled = true;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x, nw = NT >> 6;
    const int t = blockIdx.x, b = blockIdx.y, B = a.B, N = a.N;
    const bool valid = tid < N, first = (t == 0);
    const int n = valid ? tid : N - 1;
    float* wq1 = smem;
    float* wg = wq1 + MQ::kSize;
    float* red = wg + MG::kSize;     // [nw][kN] + 16
    MQ::load(wq1, a.q1, tid, NT);
    MG::load(wg, a.g, tid, NT);
    const size_t tb = (size_t)t * B + b;
    float s1[DX], s2[DX], fs[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        s1[d] = first ? a.sig0[d] : a.sig_q1[d];
        s2[d] = a.two_q ? a.sig_q2[d] : 1.f;
        fs[d] = first ? a.fsig0[d] : a.sig_q1[d];
    }
    const BStepK<DX> K = make_bstepk<DX>(s1, s2, fs, a.two_q != 0);
    float x[DX], e[DX], m2[DX], y[DY], mean1[DX], fmean[DX], dPn[DX];
    const int anc = first ? n : a.idx[(tb - B) * N + n];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        x[d] = a.X[(tb * DX + d) * N + n];
        e[d] = a.eps[(tb * DX + d) * N + n];
        m2[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
        fmean[d] = first ? a.fm0[b * DX + d] : a.Fm[((tb - B) * DX + d) * N + anc];
        mean1[d] = first ? a.m0[b * DX + d] : fmean[d];
        dPn[d] = valid ? a.dP[(tb * DX + d) * N + n] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < DY; ++k) y[k] = a.obs[tb * DY + k];
    const float sm = valid ? exp2_fast((a.logW[tb * N + n] - a.lse[tb]) * kLog2e) : 0.f;
    float dlw = (a.dlse ? a.dlse[tb] : 0.f) * sm + (a.dlogW_ext ? a.dlogW_ext[tb * N + n] : 0.f);
    if (!valid) dlw = 0.f;
    float mu[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) mu[d] = a.two_q ? K.c[d] * fmaf(K.i1[d], mean1[d], K.i2[d] * m2[d]) : mean1[d];
    __syncthreads();

    float acc[AC::kN];
#pragma unroll
    for (int i = 0; i < AC::kN; ++i) acc[i] = 0.f;
    float inc[AC::kSet];
#pragma unroll
    for (int i = 0; i < AC::kSet; ++i) inc[i] = 0.f;
    float dx[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) dx[d] = 0.f;
    {
        float gm[DY], dgm[DY];
        MG::template eval<kRolled>(wg, x, gm);
#pragma unroll
        for (int k = 0; k < DY; ++k) {
            float dmean = 1.f;
            if (a.emission) { dmean = emis_dmean(gm[k]); gm[k] = emis_mean(gm[k]); }
            const float isg = 1.f / a.sig_g[k];
            const float z = (y[k] - gm[k]) * isg;
            dgm[k] = dlw * z * isg * dmean;
            acc[AC::kSg + k] += dlw * (z * z - 1.f) * isg;
            if (valid) a.dG[(tb * DY + k) * N + n] = dgm[k];
        }
        MG::template bwd_input<kRolled>(wg, x, dgm, dx);
    }
    float dfmean[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float z = (x[d] - fmean[d]) * K.ifs[d];
        const float tf = dlw * z * K.ifs[d];
        dx[d] -= tf;
        dfmean[d] = tf;
        inc[AC::kSfs + d] += dlw * (z * z - 1.f) * K.ifs[d];
        inc[AC::kSc + d] += dlw * K.ic[d];
    }
    MQ::template bwd_input<kRolled>(wq1, x, dPn, dx);
    float dmean1[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float dmu = dx[d];
        inc[AC::kSc + d] += dmu * e[d];
        if (a.two_q) {
            dmean1[d] = dmu * K.c[d] * K.i1[d];
            inc[AC::kSmm1 + d] += dmu * mean1[d];
            inc[AC::kSmb + d] += dmu * m2[d];
            inc[AC::kSmm + d] += dmu * mu[d];
            if (valid) a.dm2_rows[(tb * DX + d) * N + n] = dmu * K.c[d] * K.i2[d];
        } else {
            dmean1[d] = dmu;
        }
    }
#pragma unroll
    for (int i = 0; i < AC::kSet; ++i) acc[(first ? 0 : AC::kSet) + i] = inc[i];
    // ---- this step's partial of the per-sequence sums (fixed order: lanes, then waves) ----------------------------------
#pragma unroll
    for (int i = 0; i < AC::kN; ++i) {
        const float v = wave_sum(acc[i]);
        if (lane == 0) red[wave * AC::kN + i] = v;
    }
    if (first) {
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float v1 = wave_sum(dmean1[d]), v2 = wave_sum(dfmean[d]);
            if (lane == 0) {
                red[nw * AC::kN + wave * 2 * DX + d] = v1;
                red[nw * AC::kN + wave * 2 * DX + DX + d] = v2;
            }
        }
    }
    __syncthreads();
    if (tid < AC::kN) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += red[w * AC::kN + tid];
        a.scanPart[tb * AC::kN + tid] = v;
    }
    if (first && tid < DX) {
        float v1 = 0.f, v2 = 0.f;
        for (int w = 0; w < nw; ++w) {
            v1 += red[nw * AC::kN + w * 2 * DX + tid];
            v2 += red[nw * AC::kN + w * 2 * DX + DX + tid];
        }
        const bool same0 = (a.fm0 == a.m0);
        a.dm0[b * DX + tid] = same0 ? v1 + v2 : v1;
        a.dfm0[b * DX + tid] = same0 ? 0.f : v2;
    }
}

// sacc[b][i] = sum_t part[t][b][i]: one wave per (b, i) strides over t and sums its lanes in a fixed order
__global__ void __launch_bounds__(256) fbs_fold_kernel(const float* __restrict__ part, int T, int B, int NACC,
                                                       float* __restrict__ sacc) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < NACC; i += 4) {
        float v = 0.f;
        for (int t = lane; t < T; t += 64) v += part[((size_t)t * B + b) * NACC + i];
        v = wave_sum(v);
        if (lane == 0) sacc[(size_t)b * NACC + i] = v;
    }
}

template <int DX, int DY, int H>
static void launch_filter_bwd_scan(const FilterBwdArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, 1>;
    using MG = MlpLds<DX, H, DY, 1>;
    using AC = FAcc<DX, DY>;
    const int NT = (a.N + 63) & ~63;
    const size_t ldsw = sizeof(float) * (MQ::kSize + MG::kSize);
    hipLaunchKernelGGL((fbs_coeff_kernel<DX, DY, H>), dim3(a.T, a.B), dim3(NT), ldsw, stream, a);
    const size_t ldss = sizeof(float) * 2 * DX * a.N * (a.N <= 128 ? 1 : 4);
    // registers of the record ring: DEPTH x PPL x REC (12 / 16 / 28 floats) -- <= ~250 of the 512 a lone wave per SIMD may use
    constexpr int D1 = (DX == 2) ? 16 : (DX == 3) ? 12 : 8, D2 = D1 / 2;
    if (a.N <= 64) hipLaunchKernelGGL((fbs_scan_kernel<DX, 1, 1, D1>), dim3(a.B), dim3(64), ldss, stream, a);
    else if (a.N <= 128) hipLaunchKernelGGL((fbs_scan_kernel<DX, 1, 2, D2>), dim3(a.B), dim3(64), ldss, stream, a);
    else if (a.N <= 256) hipLaunchKernelGGL((fbs_scan_kernel<DX, 4, 1, D1>), dim3(a.B), dim3(256), ldss, stream, a);
    else hipLaunchKernelGGL((fbs_scan_kernel<DX, 4, 2, D2>), dim3(a.B), dim3(256), ldss, stream, a);
    const size_t ldsr = ldsw + sizeof(float) * ((NT / 64) * (AC::kN + 2 * DX) + 16);
    hipLaunchKernelGGL((fbs_rows_kernel<DX, DY, H>), dim3(a.T, a.B), dim3(NT), ldsr, stream, a);
    hipLaunchKernelGGL(fbs_fold_kernel, dim3(a.B), dim3(256), 0, stream, a.scanPart, a.T, a.B, AC::kN, a.sacc);
}
#endif   // PSVO_L == 1

struct FilterBwdOut {
    float *dsig_q1, *dsig_q2, *dsig_f, *dsig_g, *dsig0, *dfsig0;
};

template <int DX, int DY, int H>
static int launch_filter_bwd(const FilterBwdArgs& a, const FilterBwdOut& o, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    const int NT = (a.N + 63) & ~63;
    FilterBwdArgs aw = a;
    const size_t arrays = a.bootstrap ? 2 : 4;       // d P1 (+ d Fm) double-buffered; the d Fm pair only without bootstrap
    size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + arrays * (size_t)(NT / 64) * DX * NT + 16);
    // (N <= 256 only: at N = 512 the eight copies cost 3 % of the C5 step -- 32 reads and clears per particle and step at
    //  Dx = 4 -- and C5 is the configuration that can least afford it; there the scatter keeps its float atomics)
    aw.wave_copies = a.N <= 256 && lds <= 150 * 1024;
    if (!aw.wave_copies) lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + arrays * DX * NT + 16);
    clear_hip_error();
    constexpr bool kLppOk = (H % 16 == 0) && (PSVO_L == 2 || (H <= 32 && DX <= 3));   // (two layers: as in filter_fwd.hip)
    bool lpp = false;
#if PSVO_L == 1
    // Used where the reverse filter IS the tail of the step: the filter-only objectives (no upstream gradient from a backward
    // simulation: C2 0.90 -> 0.67 ms) and N > 256 (one workgroup per sequence on 8 CUs at C5: 71.2 -> 66.7 ms).  Behind a
    // backward simulation at N <= 256 the tail is bound by the weight gradients that run beside it -- C* 3.45 -> 3.42, C4
    // unchanged -- and the persistent kernel stays (psvo_set_tuning(PSVO_TUNE_FILTER_BWD, 2) forces the scan everywhere).
    const bool scan_pays = (!a.dFm_ext && !a.dlogW_ext) || a.N > 256 || g_tune_filter_bwd_scan == 2;
    if (g_tune_filter_bwd_scan && scan_pays && a.bootstrap && a.resample && a.nparts <= 1 && a.scanAB && a.scanPart) {
        launch_filter_bwd_scan<DX, DY, H>(a, stream);
        lpp = true;      // (the persistent kernels are skipped)
    } else
#endif
    if constexpr (kLppOk) {
        if (a.N <= 128) {   // latency-bound regime: four lanes per particle
            const int NT4 = (4 * a.N + 63) & ~63;
            const size_t lds4 = sizeof(float) * (2 * MQ::kSize + MG::kSize + 4 * (NT4 / 64) * DX * (NT4 / 4) + 16);
            hipLaunchKernelGGL((filter_bwd_lpp_kernel<DX, DY, H>), dim3(a.B), dim3(NT4), lds4, stream, a);
            lpp = true;
        }
    }
    if (lpp) {
    } else if (NT <= 256)
        hipLaunchKernelGGL((filter_bwd_kernel<DX, DY, H, 256>), dim3(a.B), dim3(NT), lds, stream, aw);
    else
        hipLaunchKernelGGL((filter_bwd_kernel<DX, DY, H, 512>), dim3(a.B), dim3(NT), lds, stream, aw);
    if (a.two_q) {
        const long long rows = (long long)a.T * a.B * DX;
        hipLaunchKernelGGL(row_sum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a.dm2_rows, rows,
                           a.N, a.dmu2);
    }
    hipLaunchKernelGGL((filter_bwd_finalize<DX, DY>), dim3(1), dim3(64), 0, stream, a.sacc, a.B, a.two_q, a.bootstrap,
                       a.sig_q1, a.sig_q2, a.sig0, (int)(a.fsig0 == a.sig0), o.dsig_q1, o.dsig_q2, o.dsig_f, o.dsig_g, o.dsig0, o.dfsig0);
    return launch_status();
}

template <int DX, int DY>
static int fb_dispatch_h(const FilterBwdArgs& a, const FilterBwdOut& o, int H, hipStream_t s) {
    switch (H) {
#if PSVO_L == 1   // (two hidden layers: widths 32 and 64; narrower ones are zero-padded upstream)
        case 16: return launch_filter_bwd<DX, DY, 16>(a, o, s);
#endif
        case 32: return launch_filter_bwd<DX, DY, 32>(a, o, s);
        case 64: return launch_filter_bwd<DX, DY, 64>(a, o, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX>
static int fb_dispatch_dy(const FilterBwdArgs& a, const FilterBwdOut& o, int Dy, int H, hipStream_t s) {
    switch (Dy) {
        case 1: return fb_dispatch_h<DX, 1>(a, o, H, s);
        case 2: return fb_dispatch_h<DX, 2>(a, o, H, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // inline namespace PSVO_LNS
}  // namespace psvo

#if PSVO_L == 1   // (sizing helpers: independent of the number of hidden layers)
extern "C" int psvo_filter_acc_size(int Dx, int Dy) { return 10 * Dx + Dy; }

extern "C" long long psvo_filter_ws_floats(int B, int T, int N, int Dx, int Dy) {
    // per-sequence sums | d mu2 rows | affine-scan records (T,B,N,REC) | per-step partial sums (T,B,NACC)
    const long long rec = (Dx * Dx + 2 * Dx + 1 + 3) / 4 * 4;      // ScanRec<Dx>::REC
    return (long long)B * (10 * Dx + Dy) + (long long)T * B * Dx * N + 4 + (long long)T * B * rec * N + (long long)T * B * (10 * Dx + Dy);
}
#endif

PSVO_L2_DECL(psvo_filter_backward)
PSVO_ENTRY(psvo_filter_backward)(
    const psvo_desc* desc, const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g, const float* sig_q1,
    const float* sig_q2, const float* sig_f, const float* sig_g, const float* mu2, const float* m0, const float* sig0,
    const float* fm0, const float* fsig0, const float* obs, const float* eps, const float* X, const float* Fm,
    const float* P1, const float* logW, const float* lse, const int32_t* idx, const float* dlse, int nparts,
    const float* dFm_ext, const float* dlogW_ext, float* dP, float* dF, float* dG, float* dmu2, float* dm0,
    float* dfm0, float* dsig_q1, float* dsig_q2, float* dsig_f, float* dsig_g, float* dsig0, float* dfsig0,
    float* sacc, void* stream) {
    using namespace psvo;
    if (!desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;
#if PSVO_L == 1
    if (desc && desc->layers == 2)
        return psvo_filter_backward_l2(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0,
            obs, eps, X, Fm, P1, logW, lse, idx, dlse, nparts, dFm_ext, dlogW_ext, dP, dF, dG, dmu2, dm0, dfm0,
            dsig_q1, dsig_q2, dsig_f, dsig_g, dsig0, dfsig0, sacc, stream);
#endif
    if (!mlp_layers_ok(q1) || !mlp_layers_ok(f) || !mlp_layers_ok(g)) return PSVO_ERR_INVALID;
    if (!desc || !q1 || !g || !sig_q1 || !sig_g || !m0 || !sig0 || !fm0 || !fsig0 || !obs || !eps || !X || !Fm ||
        !logW || !lse || !dP || !dG || !dm0 || !dfm0 || !dsig_q1 || !dsig_q2 || !dsig_f || !dsig_g || !dsig0 ||
        !dfsig0 || !sacc)
        return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T <= 0 || desc->N <= 0) return PSVO_ERR_INVALID;
    if (desc->two_q && (!mu2 || !sig_q2 || !dmu2)) return PSVO_ERR_INVALID;
    if (!desc->bootstrap && (!f || !sig_f || !P1 || !dF)) return PSVO_ERR_INVALID;
    if (desc->resample && !idx) return PSVO_ERR_INVALID;
    if ((dFm_ext || dlogW_ext) && nparts <= 0) return PSVO_ERR_INVALID;
    if (desc->N > 512) return PSVO_ERR_UNSUPPORTED;

    FilterBwdArgs a;
    a.wave_copies = 0;
    a.B = desc->B; a.T = desc->T; a.N = desc->N;
    a.resample = desc->resample; a.two_q = desc->two_q; a.bootstrap = desc->bootstrap; a.emission = desc->emission;
    a.q1 = *q1; a.f = desc->bootstrap ? *q1 : *f; a.g = *g;
    a.sig_q1 = sig_q1; a.sig_q2 = sig_q2; a.sig_f = sig_f; a.sig_g = sig_g;
    a.mu2 = mu2; a.m0 = m0; a.sig0 = sig0; a.fm0 = fm0; a.fsig0 = fsig0; a.obs = obs; a.eps = eps;
    a.X = X; a.Fm = Fm; a.P1 = P1; a.logW = logW; a.lse = lse; a.idx = idx;
    a.dlse = dlse; a.nparts = nparts; a.dFm_ext = dFm_ext; a.dlogW_ext = dlogW_ext;
    a.dP = dP; a.dF = dF; a.dG = dG; a.dmu2 = dmu2; a.dm0 = dm0; a.dfm0 = dfm0; a.sacc = sacc;
    a.dm2_rows = sacc + (size_t)desc->B * psvo_filter_acc_size(desc->Dx, desc->Dy);
    {   // the records are read with 16-byte accesses: their offset inside the (16-byte aligned) workspace is rounded up
        const size_t off = (size_t)desc->B * psvo_filter_acc_size(desc->Dx, desc->Dy) + (size_t)desc->T * desc->B * desc->Dx * desc->N;
        a.scanAB = sacc + ((off + 3) & ~(size_t)3);
    }
    if (reinterpret_cast<uintptr_t>(a.scanAB) & 15) a.scanAB = nullptr;      // (a misaligned workspace: the persistent kernel)
    a.scanPart = a.scanAB + (size_t)desc->T * desc->B * ((desc->Dx * desc->Dx + 2 * desc->Dx + 1 + 3) / 4 * 4) * desc->N;
    FilterBwdOut o{dsig_q1, dsig_q2, dsig_f, dsig_g, dsig0, dfsig0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return fb_dispatch_dy<2>(a, o, desc->Dy, desc->H, s);
        case 3: return fb_dispatch_dy<3>(a, o, desc->Dy, desc->H, s);
        case 4: return fb_dispatch_dy<4>(a, o, desc->Dy, desc->H, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
