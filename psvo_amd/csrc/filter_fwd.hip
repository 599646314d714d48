// Forward particle filter: one persistent workgroup per sequence, time loop in-kernel.
// Restates SVO.SMC (reference src/SMC/SVO.py:60-180) -- see include/psvo_hip.h for the contract.
//
// Work decomposition (MI355X): lane = particle n, workgroup = sequence b.  Everything inside a
// step is per-lane except the log-sum-exp over particles and the multinomial draw, which are a
// wavefront shuffle reduction/scan plus one LDS hop across the <= 16 waves of the workgroup.
// MLP weights live in LDS and are read as wave-uniform float4 broadcasts.  In bootstrap mode the
// proposal mean of a resampled particle is MLP_f(X_t[a]) = Fm_t[a], so the transition MLP is
// evaluated once per pre-resampling particle and *gathered* together with the particle instead
// of being re-evaluated after the gather (halves the dependent MLP chain per step).
#include "common.h"

namespace psvo {
inline namespace PSVO_LNS {   // l1 / l2: hidden layers of the per-particle MLPs (common.h)
PSVO_TIMERS_DEFINE(filter_fwd)


struct FilterArgs {
    int B, T, N;
    int resample, two_q, bootstrap, emission;
    psvo_mlp q1, f, g;
    const float *sig_q1, *sig_q2, *sig_f, *sig_g;
    const float *mu2, *m0, *sig0, *fm0, *fsig0, *obs, *eps, *u;
    const int32_t* idx_in;
    float *X, *Xanc, *Fm, *P1, *logW;
    int32_t* idx_out;
    float* lse;
};

template <int DX>
struct StepK {
    float c[DX];    // proposal scale
    float ic[DX];   // 1 / c
    float i1[DX];   // 1 / s1
    float i2[DX];   // 1 / s2 (two_q)
    float ifs[DX];  // 1 / transition scale
    float lq, lf;   // -sum(log scale) - D/2 log(2 pi) of proposal / transition
};

template <int DX>
__device__ __forceinline__ StepK<DX> make_stepk(const float* s1, const float* s2, const float* fs, bool two_q) {
    StepK<DX> K;
    float lq = -DX * kHalfLog2Pi, lf = -DX * kHalfLog2Pi;
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
        lq -= logf(K.c[d]);
        lf -= logf(fs[d]);
    }
    K.lq = lq;
    K.lf = lf;
    return K;
}

// EM: emission variant fixed at compile time (0 / 1) or read from the arguments (2).  The one-wave-per-SIMD build
// (MAXT = 256) keeps every MLP weight in VGPRs only without the extra branch in its time loop, so it is compiled both ways.
template <int DX, int DY, int H, int MAXT, int EM = 2>
__global__ void __launch_bounds__(MAXT) filter_fwd_kernel(const FilterArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, nw = NT >> 6;
    const int b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const bool valid = tid < N;
    const int n = valid ? tid : N - 1;  // clamp so that loads stay in bounds

    float* wq1 = smem;
    float* wf = wq1 + MQ::kSize;
    float* wg = wf + MQ::kSize;
    float* cdf = wg + MG::kSize;
    float* sx = cdf + NT;          // [DX][NT] staged X_t
    float* sp = sx + DX * NT;      // [DX][NT] staged MLP_q1(X_t)
    float* sf = sp + DX * NT;      // [DX][NT] staged MLP_f(X_t) (== sp when bootstrap)
    float* red = sf + DX * NT;     // 48 floats scratch

    MQ::load(wq1, a.q1, tid, NT);
    if (!a.bootstrap) MQ::load(wf, a.f, tid, NT);
    MG::load(wg, a.g, tid, NT);
    const float* wfm = a.bootstrap ? wq1 : wf;

    // step constants: t = 0 uses (q0, sigma_q0) for the first proposal term
    float sq1[DX], sq2[DX], sfv[DX], s0[DX], fs0[DX], isg[DY];
    float lg = -DY * kHalfLog2Pi;
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        sq1[d] = a.sig_q1[d];
        sq2[d] = a.two_q ? a.sig_q2[d] : 1.f;
        sfv[d] = a.bootstrap ? a.sig_q1[d] : a.sig_f[d];
        s0[d] = a.sig0[d];
        fs0[d] = a.fsig0[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) {
        const float s = a.sig_g[e];
        isg[e] = 1.f / s;
        lg -= logf(s);
    }
    const StepK<DX> K0 = make_stepk<DX>(s0, sq2, fs0, a.two_q != 0);
    const StepK<DX> K1 = make_stepk<DX>(sq1, sq2, sfv, a.two_q != 0);
    const float neg_logN = -logf((float)N);
    const float ninf = -__builtin_huge_valf();

    float mean1[DX], fmean[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        mean1[d] = a.m0[b * DX + d];
        fmean[d] = a.fm0[b * DX + d];
    }
    float lnw = neg_logN;

    // software prefetch of the next step's inputs
    float eps_c[DX], mu2_c[DX], obs_c[DY], u_c = 0.f;
    int idx_c = 0;
    auto load_inputs = [&](int t, float (&e)[DX], float (&m)[DX], float (&o)[DY], float& uu, int& ii) {
        const size_t tb = (size_t)t * B + b;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            e[d] = a.eps[(tb * DX + d) * N + n];
            m[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) o[k] = a.obs[tb * DY + k];
        if (a.resample) {
            if (a.idx_in) ii = a.idx_in[tb * N + n];
            else uu = a.u[tb * N + n];
        }
    };
    load_inputs(0, eps_c, mu2_c, obs_c, u_c, idx_c);
    __syncthreads();  // weights visible

    // Weights are loop-invariant LDS reads: when they fit the register budget the compiler keeps
    // them all resident in VGPRs (no LDS traffic in the MLPs); otherwise make the LDS offset
    // opaque per step so they are re-read as broadcasts instead of being spilled to scratch.
    constexpr int kWeights = MQ::kSize + MG::kSize;
    constexpr bool kOpaque = (MAXT > 256) ? (kWeights > 120) : (kWeights > 330);

    SEC_INIT(filter_fwd)
    for (int t = 0; t < T; ++t) {
        SEC(0);
        const size_t tb = (size_t)t * B + b;
        const StepK<DX> K = (t == 0) ? K0 : K1;
        int zo = 0;
        if (kOpaque) asm volatile("" : "+v"(zo));
        const float* wq1_t = wq1 + zo;
        const float* wfm_t = wfm + zo;
        const float* wg_t = wg + zo;

        float eps_n[DX], mu2_n[DX], obs_n[DY], u_n = 0.f;
        int idx_n = 0;
        if (t + 1 < T) load_inputs(t + 1, eps_n, mu2_n, obs_n, u_n, idx_n);

        SEC(1);   // issue of the prefetch loads
        // ---- proposal: product of two diagonal Gaussians on *scales* (SVO.py:186-197) ----------
        float mu[DX], x[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            mu[d] = a.two_q ? K.c[d] * fmaf(K.i1[d], mean1[d], K.i2[d] * mu2_c[d]) : mean1[d];
            x[d] = fmaf(K.c[d], eps_c[d], mu[d]);
        }
        const float q_lp = diag_lp<DX>(x, mu, K.ic, K.lq);
        const float f_lp = diag_lp<DX>(x, fmean, K.ifs, K.lf);

        SEC(2);   // proposal, q / f densities
        // ---- emission ------------------------------------------------------------------------
        float gm[DY];
        MG::template eval<kOpaque>(wg_t, x, gm);
        if (EM == 2 ? (a.emission != 0) : (EM == 1)) {
#pragma unroll
            for (int k = 0; k < DY; ++k) gm[k] = emis_mean(gm[k]);
        }
        const float g_lp = diag_lp<DY>(obs_c, gm, isg, lg);

        float lw = f_lp + g_lp - q_lp + lnw;
        if (!valid) lw = ninf;

        SEC(3);   // MLP_g, weight
        // ---- next-step proposal / transition means of every pre-resampling particle -----------
        float p1[DX], fm[DX];
        MQ::template eval<kOpaque>(wq1_t, x, p1);
        if (a.bootstrap) {
#pragma unroll
            for (int d = 0; d < DX; ++d) fm[d] = p1[d];
        } else {
            MQ::template eval<kOpaque>(wfm_t, x, fm);
        }
        if (valid) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                a.X[(tb * DX + d) * N + n] = x[d];
                a.Fm[(tb * DX + d) * N + n] = fm[d];
                if (a.P1) a.P1[(tb * DX + d) * N + n] = p1[d];
            }
            a.logW[tb * N + n] = lw;
        }

        SEC(4);   // MLP_q1 (/ MLP_f), history stores
        // ---- log-sum-exp over particles and multinomial ancestors (SVO.py:266-300) -------------
        const float mx = block_max(lw, red, 0, wave, lane, nw);
        const float w = valid ? exp2_fast((lw - mx) * kLog2e) : 0.f;
        float sc = wave_incl_scan(w, lane);
        float total;
        if (nw > 1) {
            float* wt = red + 32;
            if (lane == 63) wt[wave] = sc;
            __syncthreads();
            float off = 0.f, tot = 0.f;
            for (int i = 0; i < nw; ++i) {
                const float v = wt[i];
                if (i < wave) off += v;
                tot += v;
            }
            sc += off;
            total = tot;
        } else {
            total = lane_bcast(sc, 63);
        }
        const float lse_t = fmaf(kLn2, log2_fast(total), mx);
        if (tid == 0) a.lse[tb] = lse_t;

        SEC(5);   // block max, exp, scan, total
        if (a.resample) {
            cdf[tid] = sc;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                sx[d * NT + tid] = x[d];
                sp[d * NT + tid] = p1[d];
                if (!a.bootstrap) sf[d * NT + tid] = fm[d];
            }
            __syncthreads();
            SEC(6);   // stage particles / CDF in LDS + barrier
            int idx;
            if (a.idx_in) {
                idx = idx_c;
            } else {
                // count of cdf entries <= u * total (cdf is non-decreasing)
                const float target = u_c * total;
                int pos = 0;
                for (int s = 1 << (31 - __clz(N)); s > 0; s >>= 1) {
                    const int p = pos + s;
                    if (p <= N && cdf[p - 1] <= target) pos = p;
                }
                idx = min(pos, N - 1);
            }
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                x[d] = sx[d * NT + idx];
                mean1[d] = sp[d * NT + idx];
                fmean[d] = a.bootstrap ? mean1[d] : sf[d * NT + idx];
            }
            if (valid) {
                a.idx_out[tb * N + n] = idx;
#pragma unroll
                for (int d = 0; d < DX; ++d) a.Xanc[(tb * DX + d) * N + n] = x[d];
            }
            lnw = neg_logN;
            SEC(7);   // CDF search, gather, ancestor stores
            __syncthreads();  // staged tiles are rewritten next step
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                mean1[d] = p1[d];
                fmean[d] = fm[d];
            }
            if (valid) {
                a.idx_out[tb * N + n] = n;
#pragma unroll
                for (int d = 0; d < DX; ++d) a.Xanc[(tb * DX + d) * N + n] = x[d];
            }
            lnw = lw - lse_t;
            if (nw > 1) __syncthreads();  // red[] reuse
        }

#pragma unroll
        for (int d = 0; d < DX; ++d) {
            eps_c[d] = eps_n[d];
            mu2_c[d] = mu2_n[d];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) obs_c[k] = obs_n[k];
        u_c = u_n;
        idx_c = idx_n;
    }
}

// ---------------------------------------------------------------------------------------------
// Four lanes per particle (N <= 256).  The filter runs ONE workgroup per sequence, so a step is a pure
// latency chain; with lane = particle the two dependent MLP evaluations were > 40 % of it.  Here particle
// n owns the quad of lanes 4n..4n+3: lane p evaluates hidden units [p*H/4, (p+1)*H/4) of every MLP (its
// slice of the weights is loop-invariant and stays in registers) and the outputs are summed over the quad
// with two DPP adds.  Everything else is computed redundantly in the four lanes; the multinomial search is a
// two-round 16-ary search in which the quad reads 16 CDF pivots / 16 CDF entries as four float4 per round.
// The per-sequence log-sum-exp and CDF use per-wave maxima and ONE barrier.
// ---------------------------------------------------------------------------------------------
template <int DX, int DY, int H, int MAXT>
__global__ void __launch_bounds__(MAXT) filter_fwd_lpp_kernel(const FilterArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    constexpr int P = 4;
    constexpr bool kOpaque = false;   // (every instantiated shape keeps its weight slice in VGPRs without scratch)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, nw = NT >> 6;
    const int NPT = NT / P;                       // particle slots, a multiple of 16
    const int b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const int pn = tid >> 2, p = tid & 3;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);   // wave index as a scalar
    const bool valid = pn < N;
    const int n = valid ? pn : N - 1;

    float* wq1 = smem;
    float* wf = wq1 + MQ::kSize;
    float* wg = wf + MQ::kSize;
    float* cdf = wg + MG::kSize;   // [NPT]
    float* piv = cdf + NPT;        // [16] cdf[16 i + 15], +inf beyond the last block
    float* sx = piv + 16;          // [DX][NPT] staged X_t
    float* sp = sx + DX * NPT;     // [DX][NPT] staged MLP_q1(X_t)
    float* sf = sp + DX * NPT;     // [DX][NPT] staged MLP_f(X_t) (unused when bootstrap)
    float* red = sf + DX * NPT;    // [2][16] per-wave (max, sum)

    MQ::load(wq1, a.q1, tid, NT);
    if (!a.bootstrap) MQ::load(wf, a.f, tid, NT);
    MG::load(wg, a.g, tid, NT);
    const float* wfm = a.bootstrap ? wq1 : wf;
    if (tid < 16) piv[tid] = __builtin_huge_valf();

    float sq1[DX], sq2[DX], sfv[DX], s0[DX], fs0[DX], isg[DY];
    float lg = -DY * kHalfLog2Pi;
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        sq1[d] = a.sig_q1[d];
        sq2[d] = a.two_q ? a.sig_q2[d] : 1.f;
        sfv[d] = a.bootstrap ? a.sig_q1[d] : a.sig_f[d];
        s0[d] = a.sig0[d];
        fs0[d] = a.fsig0[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) {
        const float s = a.sig_g[e];
        isg[e] = 1.f / s;
        lg -= logf(s);
    }
    const StepK<DX> K0 = make_stepk<DX>(s0, sq2, fs0, a.two_q != 0);
    const StepK<DX> K1 = make_stepk<DX>(sq1, sq2, sfv, a.two_q != 0);
    const float neg_logN = -logf((float)N);
    const float ninf = -__builtin_huge_valf();

    float mean1[DX], fmean[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        mean1[d] = a.m0[b * DX + d];
        fmean[d] = a.fm0[b * DX + d];
    }
    float lnw = neg_logN;

    float eps_c[DX], mu2_c[DX], obs_c[DY], u_c = 0.f;
    int idx_c = 0;
    auto load_inputs = [&](int t, float (&e)[DX], float (&m)[DX], float (&o)[DY], float& uu, int& ii) {
        const size_t tb = (size_t)t * B + b;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            e[d] = a.eps[(tb * DX + d) * N + n];
            m[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) o[k] = a.obs[tb * DY + k];
        if (a.resample) {
            if (a.idx_in) ii = a.idx_in[tb * N + n];
            else uu = a.u[tb * N + n];
        }
    };
    load_inputs(0, eps_c, mu2_c, obs_c, u_c, idx_c);
    __syncthreads();  // weights visible

    SEC_INIT(filter_fwd)
    // t = 0 (its own proposal constants) is peeled: the loop proper then carries one set of step constants
    // instead of selecting between two every step (scalar-register pressure: spills cost v_readlane + s_nop)
    auto step = [&](auto first_tag, const int t) {
        SEC(0);   // (lpp) loop overhead
        const size_t tb = (size_t)t * B + b;
        const StepK<DX> K = decltype(first_tag)::value ? K0 : K1;
        float eps_n[DX], mu2_n[DX], obs_n[DY], u_n = 0.f;
        int idx_n = 0;
        if (t + 1 < T) load_inputs(t + 1, eps_n, mu2_n, obs_n, u_n, idx_n);

        SEC(1);   // (lpp) issue of the prefetch loads
        // ---- proposal (SVO.py:186-197), densities: the same in the four lanes of the particle --------
        float mu[DX], x[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            mu[d] = a.two_q ? K.c[d] * fmaf(K.i1[d], mean1[d], K.i2[d] * mu2_c[d]) : mean1[d];
            x[d] = fmaf(K.c[d], eps_c[d], mu[d]);
        }
        const float q_lp = diag_lp<DX>(x, mu, K.ic, K.lq);
        const float f_lp = diag_lp<DX>(x, fmean, K.ifs, K.lf);

        SEC(2);   // (lpp) proposal, q / f densities
        // ---- the MLPs of the step, hidden units split over the quad ------------------------------------
        // (the lane's weight slice is loop-invariant: the compiler keeps it in VGPRs when the budget allows;
        //  otherwise an opaque LDS offset per step makes it re-read the slice instead of spilling it)
        int zo = 0;
        if (kOpaque) asm volatile("" : "+v"(zo));
        float gm[DY], p1[DX], fm[DX];
        MG::template eval_part<P>(wg + zo, p, x, gm);
        MQ::template eval_part<P>(wq1 + zo, p, x, p1);
        if (!a.bootstrap) MQ::template eval_part<P>(wfm + zo, p, x, fm);
#pragma unroll
        for (int k = 0; k < DY; ++k) gm[k] = group_sum<P>(gm[k]);
        if (a.emission) {
#pragma unroll
            for (int k = 0; k < DY; ++k) gm[k] = emis_mean(gm[k]);
        }
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            p1[d] = group_sum<P>(p1[d]);
            fm[d] = a.bootstrap ? p1[d] : group_sum<P>(fm[d]);
        }
        const float g_lp = diag_lp<DY>(obs_c, gm, isg, lg);
        float lw = f_lp + g_lp - q_lp + lnw;
        if (!valid) lw = ninf;

        if (valid) {   // history: one lane of the quad per array
            if (p == 0) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.X[(tb * DX + d) * N + n] = x[d];
                a.logW[tb * N + n] = lw;
            } else if (p == 1) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.Fm[(tb * DX + d) * N + n] = fm[d];
            } else if (p == 2 && a.P1) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.P1[(tb * DX + d) * N + n] = p1[d];
            }
        }

        SEC(3);   // (lpp) MLPs, quad sums, weight, history stores
        // ---- log-sum-exp over particles + CDF: per-wave maxima, one barrier ------------------------------
        const float wmx = wave_max(lw);
        const float wbase = (wmx == ninf) ? 0.f : wmx;
        const float w = (valid && p == 0) ? exp2_fast((lw - wbase) * kLog2e) : 0.f;
        float sc = wave_incl_scan(w, lane);
        if (lane == 63) {
            red[wave] = wmx;
            red[16 + wave] = sc;
        }
        if (a.resample) {   // stage the pre-resampling particle for the gather (one lane of the quad per array)
            if (p == 1) {
#pragma unroll
                for (int d = 0; d < DX; ++d) sx[d * NPT + pn] = x[d];
            } else if (p == 2) {
#pragma unroll
                for (int d = 0; d < DX; ++d) sp[d * NPT + pn] = p1[d];
            } else if (p == 3 && !a.bootstrap) {
#pragma unroll
                for (int d = 0; d < DX; ++d) sf[d * NPT + pn] = fm[d];
            }
        }
        SEC(4);   // (lpp) wave max / scan, staging writes
        __syncthreads();
        SEC(5);   // (lpp) barrier
        // combine the (max, sum) pairs of the <= 16 waves in the lanes of a row: lane i holds wave i
        const int li = lane & 15;
        const float m_i = li < nw ? red[li] : ninf;
        const float s_i = li < nw ? red[16 + li] : 0.f;
        const float gmx = lane_bcast(group_max<16>(m_i), 0);
        const float gbase = (gmx == ninf) ? 0.f : gmx;
        const float v_i = s_i * exp2_fast((m_i - gbase) * kLog2e);        // (empty wave: 0 * exp2(-inf) = 0)
        const float pre = group_incl_scan<16>(v_i, li);
        const float total = lane_bcast(pre, nw - 1);
        const float off = uwave > 0 ? lane_bcast(pre, max(uwave - 1, 0)) : 0.f;   // sum over the waves before this one
        sc = fmaf(sc, exp2_fast((wbase - gbase) * kLog2e), off);
        const float lse_t = fmaf(kLn2, log2_fast(total), gmx);
        if (tid == 3) a.lse[tb] = lse_t;

        SEC(6);   // (lpp) cross-wave combination
        if (a.resample) {
            if (p == 0) {
                cdf[pn] = sc;
                if ((pn & 15) == 15) piv[pn >> 4] = sc;
            }
            __syncthreads();
            SEC(7);   // (lpp) cdf store + barrier
            int idx;
            if (a.idx_in) {
                idx = idx_c;
            } else {
                // idx = #{k : cdf[k] <= u * total} (SVO.py:266-300 as defined in the oracle), two 16-ary rounds
                const float target = u_c * total;
                const float4 pv = *reinterpret_cast<const float4*>(piv + 4 * p);
                float c1 = (pv.x <= target ? 1.f : 0.f) + (pv.y <= target ? 1.f : 0.f) +
                           (pv.z <= target ? 1.f : 0.f) + (pv.w <= target ? 1.f : 0.f);
                c1 = group_sum<P>(c1);
                const int blk = min((int)c1, (NPT >> 4) - 1);
                const float4 cv = *reinterpret_cast<const float4*>(cdf + 16 * blk + 4 * p);
                float c2 = (cv.x <= target ? 1.f : 0.f) + (cv.y <= target ? 1.f : 0.f) +
                           (cv.z <= target ? 1.f : 0.f) + (cv.w <= target ? 1.f : 0.f);
                c2 = group_sum<P>(c2);
                idx = min(16 * blk + (int)c2, N - 1);
            }
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                x[d] = sx[d * NPT + idx];
                mean1[d] = sp[d * NPT + idx];
                fmean[d] = a.bootstrap ? mean1[d] : sf[d * NPT + idx];
            }
            if (valid) {
                if (p == 3) a.idx_out[tb * N + n] = idx;
                if (p == 0) {
#pragma unroll
                    for (int d = 0; d < DX; ++d) a.Xanc[(tb * DX + d) * N + n] = x[d];
                }
            }
            lnw = neg_logN;
            SEC(8);   // (lpp) search, gather, stores
            __syncthreads();  // staged arrays / red[] are rewritten next step
            SEC(9);   // (lpp) barrier
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                mean1[d] = p1[d];
                fmean[d] = fm[d];
            }
            if (valid) {
                if (p == 3) a.idx_out[tb * N + n] = n;
                if (p == 0) {
#pragma unroll
                    for (int d = 0; d < DX; ++d) a.Xanc[(tb * DX + d) * N + n] = x[d];
                }
            }
            lnw = lw - lse_t;
            __syncthreads();  // red[] reuse
        }

#pragma unroll
        for (int d = 0; d < DX; ++d) {
            eps_c[d] = eps_n[d];
            mu2_c[d] = mu2_n[d];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) obs_c[k] = obs_n[k];
        u_c = u_n;
        idx_c = idx_n;
    };
    step(std::true_type{}, 0);
    for (int t = 1; t < T; ++t) step(std::false_type{}, t);
}

template <int DX, int DY, int H>
static int launch_filter(const FilterArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    // latency-bound regime (N <= 128: at most 512 lanes, 256 VGPRs each): four lanes per particle.  Only the
    // shapes that compile without scratch use it; the rest keep one lane per particle.
    // (two hidden layers: always -- a lane then walks H / 4 units of the H x H layer instead of all H, and this kernel is a
    //  latency chain of one workgroup per sequence: C* sizes at "64,64", filter_fwd 6.4 -> 2.0 ms)
    constexpr bool kLppOk = (H % 16 == 0) && (PSVO_L == 2 || (H <= 32 && DX <= 3) || H == 16);
    if constexpr (kLppOk) {
        if (a.N <= 128) {
            const int NT4 = (4 * a.N + 63) & ~63;
            const int NPT = NT4 / 4;
            const size_t lds4 = sizeof(float) * (2 * MQ::kSize + MG::kSize + NPT + 16 + 3 * DX * NPT + 32);
            clear_hip_error();
            hipLaunchKernelGGL((filter_fwd_lpp_kernel<DX, DY, H, 512>), dim3(a.B), dim3(NT4), lds4, stream, a);
            return launch_status();
        }
    }
    const int NT = (a.N + 63) & ~63;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + NT + 3 * DX * NT + 48);
    clear_hip_error();
    // the register budget follows the workgroup size: <= 256 threads is one wave per SIMD, so
    // the compiler may keep every MLP weight resident in VGPRs
    if (NT <= 256 && a.emission)
        hipLaunchKernelGGL((filter_fwd_kernel<DX, DY, H, 256, 1>), dim3(a.B), dim3(NT), lds, stream, a);
    else if (NT <= 256)
        hipLaunchKernelGGL((filter_fwd_kernel<DX, DY, H, 256, 0>), dim3(a.B), dim3(NT), lds, stream, a);
    else
        hipLaunchKernelGGL((filter_fwd_kernel<DX, DY, H, 512>), dim3(a.B), dim3(NT), lds, stream, a);
    return launch_status();
}

template <int DX, int DY>
static int dispatch_h(const FilterArgs& a, int H, hipStream_t s) {
    switch (H) {
#if PSVO_L == 1   // (two hidden layers: widths 32 and 64; narrower ones are zero-padded upstream)
        case 16: return launch_filter<DX, DY, 16>(a, s);
#endif
        case 32: return launch_filter<DX, DY, 32>(a, s);
        case 64: return launch_filter<DX, DY, 64>(a, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX>
static int dispatch_dy(const FilterArgs& a, int Dy, int H, hipStream_t s) {
    switch (Dy) {
        case 1: return dispatch_h<DX, 1>(a, H, s);
        case 2: return dispatch_h<DX, 2>(a, H, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // inline namespace PSVO_LNS
}  // namespace psvo

PSVO_L2_DECL(psvo_filter_forward)
PSVO_ENTRY(psvo_filter_forward)(const psvo_desc* desc, const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g,
                                   const float* sig_q1, const float* sig_q2, const float* sig_f,
                                   const float* sig_g, const float* mu2, const float* m0, const float* sig0,
                                   const float* fm0, const float* fsig0, const float* obs, const float* eps,
                                   const float* u, const int32_t* idx_in, float* X, float* Xanc, float* Fm, float* P1,
                                   float* logW, int32_t* idx_out, float* lse, void* stream) {
    using namespace psvo;
    if (!desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;
#if PSVO_L == 1
    if (desc && desc->layers == 2)
        return psvo_filter_forward_l2(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0, obs,
            eps, u, idx_in, X, Xanc, Fm, P1, logW, idx_out, lse, stream);
#endif
    if (!mlp_layers_ok(q1) || !mlp_layers_ok(f) || !mlp_layers_ok(g)) return PSVO_ERR_INVALID;
    if (!desc || !q1 || !g || !sig_q1 || !sig_g || !m0 || !sig0 || !fm0 || !fsig0 || !obs || !eps || !X ||
        !Xanc || !Fm || !logW || !idx_out || !lse)
        return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T <= 0 || desc->N <= 0) return PSVO_ERR_INVALID;
    if (desc->two_q && (!mu2 || !sig_q2)) return PSVO_ERR_INVALID;
    if (!desc->bootstrap && (!f || !sig_f)) return PSVO_ERR_INVALID;
    if (desc->resample && !u && !idx_in) return PSVO_ERR_INVALID;
    if (desc->N > 512) return PSVO_ERR_UNSUPPORTED;

    FilterArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N;
    a.resample = desc->resample; a.two_q = desc->two_q; a.bootstrap = desc->bootstrap; a.emission = desc->emission;
    a.q1 = *q1; a.f = desc->bootstrap ? *q1 : *f; a.g = *g;
    a.sig_q1 = sig_q1; a.sig_q2 = sig_q2; a.sig_f = sig_f; a.sig_g = sig_g;
    a.mu2 = mu2; a.m0 = m0; a.sig0 = sig0; a.fm0 = fm0; a.fsig0 = fsig0;
    a.obs = obs; a.eps = eps; a.u = u; a.idx_in = idx_in;
    a.X = X; a.Xanc = Xanc; a.Fm = Fm; a.P1 = P1; a.logW = logW; a.idx_out = idx_out; a.lse = lse;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return dispatch_dy<2>(a, desc->Dy, desc->H, s);
        case 3: return dispatch_dy<3>(a, desc->Dy, desc->H, s);
        case 4: return dispatch_dy<4>(a, desc->Dy, desc->H, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
