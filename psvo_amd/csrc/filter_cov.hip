// Forward particle filter and its reverse pass with STATE-DEPENDENT diagonal scales: the reference's `output_cov` and
// `diag_cov` flags (src/runner_flag.py:67-70, 221-222).  Every MLP then carries a second output head (`sigma_layer`,
// src/transformation/MLP.py:40-46) and the scale of every distribution becomes a function of the distribution's input,
//     sigma(x) = sigma_con + 0.1 * (exp(h(x) W_sigma + b_sigma) + 1e-6)      (MLP.py:58-61, src/distribution/mvn.py:66-71),
// sigma_con the state-independent vector of get_sigma (mvn.py:80-90).  The loop is SVO.SMC (src/SMC/SVO.py:60-180) as in
// filter_fwd.hip / filter_bwd.hip; what changes is that no scale is a launch constant any more:
//   * a per-particle MLP is handed over with W2 = [mu_layer | sigma_layer] (H, 2 Dout), b2 (2 Dout) and evaluated once for
//     both heads (MlpLds<DIN, H, 2 DOUT>); the head's scale travels with the mean -- history rows Fs / P1s beside Fm / P1,
//     the resampling gather moves (x, mean, scale) together, and the reverse pass scatters d mean AND d scale into the parent;
//   * the hoisted distributions (q0 on the X0 feature, q2 on the step's feature; f on the X0 feature) arrive as mean and
//     scale per row: sig0 / fsig0 (B, Dx), sig2 (T, B, Dx) -- their heads are B T rows evaluated outside;
//   * the product of two Gaussians on scales (SVO.py:186-197) is formed per particle and step, and its reverse returns the
//     gradient of both scales: d s1 into the parent's head, d s2 as rows summed over the particles.
// Work decomposition as in the plain kernels' generic shape: one persistent workgroup per sequence, lane = particle.
// The reverse pass writes, for every MLP evaluation, the gradient w.r.t. BOTH heads' outputs (rows dP / dPs, dF / dFs,
// dG / dGs; the *s rows are w.r.t. the raw head output, i.e. already multiplied by d sigma / d raw = 0.1 exp(raw)) and
// psvo_mlp_wgrad turns rows into weights, one launch per head.  The reference's Poisson emission (desc->emission = 1) drops
// MLP_g's covariance head (`lambdas, _ = transform(Input)`, src/distribution/poisson.py:33): unit scale, softplus mean, and
// the head's rows dGs are zero.
#include "common.h"

namespace psvo {
namespace cov {

// sigma_con + 0.1 * (exp(raw) + 1e-6)
__device__ __forceinline__ float head_sigma(float con, float raw) {
    return con + 0.1f * (exp2_fast(raw * kLog2e) + 1e-6f);
}
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float ln(float x) { return kLn2 * log2_fast(x); }

struct FwdArgs {
    int B, T, N;
    int resample, two_q, bootstrap, emission;
    psvo_mlp q1, f, g;
    const float *sc_q1, *sc_f, *sc_g;
    const float *mu2, *sig2, *m0, *sig0, *fm0, *fsig0, *obs, *eps, *u;
    const int32_t* idx_in;
    float *X, *Xanc, *Fm, *Fs, *P1, *P1s, *logW;
    int32_t* idx_out;
    float* lse;
};

template <int DX, int DY, int H, int MAXT>
__global__ void __launch_bounds__(MAXT) filter_cov_fwd_kernel(const FwdArgs a) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    constexpr bool kRolled = (MAXT > 256) || (2 * MQ::kSize + MG::kSize > 330);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, nw = NT >> 6;
    const int b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const bool valid = tid < N;
    const int n = valid ? tid : N - 1;

    float* wq1 = smem;
    float* wf = wq1 + MQ::kSize;
    float* wg = wf + MQ::kSize;
    float* cdf = wg + MG::kSize;   // [NT]
    float* sx = cdf + NT;          // [DX][NT]    staged X_t
    float* sp = sx + DX * NT;      // [2 DX][NT]  staged MLP_q1(X_t): mean | scale
    float* sf = sp + 2 * DX * NT;  // [2 DX][NT]  staged MLP_f(X_t) (unused when bootstrap)
    float* red = sf + 2 * DX * NT; // 48

    MQ::load(wq1, a.q1, tid, NT);
    if (!a.bootstrap) MQ::load(wf, a.f, tid, NT);
    MG::load(wg, a.g, tid, NT);
    const float* wfm = a.bootstrap ? wq1 : wf;

    float cq1[DX], cf[DX], cg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        cq1[d] = a.sc_q1[d];
        cf[d] = a.bootstrap ? a.sc_q1[d] : a.sc_f[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) cg[e] = a.sc_g[e];
    const float neg_logN = -logf((float)N);
    const float ninf = -__builtin_huge_valf();

    // first term of the proposal and the transition density of the particle this lane continues: mean and scale
    float mean1[DX], s1[DX], fmean[DX], fs[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        mean1[d] = a.m0[b * DX + d];
        s1[d] = a.sig0[b * DX + d];
        fmean[d] = a.fm0[b * DX + d];
        fs[d] = a.fsig0[b * DX + d];
    }
    float lnw = neg_logN;

    float eps_c[DX], mu2_c[DX], s2_c[DX], obs_c[DY], u_c = 0.f;
    int idx_c = 0;
    auto load_inputs = [&](int t, float (&e)[DX], float (&m)[DX], float (&s)[DX], float (&o)[DY], float& uu, int& ii) {
        const size_t tb = (size_t)t * B + b;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            e[d] = a.eps[(tb * DX + d) * N + n];
            m[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
            s[d] = a.two_q ? a.sig2[tb * DX + d] : 1.f;
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) o[k] = a.obs[tb * DY + k];
        if (a.resample) {
            if (a.idx_in) ii = a.idx_in[tb * N + n];
            else uu = a.u[tb * N + n];
        }
    };
    load_inputs(0, eps_c, mu2_c, s2_c, obs_c, u_c, idx_c);
    __syncthreads();  // weights visible

    for (int t = 0; t < T; ++t) {
        const size_t tb = (size_t)t * B + b;
        float eps_n[DX], mu2_n[DX], s2_n[DX], obs_n[DY], u_n = 0.f;
        int idx_n = 0;
        if (t + 1 < T) load_inputs(t + 1, eps_n, mu2_n, s2_n, obs_n, u_n, idx_n);

        // ---- proposal: product of two diagonal Gaussians on *scales* (SVO.py:186-197), per particle --------------
        float mu[DX], x[DX], ic[DX], ifs[DX];
        float lq = -DX * kHalfLog2Pi, lf = -DX * kHalfLog2Pi;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            float c;
            if (a.two_q) {
                const float i1 = rcp(s1[d]), i2 = rcp(s2_c[d]);
                ic[d] = i1 + i2;
                c = rcp(ic[d]);
                mu[d] = c * fmaf(i1, mean1[d], i2 * mu2_c[d]);
            } else {
                c = s1[d];
                ic[d] = rcp(c);
                mu[d] = mean1[d];
            }
            x[d] = fmaf(c, eps_c[d], mu[d]);
            lq -= ln(c);
            ifs[d] = rcp(fs[d]);
            lf -= ln(fs[d]);
        }
        const float q_lp = diag_lp<DX>(x, mu, ic, lq);
        const float f_lp = diag_lp<DX>(x, fmean, ifs, lf);

        // ---- emission: mean and scale from the two heads of MLP_g ------------------------------------------------------
        float go[2 * DY], gm[DY], isg[DY];
        MG::template eval<kRolled>(wg, x, go);
        float lg = -DY * kHalfLog2Pi;
#pragma unroll
        for (int k = 0; k < DY; ++k) {
            if (a.emission) {
                gm[k] = emis_mean(go[k]);
                isg[k] = 1.f;
            } else {
                gm[k] = go[k];
                const float sg = head_sigma(cg[k], go[DY + k]);
                isg[k] = rcp(sg);
                lg -= ln(sg);
            }
        }
        const float g_lp = diag_lp<DY>(obs_c, gm, isg, lg);
        float lw = f_lp + g_lp - q_lp + lnw;
        if (!valid) lw = ninf;

        // ---- next step's proposal / transition mean and scale of every pre-resampling particle ---------------------------
        float po[2 * DX], fo[2 * DX];
        MQ::template eval<kRolled>(wq1, x, po);
#pragma unroll
        for (int d = 0; d < DX; ++d) po[DX + d] = head_sigma(cq1[d], po[DX + d]);
        if (a.bootstrap) {
#pragma unroll
            for (int d = 0; d < 2 * DX; ++d) fo[d] = po[d];
        } else {
            MQ::template eval<kRolled>(wfm, x, fo);
#pragma unroll
            for (int d = 0; d < DX; ++d) fo[DX + d] = head_sigma(cf[d], fo[DX + d]);
        }
        if (valid) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                a.X[(tb * DX + d) * N + n] = x[d];
                a.Fm[(tb * DX + d) * N + n] = fo[d];
                a.Fs[(tb * DX + d) * N + n] = fo[DX + d];
                if (!a.bootstrap) {
                    a.P1[(tb * DX + d) * N + n] = po[d];
                    a.P1s[(tb * DX + d) * N + n] = po[DX + d];
                }
            }
            a.logW[tb * N + n] = lw;
        }

        // ---- log-sum-exp over particles and multinomial ancestors (SVO.py:266-300) -----------------------------------
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

        if (a.resample) {
            cdf[tid] = sc;
#pragma unroll
            for (int d = 0; d < DX; ++d) sx[d * NT + tid] = x[d];
#pragma unroll
            for (int d = 0; d < 2 * DX; ++d) {
                sp[d * NT + tid] = po[d];
                if (!a.bootstrap) sf[d * NT + tid] = fo[d];
            }
            __syncthreads();
            int idx;
            if (a.idx_in) {
                idx = idx_c;
            } else {   // count of cdf entries <= u * total (cdf is non-decreasing)
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
                s1[d] = sp[(DX + d) * NT + idx];
                fmean[d] = a.bootstrap ? mean1[d] : sf[d * NT + idx];
                fs[d] = a.bootstrap ? s1[d] : sf[(DX + d) * NT + idx];
            }
            if (valid) {
                a.idx_out[tb * N + n] = idx;
#pragma unroll
                for (int d = 0; d < DX; ++d) a.Xanc[(tb * DX + d) * N + n] = x[d];
            }
            lnw = neg_logN;
            __syncthreads();  // staged arrays are rewritten next step
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                mean1[d] = po[d];
                s1[d] = po[DX + d];
                fmean[d] = fo[d];
                fs[d] = fo[DX + d];
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
            s2_c[d] = s2_n[d];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) obs_c[k] = obs_n[k];
        u_c = u_n;
        idx_c = idx_n;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// reverse pass
// ---------------------------------------------------------------------------------------------------------------------------
struct BwdArgs {
    int B, T, N;
    int resample, two_q, bootstrap, emission;
    psvo_mlp q1, f, g;
    const float *sc_q1, *sc_f, *sc_g;
    const float *mu2, *sig2, *m0, *sig0, *fm0, *fsig0, *obs, *eps;
    const float *X, *Fm, *Fs, *P1, *P1s, *logW, *lse;
    const int32_t* idx;
    const float *dlse, *dFm_ext, *dFs_ext, *dlogW_ext;
    float *dP, *dPs, *dF, *dFs, *dG, *dGs;
    float *dm0, *dsig0, *dfm0, *dfsig0;
    float* sacc;       // (B, 2 Dx + Dy): sums of d sigma over the rows of the q1 / f / g heads (= d sigma_con)
    float* dm2_rows;   // (T,B,Dx,N) per-particle d mu2, summed over N afterwards
    float* ds2_rows;   // (T,B,Dx,N) per-particle d sig2
    float* scanRec;    // affine-scan path: float4 planes [(t, b), float4 index, particle] of the step records; scanPart (T,B,2Dx+Dy)
    float* scanPart;
};

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
__global__ void __launch_bounds__(MAXT) filter_cov_bwd_kernel(const BwdArgs a) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    constexpr bool kRolled = (MAXT > 256) || (2 * MQ::kSize + MG::kSize > 330);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = blockDim.x, nw = NT >> 6;
    const int b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const bool valid = tid < N;
    const int n = valid ? tid : N - 1;
    const bool boot = a.bootstrap != 0;

    float* wq1 = smem;
    float* wf = wq1 + MQ::kSize;
    float* wg = wf + MQ::kSize;
    const int CP = 2 * DX * NT;            // one buffer of scatter targets: d mean | d scale
    float* accP = wg + MG::kSize;          // [2][2 DX][NT] into MLP_q1's outputs (+ MLP_f's == the same when bootstrap)
    float* accF = accP + 2 * CP;           // [2][2 DX][NT] into MLP_f's outputs (only when !bootstrap)
    float* red = accF + (boot ? 0 : 2 * CP);   // 16

    MQ::load(wq1, a.q1, tid, NT);
    if (!boot) MQ::load(wf, a.f, tid, NT);
    MG::load(wg, a.g, tid, NT);
    const float* wfm = boot ? wq1 : wf;
    for (int i = tid; i < (boot ? 2 : 4) * CP; i += NT) accP[i] = 0.f;

    float cq1[DX], cf[DX], cg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        cq1[d] = a.sc_q1[d];
        cf[d] = boot ? a.sc_q1[d] : a.sc_f[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) cg[e] = a.sc_g[e];

    float acc_q[DX], acc_f[DX], acc_g[DY];     // sums of d sigma over this lane's rows
#pragma unroll
    for (int d = 0; d < DX; ++d) acc_q[d] = acc_f[d] = 0.f;
#pragma unroll
    for (int e = 0; e < DY; ++e) acc_g[e] = 0.f;
    float dlnw = 0.f;   // IWAE: gradient w.r.t. the normalised log-weight carried into step t + 1

    float m0r[DX], s0r[DX], fm0r[DX], fs0r[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        m0r[d] = a.m0[b * DX + d];
        s0r[d] = a.sig0[b * DX + d];
        fm0r[d] = a.fm0[b * DX + d];
        fs0r[d] = a.fsig0[b * DX + d];
    }
    auto load_anc = [&](int t) -> int {   // ancestor of particle n at step t (its parent lives at t - 1)
        return (t >= 1 && a.resample) ? a.idx[((size_t)(t - 1) * B + b) * N + n] : n;
    };
    // everything step t reads from memory, requested one step ahead (issue only: no arithmetic on the loaded values here).
    // (plain arrays rather than a struct: a struct of arrays copied per step ends up in scratch)
    //   px / pe / pm2 / ps2 / py: X, eps, mu2, sig2, obs of the step;  ppm / pps, pqm / pqs: the parent's MLP_f and MLP_q1
    //   outputs (mean | scale);  pofs / pops: this particle's own head scales at the step;  pdfm / pdfs: upstream gradients
    //   w.r.t. Fm[t][n], Fs[t][n];  psc: logW, lse, d lse, d logW
    auto load_step = [&](int t, int anc, float (&px)[DX], float (&pe)[DX], float (&pm2)[DX], float (&ps2)[DX],
                         float (&py)[DY], float (&ppm)[DX], float (&pps)[DX], float (&pqm)[DX], float (&pqs)[DX],
                         float (&pofs)[DX], float (&pops)[DX], float (&pdfm)[DX], float (&pdfs)[DX], float (&psc)[4]) {
        const size_t tb = (size_t)t * B + b;
        const size_t tp = (t == 0) ? tb : tb - B;      // (t = 0 has no parent: values replaced at use)
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const size_t o = (tb * DX + d) * N + n, op = (tp * DX + d) * N + anc;
            px[d] = a.X[o];
            pe[d] = a.eps[o];
            pm2[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
            ps2[d] = a.two_q ? a.sig2[tb * DX + d] : 1.f;
            ppm[d] = a.Fm[op];
            pps[d] = a.Fs[op];
            pqm[d] = boot ? 0.f : a.P1[op];
            pqs[d] = boot ? 1.f : a.P1s[op];
            pofs[d] = a.Fs[o];
            pops[d] = boot ? 1.f : a.P1s[o];
            pdfm[d] = a.dFm_ext ? a.dFm_ext[o] : 0.f;
            pdfs[d] = a.dFs_ext ? a.dFs_ext[o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) py[k] = a.obs[tb * DY + k];
        psc[0] = a.logW[tb * N + n];
        psc[1] = a.lse[tb];
        psc[2] = a.dlse ? a.dlse[tb] : 0.f;
        psc[3] = a.dlogW_ext ? a.dlogW_ext[tb * N + n] : 0.f;
    };
    float c_x[DX], c_e[DX], c_m2[DX], c_s2[DX], c_y[DY], c_pm[DX], c_ps[DX], c_qm[DX], c_qs[DX], c_ofs[DX], c_ops[DX],
        c_dfm[DX], c_dfs[DX], c_sc[4];
    int c_anc = load_anc(T - 1);
    int anc_next = load_anc(T - 2);
    load_step(T - 1, c_anc, c_x, c_e, c_m2, c_s2, c_y, c_pm, c_ps, c_qm, c_qs, c_ofs, c_ops, c_dfm, c_dfs, c_sc);
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * B + b;
        const bool first = (t == 0);
        float* curP = accP + (t & 1) * CP;
        float* nxtP = accP + ((t + 1) & 1) * CP;
        float* curF = accF + (t & 1) * CP;
        float* nxtF = accF + ((t + 1) & 1) * CP;

        float n_x[DX], n_e[DX], n_m2[DX], n_s2[DX], n_y[DY], n_pm[DX], n_ps[DX], n_qm[DX], n_qs[DX], n_ofs[DX], n_ops[DX],
            n_dfm[DX], n_dfs[DX], n_sc[4];
        int n_anc = n;
        if (t >= 1) {
            n_anc = anc_next;
            load_step(t - 1, n_anc, n_x, n_e, n_m2, n_s2, n_y, n_pm, n_ps, n_qm, n_qs, n_ofs, n_ops, n_dfm, n_dfs, n_sc);
            anc_next = load_anc(t - 2);
        }

        // ---- forward quantities of this step --------------------------------------------------------------------------
        const int anc = c_anc;
        float x[DX], mean1[DX], s1[DX], fmean[DX], fs[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            x[d] = c_x[d];
            fmean[d] = first ? fm0r[d] : c_pm[d];
            fs[d] = first ? fs0r[d] : c_ps[d];
            mean1[d] = first ? m0r[d] : (boot ? c_pm[d] : c_qm[d]);
            s1[d] = first ? s0r[d] : (boot ? c_ps[d] : c_qs[d]);
        }
        float i1[DX], i2[DX], ic[DX], c[DX], mu[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            if (a.two_q) {
                i1[d] = rcp(s1[d]);
                i2[d] = rcp(c_s2[d]);
                ic[d] = i1[d] + i2[d];
                c[d] = rcp(ic[d]);
                mu[d] = c[d] * fmaf(i1[d], mean1[d], i2[d] * c_m2[d]);
            } else {
                i1[d] = i2[d] = 0.f;
                c[d] = s1[d];
                ic[d] = rcp(c[d]);
                mu[d] = mean1[d];
            }
        }

        // ---- gradient w.r.t. logW_t[n] -----------------------------------------------------------------------------------
        const float sm = valid ? exp2_fast((c_sc[0] - c_sc[1]) * kLog2e) : 0.f;
        float dlw = c_sc[2] * sm + c_sc[3];
        if (!a.resample) {
            const float tot = block_sum(dlnw, red, wave, lane, nw);
            dlw += dlnw - sm * tot;
        }
        if (!valid) dlw = 0.f;
        dlnw = first ? 0.f : dlw;

        // ---- emission -----------------------------------------------------------------------------------------------------
        float dx[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dx[d] = 0.f;
        {
            float go[2 * DY], dgo[2 * DY];
            MG::template eval<kRolled>(wg, x, go);
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                if (a.emission) {      // unit-scale normal around softplus(raw) + 1e-6; the covariance head is not used
                    dgo[k] = dlw * (c_y[k] - emis_mean(go[k])) * emis_dmean(go[k]);
                    dgo[DY + k] = 0.f;
                } else {
                    const float hx = 0.1f * exp2_fast(go[DY + k] * kLog2e);       // d sigma / d raw
                    const float isg = rcp(cg[k] + (hx + 1e-7f));
                    const float z = (c_y[k] - go[k]) * isg;
                    dgo[k] = dlw * z * isg;
                    const float dsg = dlw * (z * z - 1.f) * isg;
                    dgo[DY + k] = dsg * hx;
                    acc_g[k] += dsg;
                }
                if (valid) {
                    a.dG[(tb * DY + k) * N + n] = dgo[k];
                    a.dGs[(tb * DY + k) * N + n] = dgo[DY + k];
                }
            }
            MG::template bwd_input<kRolled>(wg, x, dgo, dx);
        }
        // ---- transition and proposal densities ------------------------------------------------------------------------------
        float dfmean[DX], dfsc[DX], dc[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float ifs = rcp(fs[d]);
            const float z = (x[d] - fmean[d]) * ifs;
            const float tf = dlw * z * ifs;
            dx[d] -= tf;
            dfmean[d] = tf;
            dfsc[d] = dlw * (z * z - 1.f) * ifs;
            dc[d] = dlw * ic[d];                       // -d q_lp / d c = +1 / c
        }
        // ---- MLP_q1(x_t), MLP_f(x_t): gradients scattered here by step t + 1 (+ upstream) ------------------------------------
        float dPo[2 * DX], dFo[2 * DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            float dpm = curP[d * NT + tid], dps = curP[(DX + d) * NT + tid];
            curP[d * NT + tid] = 0.f;
            curP[(DX + d) * NT + tid] = 0.f;
            if (boot) {
                dpm += c_dfm[d];
                dps += c_dfs[d];
                dFo[d] = dFo[DX + d] = 0.f;
                if (!valid) dpm = dps = 0.f;
                acc_q[d] += dps;
                dPo[d] = dpm;
                dPo[DX + d] = dps * (c_ofs[d] - cq1[d] - 1e-7f);       // 0.1 exp(raw) of this particle's own head
            } else {
                float dfm = curF[d * NT + tid] + c_dfm[d], dfs = curF[(DX + d) * NT + tid] + c_dfs[d];
                curF[d * NT + tid] = 0.f;
                curF[(DX + d) * NT + tid] = 0.f;
                if (!valid) dpm = dps = dfm = dfs = 0.f;
                acc_q[d] += dps;
                acc_f[d] += dfs;
                dPo[d] = dpm;
                dPo[DX + d] = dps * (c_ops[d] - cq1[d] - 1e-7f);
                dFo[d] = dfm;
                dFo[DX + d] = dfs * (c_ofs[d] - cf[d] - 1e-7f);
            }
        }
        if (valid) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                a.dP[(tb * DX + d) * N + n] = dPo[d];
                a.dPs[(tb * DX + d) * N + n] = dPo[DX + d];
                if (!boot) {
                    a.dF[(tb * DX + d) * N + n] = dFo[d];
                    a.dFs[(tb * DX + d) * N + n] = dFo[DX + d];
                }
            }
        }
        MQ::template bwd_input<kRolled>(wq1, x, dPo, dx);
        if (!boot) MQ::template bwd_input<kRolled>(wfm, x, dFo, dx);

        // ---- x = mu + c eps;  two_q: c = 1 / (1/s1 + 1/s2), mu = c (mean1 / s1 + mu2 / s2) --------------------------------------
        float dmean1[DX], ds1[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float dmu = dx[d];
            const float dct = fmaf(dmu, c_e[d], dc[d]);
            if (a.two_q) {
                const float dA = dmu * c[d];                         // A = mean1 / s1 + mu2 / s2 = mu / c
                const float dic = -(dct + dmu * mu[d] * ic[d]) * c[d] * c[d];
                const float di1 = fmaf(dA, mean1[d], dic), di2 = fmaf(dA, c_m2[d], dic);
                dmean1[d] = dA * i1[d];
                ds1[d] = -di1 * i1[d] * i1[d];
                if (valid) {
                    a.dm2_rows[(tb * DX + d) * N + n] = dA * i2[d];
                    a.ds2_rows[(tb * DX + d) * N + n] = -di2 * i2[d] * i2[d];
                }
            } else {
                dmean1[d] = dmu;
                ds1[d] = dct;
            }
        }
        // ---- gather backward: scatter-add into the parents (SVO.py:255-257) ---------------------------------------------------
        if (!first) {
            if (valid) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    if (boot) {
                        atomicAdd(&nxtP[d * NT + anc], dmean1[d] + dfmean[d]);
                        atomicAdd(&nxtP[(DX + d) * NT + anc], ds1[d] + dfsc[d]);
                    } else {
                        atomicAdd(&nxtP[d * NT + anc], dmean1[d]);
                        atomicAdd(&nxtP[(DX + d) * NT + anc], ds1[d]);
                        atomicAdd(&nxtF[d * NT + anc], dfmean[d]);
                        atomicAdd(&nxtF[(DX + d) * NT + anc], dfsc[d]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float v1 = block_sum(dmean1[d], red, wave, lane, nw);
                const float v2 = block_sum(ds1[d], red, wave, lane, nw);
                const float v3 = block_sum(dfmean[d], red, wave, lane, nw);
                const float v4 = block_sum(dfsc[d], red, wave, lane, nw);
                if (tid == 0) {
                    a.dm0[b * DX + d] = v1;
                    a.dsig0[b * DX + d] = v2;
                    a.dfm0[b * DX + d] = v3;
                    a.dfsig0[b * DX + d] = v4;
                }
            }
        }
        if (t >= 1) {
            c_anc = n_anc;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                c_x[d] = n_x[d]; c_e[d] = n_e[d]; c_m2[d] = n_m2[d]; c_s2[d] = n_s2[d];
                c_pm[d] = n_pm[d]; c_ps[d] = n_ps[d]; c_qm[d] = n_qm[d]; c_qs[d] = n_qs[d];
                c_ofs[d] = n_ofs[d]; c_ops[d] = n_ops[d]; c_dfm[d] = n_dfm[d]; c_dfs[d] = n_dfs[d];
            }
#pragma unroll
            for (int k = 0; k < DY; ++k) c_y[k] = n_y[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) c_sc[k] = n_sc[k];
        }
        __syncthreads();
    }

    // ---- per-sequence sums of d sigma over the rows of each head -------------------------------------------------------------
    constexpr int NA = 2 * DX + DY;
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float sq = block_sum(acc_q[d], red, wave, lane, nw);
        const float sf = block_sum(acc_f[d], red, wave, lane, nw);
        if (tid == 0) {
            a.sacc[(size_t)b * NA + d] = sq;
            a.sacc[(size_t)b * NA + DX + d] = sf;
        }
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) {
        const float sg = block_sum(acc_g[e], red, wave, lane, nw);
        if (tid == 0) a.sacc[(size_t)b * NA + 2 * DX + e] = sg;
    }
}

// d sigma_con of the three heads: sums of the per-sequence accumulators over the batch, fixed order
__global__ void cov_finalize(const float* __restrict__ sacc, int B, int DX, int DY, int bootstrap, float* dsc_q1,
                             float* dsc_f, float* dsc_g) {
    const int i = threadIdx.x, NA = 2 * DX + DY;
    if (i >= NA) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sacc[(size_t)b * NA + i];
    if (i < DX) dsc_q1[i] = s;
    else if (i < 2 * DX) { if (!bootstrap && dsc_f) dsc_f[i - DX] = s; }
    else dsc_g[i - 2 * DX] = s;
}

// out[r] = sum_l in[r * L + l]: one wave per row
__global__ void __launch_bounds__(256) cov_row_sum(const float* __restrict__ in, long long rows, int L,
                                                   float* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int l = lane; l < L; l += 64) s += in[r * L + l];
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

template <int DX, int DY, int H>
static int launch_fwd(const FwdArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    const int NT = (a.N + 63) & ~63;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + NT + 5 * DX * NT + 48);
    clear_hip_error();
    if (NT <= 256) hipLaunchKernelGGL((filter_cov_fwd_kernel<DX, DY, H, 256>), dim3(a.B), dim3(NT), lds, stream, a);
    else hipLaunchKernelGGL((filter_cov_fwd_kernel<DX, DY, H, 512>), dim3(a.B), dim3(NT), lds, stream, a);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The reverse pass as an AFFINE SCAN (filter_bwd.hip has the idea; bootstrap wiring with resampling).  Here a particle hands
// D = (d mean, d scale) of its MLP_q1 evaluation -- 2 Dx values -- to its parent, and the parent's contribution is affine in it:
//     [alpha dx + tf | beta dx + gamma + dfs],   dx = dx0 + J^T [d mean | hx d scale]   (J: the 2 Dx-output MLP's Jacobian)
// with per-particle alpha = c / s1, beta = -(c mean1 - c^2 (eps + mu / c)) / s1^2, gamma = c^2 (dlw / c) / s1^2 from the product of
// Gaussians on scales (not two_q: alpha = 1, beta = eps, gamma = dlw / c).  Record per (t, sequence, particle):
//     [ M (2 Dx x 2 Dx) | b (2 Dx) | upstream d Fm, d Fs (2 Dx) | ancestor ]  as float4 planes.
// ---------------------------------------------------------------------------------------------------------------------------
template <int DX>
struct CovRec {
    static constexpr int D2 = 2 * DX;
    static constexpr int kB = D2 * D2, kExt = kB + D2, kAnc = kExt + D2;
    static constexpr int REC = (kAnc + 1 + 3) & ~3;
};

// forward quantities of one (t, particle) that both the coefficient and the rows kernel need
template <int DX, int DY>
struct CovStep {
    float x[DX], e[DX], m2[DX], y[DY], mean1[DX], s1[DX], fmean[DX], fs[DX], own[DX];
    float i1[DX], i2[DX], ic[DX], c[DX], mu[DX], dlw;
    int anc;
    __device__ __forceinline__ void load(const BwdArgs& a, int t, int b, int n, bool valid) {
        const int B = a.B, N = a.N;
        const size_t tb = (size_t)t * B + b;
        const bool first = (t == 0);
        anc = first ? n : a.idx[(tb - B) * N + n];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            x[d] = a.X[(tb * DX + d) * N + n];
            e[d] = a.eps[(tb * DX + d) * N + n];
            m2[d] = a.two_q ? a.mu2[tb * DX + d] : 0.f;
            const float s2 = a.two_q ? a.sig2[tb * DX + d] : 1.f;
            fmean[d] = first ? a.fm0[b * DX + d] : a.Fm[((tb - B) * DX + d) * N + anc];
            fs[d] = first ? a.fsig0[b * DX + d] : a.Fs[((tb - B) * DX + d) * N + anc];
            mean1[d] = first ? a.m0[b * DX + d] : fmean[d];
            s1[d] = first ? a.sig0[b * DX + d] : fs[d];
            own[d] = a.Fs[(tb * DX + d) * N + n];
            if (a.two_q) {
                i1[d] = rcp(s1[d]);
                i2[d] = rcp(s2);
                ic[d] = i1[d] + i2[d];
                c[d] = rcp(ic[d]);
                mu[d] = c[d] * fmaf(i1[d], mean1[d], i2[d] * m2[d]);
            } else {
                i1[d] = i2[d] = 0.f;
                c[d] = s1[d];
                ic[d] = rcp(c[d]);
                mu[d] = mean1[d];
            }
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = a.obs[tb * DY + k];
        const float sm = valid ? exp2_fast((a.logW[tb * N + n] - a.lse[tb]) * kLog2e) : 0.f;
        dlw = valid ? (a.dlse ? a.dlse[tb] : 0.f) * sm + (a.dlogW_ext ? a.dlogW_ext[tb * N + n] : 0.f) : 0.f;
    }
};

template <int DX, int DY, int H>
__global__ void __launch_bounds__(512) fcs_coeff_kernel(const BwdArgs a) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    using CR = CovRec<DX>;
    constexpr bool kRolled = true;
    constexpr int D2 = 2 * DX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int t = blockIdx.x, b = blockIdx.y, B = a.B, N = a.N;
    const bool valid = tid < N;
    const int n = valid ? tid : N - 1;
    const size_t tb = (size_t)t * B + b;
    float rec[CR::REC];
#pragma unroll
    for (int i = 0; i < CR::REC; ++i) rec[i] = 0.f;
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        rec[CR::kExt + d] = a.dFm_ext ? a.dFm_ext[(tb * DX + d) * N + n] : 0.f;
        rec[CR::kExt + DX + d] = a.dFs_ext ? a.dFs_ext[(tb * DX + d) * N + n] : 0.f;
    }
    if (t >= 1) {
        float* wq1 = smem;
        float* wg = wq1 + MQ::kSize;
        MQ::load(wq1, a.q1, tid, NT);
        MG::load(wg, a.g, tid, NT);
        CovStep<DX, DY> s;
        s.load(a, t, b, n, valid);
        rec[CR::kAnc] = __int_as_float(s.anc);
        __syncthreads();
        float dx0[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dx0[d] = 0.f;
        {
            float go[2 * DY], dgo[2 * DY];
            MG::template eval<kRolled>(wg, s.x, go);
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                if (a.emission) {
                    dgo[k] = s.dlw * (s.y[k] - emis_mean(go[k])) * emis_dmean(go[k]);
                    dgo[DY + k] = 0.f;
                } else {
                    const float hx = 0.1f * exp2_fast(go[DY + k] * kLog2e);
                    const float isg = rcp(a.sc_g[k] + (hx + 1e-7f));
                    const float z = (s.y[k] - go[k]) * isg;
                    dgo[k] = s.dlw * z * isg;
                    dgo[DY + k] = s.dlw * (z * z - 1.f) * isg * hx;
                }
            }
            MG::template bwd_input<kRolled>(wg, s.x, dgo, dx0);
        }
        float al[DX], be[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float ifs = rcp(s.fs[d]);
            const float z = (s.x[d] - s.fmean[d]) * ifs;
            const float tf = s.dlw * z * ifs;
            const float dfsc = s.dlw * (z * z - 1.f) * ifs;
            const float dc0 = s.dlw * s.ic[d];
            dx0[d] -= tf;
            float ga;
            if (a.two_q) {
                const float q = s.i1[d] * s.i1[d];
                al[d] = s.c[d] * s.i1[d];
                be[d] = -q * (s.c[d] * s.mean1[d] - s.c[d] * s.c[d] * (s.e[d] + s.mu[d] * s.ic[d]));
                ga = q * s.c[d] * s.c[d] * dc0;
            } else {
                al[d] = 1.f;
                be[d] = s.e[d];
                ga = dc0;
            }
            rec[CR::kB + d] = fmaf(al[d], dx0[d], tf);
            rec[CR::kB + DX + d] = fmaf(be[d], dx0[d], ga + dfsc);
        }
#pragma unroll
        for (int k = 0; k < D2; ++k) {
            float ek[D2], col[DX];
#pragma unroll
            for (int j = 0; j < D2; ++j) ek[j] = 0.f;
            ek[k] = (k < DX) ? 1.f : (s.own[k < DX ? 0 : k - DX] - a.sc_q1[k < DX ? 0 : k - DX] - 1e-7f);   // d sigma / d raw of the own head
#pragma unroll
            for (int d = 0; d < DX; ++d) col[d] = 0.f;
            MQ::template bwd_input<kRolled>(wq1, s.x, ek, col);
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                rec[d * D2 + k] = al[d] * col[d];
                rec[(DX + d) * D2 + k] = be[d] * col[d];
            }
        }
    }
    if (valid) {
        float4* dst = reinterpret_cast<float4*>(a.scanRec) + (tb * (CR::REC / 4)) * N + n;
#pragma unroll
        for (int i = 0; i < CR::REC / 4; ++i) dst[(size_t)i * N] = make_float4(rec[4 * i], rec[4 * i + 1], rec[4 * i + 2], rec[4 * i + 3]);
    }
}

// D <- ext + scatter(M D + b), one or four waves per sequence; writes d mean into dP and d SCALE into dPs (the rows kernel turns
// the latter into the raw-head row).  Same structure as psvo::l1::fbs_scan_kernel.
template <int DX, int NWV, int PPL, int DEPTH>
__global__ void __launch_bounds__(64 * NWV) fcs_scan_kernel(const BwdArgs a) {
    using CR = CovRec<DX>;
    constexpr int NTS = 64 * NWV, R4 = CR::REC / 4, D2 = 2 * DX;
    extern __shared__ __attribute__((aligned(16))) float acc[];      // [2][NWV][D2][N]
    const int tid = threadIdx.x, b = blockIdx.x, B = a.B, T = a.T, N = a.N;
    const int CPY = D2 * N, mine = (tid >> 6) * CPY;
    for (int i = tid; i < 2 * NWV * CPY; i += NTS) acc[i] = 0.f;
    float4 rec[DEPTH][PPL][R4];
    auto load = [&](int t, float4 (&r)[PPL][R4]) {
        const size_t tb = (size_t)t * B + b;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            const int n = min(tid + NTS * p, N - 1);
            const float4* src = reinterpret_cast<const float4*>(a.scanRec) + (tb * R4) * N + n;
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
            float* cur = acc + (t & 1) * NWV * CPY;
            float* nxt = acc + ((t + 1) & 1) * NWV * CPY + mine;
            float cf[PPL][CR::REC], D[PPL][D2];
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
#pragma unroll
                for (int i = 0; i < R4; ++i) {
                    cf[p][4 * i] = rec[s][p][i].x; cf[p][4 * i + 1] = rec[s][p][i].y;
                    cf[p][4 * i + 2] = rec[s][p][i].z; cf[p][4 * i + 3] = rec[s][p][i].w;
                }
            }
            if (t - DEPTH >= 0) load(t - DEPTH, rec[s]);
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const int n = tid + NTS * p;
                if (n < N) {
#pragma unroll
                    for (int d = 0; d < D2; ++d) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < NWV; ++w) {
                            v += cur[w * CPY + d * N + n];
                            cur[w * CPY + d * N + n] = 0.f;
                        }
                        D[p][d] = v + cf[p][CR::kExt + d];
                    }
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        a.dP[(tb * DX + d) * N + n] = D[p][d];
                        a.dPs[(tb * DX + d) * N + n] = D[p][DX + d];
                    }
                }
            }
            if (t >= 1) {
#pragma unroll
                for (int p = 0; p < PPL; ++p) {
                    const int n = tid + NTS * p;
                    if (n < N) {
                        const int an = __float_as_int(cf[p][CR::kAnc]);
#pragma unroll
                        for (int d = 0; d < D2; ++d) {
                            float v = cf[p][CR::kB + d];
#pragma unroll
                            for (int k = 0; k < D2; ++k) v = fmaf(cf[p][d * D2 + k], D[p][k], v);
                            atomicAdd(&nxt[d * N + an], v);
                        }
                    }
                }
            }
            if (NWV > 1) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS only: see fbs_scan_kernel)
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}

// everything else filter_cov_bwd_kernel produces, from D: one workgroup per (t, sequence)
template <int DX, int DY, int H>
__global__ void __launch_bounds__(512) fcs_rows_kernel(const BwdArgs a) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    constexpr bool kRolled = true;
    constexpr int NA = 2 * DX + DY;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x, nw = NT >> 6;
    const int t = blockIdx.x, b = blockIdx.y, B = a.B, N = a.N;
    const bool valid = tid < N, first = (t == 0);
    const int n = valid ? tid : N - 1;
    const size_t tb = (size_t)t * B + b;
    float* wq1 = smem;
    float* wg = wq1 + MQ::kSize;
    float* red = wg + MG::kSize;      // [nw][NA + 4 DX]
    MQ::load(wq1, a.q1, tid, NT);
    MG::load(wg, a.g, tid, NT);
    CovStep<DX, DY> s;
    s.load(a, t, b, n, valid);
    float dpm[DX], dps[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        dpm[d] = valid ? a.dP[(tb * DX + d) * N + n] : 0.f;
        dps[d] = valid ? a.dPs[(tb * DX + d) * N + n] : 0.f;
    }
    __syncthreads();
    float sums[NA + 4 * DX];      // d sigma sums of the q1 head (DX), [f head: unused, DX], g head (DY); t = 0: d m0, d sig0, d fm0, d fsig0
#pragma unroll
    for (int i = 0; i < NA + 4 * DX; ++i) sums[i] = 0.f;
    float dx[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) dx[d] = 0.f;
    {
        float go[2 * DY], dgo[2 * DY];
        MG::template eval<kRolled>(wg, s.x, go);
#pragma unroll
        for (int k = 0; k < DY; ++k) {
            if (a.emission) {
                dgo[k] = s.dlw * (s.y[k] - emis_mean(go[k])) * emis_dmean(go[k]);
                dgo[DY + k] = 0.f;
            } else {
                const float hx = 0.1f * exp2_fast(go[DY + k] * kLog2e);
                const float isg = rcp(a.sc_g[k] + (hx + 1e-7f));
                const float z = (s.y[k] - go[k]) * isg;
                dgo[k] = s.dlw * z * isg;
                const float dsg = s.dlw * (z * z - 1.f) * isg;
                dgo[DY + k] = dsg * hx;
                sums[2 * DX + k] = dsg;
            }
            if (valid) {
                a.dG[(tb * DY + k) * N + n] = dgo[k];
                a.dGs[(tb * DY + k) * N + n] = dgo[DY + k];
            }
        }
        MG::template bwd_input<kRolled>(wg, s.x, dgo, dx);
    }
    float dfmean[DX], dfsc[DX], dc[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float ifs = rcp(s.fs[d]);
        const float z = (s.x[d] - s.fmean[d]) * ifs;
        const float tf = s.dlw * z * ifs;
        dx[d] -= tf;
        dfmean[d] = tf;
        dfsc[d] = s.dlw * (z * z - 1.f) * ifs;
        dc[d] = s.dlw * s.ic[d];
    }
    float dPo[2 * DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        sums[d] = dps[d];
        dPo[d] = dpm[d];
        dPo[DX + d] = dps[d] * (s.own[d] - a.sc_q1[d] - 1e-7f);
        if (valid) a.dPs[(tb * DX + d) * N + n] = dPo[DX + d];       // (dP already holds d mean)
    }
    MQ::template bwd_input<kRolled>(wq1, s.x, dPo, dx);
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float dmu = dx[d];
        const float dct = fmaf(dmu, s.e[d], dc[d]);
        float dmean1, ds1;
        if (a.two_q) {
            const float dA = dmu * s.c[d];
            const float dic = -(dct + dmu * s.mu[d] * s.ic[d]) * s.c[d] * s.c[d];
            const float di1 = fmaf(dA, s.mean1[d], dic), di2 = fmaf(dA, s.m2[d], dic);
            dmean1 = dA * s.i1[d];
            ds1 = -di1 * s.i1[d] * s.i1[d];
            if (valid) {
                a.dm2_rows[(tb * DX + d) * N + n] = dA * s.i2[d];
                a.ds2_rows[(tb * DX + d) * N + n] = -di2 * s.i2[d] * s.i2[d];
            }
        } else {
            dmean1 = dmu;
            ds1 = dct;
        }
        if (first) {
            sums[NA + d] = dmean1;
            sums[NA + DX + d] = ds1;
            sums[NA + 2 * DX + d] = dfmean[d];
            sums[NA + 3 * DX + d] = dfsc[d];
        }
    }
#pragma unroll
    for (int i = 0; i < NA + 4 * DX; ++i) {
        const float v = wave_sum(valid ? sums[i] : 0.f);
        if (lane == 0) red[wave * (NA + 4 * DX) + i] = v;
    }
    __syncthreads();
    if (tid < NA + 4 * DX) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += red[w * (NA + 4 * DX) + tid];
        if (tid < NA) a.scanPart[tb * NA + tid] = v;
        else if (first) {
            const int q = (tid - NA) / DX, d = (tid - NA) % DX;
            float* dst = q == 0 ? a.dm0 : q == 1 ? a.dsig0 : q == 2 ? a.dfm0 : a.dfsig0;
            dst[b * DX + d] = v;
        }
    }
}

// sacc[b][i] = sum_t part[t][b][i]: one wave per (b, i)
__global__ void __launch_bounds__(256) fcs_fold_kernel(const float* __restrict__ part, int T, int B, int NA, float* __restrict__ sacc) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < NA; i += 4) {
        float v = 0.f;
        for (int t = lane; t < T; t += 64) v += part[((size_t)t * B + b) * NA + i];
        v = wave_sum(v);
        if (lane == 0) sacc[(size_t)b * NA + i] = v;
    }
}

template <int DX, int DY, int H>
static void launch_bwd_scan(const BwdArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    const int NT = (a.N + 63) & ~63;
    const size_t ldsw = sizeof(float) * (MQ::kSize + MG::kSize);
    hipLaunchKernelGGL((fcs_coeff_kernel<DX, DY, H>), dim3(a.T, a.B), dim3(NT), ldsw, stream, a);
    const size_t ldss = sizeof(float) * 2 * 2 * DX * a.N * (a.N <= 128 ? 1 : 4);
    // record ring: DEPTH x PPL x REC (28 / 52 / 84 floats)
    constexpr int D1 = (DX == 2) ? 8 : (DX == 3) ? 4 : 2, D2 = (DX == 2) ? 4 : (DX == 3) ? 2 : 1;
    if (a.N <= 64) hipLaunchKernelGGL((fcs_scan_kernel<DX, 1, 1, D1>), dim3(a.B), dim3(64), ldss, stream, a);
    else if (a.N <= 128) hipLaunchKernelGGL((fcs_scan_kernel<DX, 1, 2, D2>), dim3(a.B), dim3(64), ldss, stream, a);
    else if (a.N <= 256) hipLaunchKernelGGL((fcs_scan_kernel<DX, 4, 1, D1>), dim3(a.B), dim3(256), ldss, stream, a);
    else hipLaunchKernelGGL((fcs_scan_kernel<DX, 4, 2, D2>), dim3(a.B), dim3(256), ldss, stream, a);
    const size_t ldsr = ldsw + sizeof(float) * ((NT / 64) * (2 * DX + DY + 4 * DX) + 16);
    hipLaunchKernelGGL((fcs_rows_kernel<DX, DY, H>), dim3(a.T, a.B), dim3(NT), ldsr, stream, a);
    hipLaunchKernelGGL(fcs_fold_kernel, dim3(a.B), dim3(256), 0, stream, a.scanPart, a.T, a.B, 2 * DX + DY, a.sacc);
}

struct BwdOut {
    float *dmu2, *dsig2, *dsc_q1, *dsc_f, *dsc_g;
};

template <int DX, int DY, int H>
static int launch_bwd(const BwdArgs& a, const BwdOut& o, hipStream_t stream) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    const int NT = (a.N + 63) & ~63;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + (a.bootstrap ? 2 : 4) * (size_t)2 * DX * NT + 16);
    clear_hip_error();
    // the affine scan wherever it applies (this path is one stream: the reverse filter is always on its critical path)
    if (g_tune_filter_bwd_scan && a.bootstrap && a.resample && a.scanRec && a.scanPart) launch_bwd_scan<DX, DY, H>(a, stream);
    else if (NT <= 256) hipLaunchKernelGGL((filter_cov_bwd_kernel<DX, DY, H, 256>), dim3(a.B), dim3(NT), lds, stream, a);
    else hipLaunchKernelGGL((filter_cov_bwd_kernel<DX, DY, H, 512>), dim3(a.B), dim3(NT), lds, stream, a);
    if (a.two_q) {
        const long long rows = (long long)a.T * a.B * DX;
        const unsigned nb = (unsigned)((rows + 3) / 4);
        hipLaunchKernelGGL(cov_row_sum, dim3(nb), dim3(256), 0, stream, a.dm2_rows, rows, a.N, o.dmu2);
        hipLaunchKernelGGL(cov_row_sum, dim3(nb), dim3(256), 0, stream, a.ds2_rows, rows, a.N, o.dsig2);
    }
    hipLaunchKernelGGL(cov_finalize, dim3(1), dim3(64), 0, stream, a.sacc, a.B, DX, DY, a.bootstrap, o.dsc_q1, o.dsc_f,
                       o.dsc_g);
    return launch_status();
}

#define PSVO_COV_DISPATCH(LAUNCH, ...)                                                         \
    do {                                                                                       \
        const int key = desc->Dx * 1000 + desc->Dy * 100 + desc->H;                            \
        switch (key) {                                                                         \
            case 2116: return LAUNCH<2, 1, 16>(__VA_ARGS__);                                   \
            case 2132: return LAUNCH<2, 1, 32>(__VA_ARGS__);                                   \
            case 2164: return LAUNCH<2, 1, 64>(__VA_ARGS__);                                   \
            case 2216: return LAUNCH<2, 2, 16>(__VA_ARGS__);                                   \
            case 2232: return LAUNCH<2, 2, 32>(__VA_ARGS__);                                   \
            case 2264: return LAUNCH<2, 2, 64>(__VA_ARGS__);                                   \
            case 3116: return LAUNCH<3, 1, 16>(__VA_ARGS__);                                   \
            case 3132: return LAUNCH<3, 1, 32>(__VA_ARGS__);                                   \
            case 3164: return LAUNCH<3, 1, 64>(__VA_ARGS__);                                   \
            case 3216: return LAUNCH<3, 2, 16>(__VA_ARGS__);                                   \
            case 3232: return LAUNCH<3, 2, 32>(__VA_ARGS__);                                   \
            case 3264: return LAUNCH<3, 2, 64>(__VA_ARGS__);                                   \
            case 4116: return LAUNCH<4, 1, 16>(__VA_ARGS__);                                   \
            case 4132: return LAUNCH<4, 1, 32>(__VA_ARGS__);                                   \
            case 4164: return LAUNCH<4, 1, 64>(__VA_ARGS__);                                   \
            case 4216: return LAUNCH<4, 2, 16>(__VA_ARGS__);                                   \
            case 4232: return LAUNCH<4, 2, 32>(__VA_ARGS__);                                   \
            case 4264: return LAUNCH<4, 2, 64>(__VA_ARGS__);                                   \
            default: return PSVO_ERR_UNSUPPORTED;                                              \
        }                                                                                      \
    } while (0)

static bool desc_ok(const psvo_desc* d) {
    return d && d->B > 0 && d->T > 0 && d->N > 0;
}

}  // namespace cov
}  // namespace psvo

extern "C" long long psvo_filter_cov_ws_floats(int B, int T, int N, int Dx, int Dy) {
    // per-sequence sums | d mu2 rows | d sig2 rows | (16-byte aligned) affine-scan records | per-step partial sums
    const long long rec = (4 * Dx * Dx + 4 * Dx + 1 + 3) / 4 * 4;      // CovRec<Dx>::REC
    return (long long)B * (2 * Dx + Dy) + 2LL * T * B * Dx * N + 4 + (long long)T * B * rec * N + (long long)T * B * (2 * Dx + Dy);
}

extern "C" int psvo_filter_forward_cov(const psvo_desc* desc, const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g,
                                       const float* sigc_q1, const float* sigc_f, const float* sigc_g, const float* mu2,
                                       const float* sig2, const float* m0, const float* sig0, const float* fm0,
                                       const float* fsig0, const float* obs, const float* eps, const float* u,
                                       const int32_t* idx_in, float* X, float* Xanc, float* Fm, float* Fs, float* P1,
                                       float* P1s, float* logW, int32_t* idx_out, float* lse, void* stream) {
    using namespace psvo;
    using namespace psvo::cov;
    if (!desc_ok(desc)) return PSVO_ERR_INVALID;
    if (desc->layers > 1 || desc->layers < 0) return PSVO_ERR_UNSUPPORTED;
    if (!q1 || !g || !sigc_q1 || !sigc_g || !m0 || !sig0 || !fm0 || !fsig0 || !obs || !eps || !X || !Xanc || !Fm || !Fs ||
        !logW || !idx_out || !lse)
        return PSVO_ERR_INVALID;
    if (desc->two_q && (!mu2 || !sig2)) return PSVO_ERR_INVALID;
    if (!desc->bootstrap && (!f || !sigc_f || !P1 || !P1s)) return PSVO_ERR_INVALID;
    if (desc->resample && !u && !idx_in) return PSVO_ERR_INVALID;
    if (desc->N > 512) return PSVO_ERR_UNSUPPORTED;
    FwdArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N;
    a.resample = desc->resample; a.two_q = desc->two_q; a.bootstrap = desc->bootstrap; a.emission = desc->emission;
    a.q1 = *q1; a.f = desc->bootstrap ? *q1 : *f; a.g = *g;
    a.sc_q1 = sigc_q1; a.sc_f = sigc_f; a.sc_g = sigc_g;
    a.mu2 = mu2; a.sig2 = sig2; a.m0 = m0; a.sig0 = sig0; a.fm0 = fm0; a.fsig0 = fsig0;
    a.obs = obs; a.eps = eps; a.u = u; a.idx_in = idx_in;
    a.X = X; a.Xanc = Xanc; a.Fm = Fm; a.Fs = Fs; a.P1 = P1; a.P1s = P1s; a.logW = logW; a.idx_out = idx_out; a.lse = lse;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PSVO_COV_DISPATCH(launch_fwd, a, s);
}

extern "C" int psvo_filter_backward_cov(
    const psvo_desc* desc, const psvo_mlp* q1, const psvo_mlp* f, const psvo_mlp* g, const float* sigc_q1,
    const float* sigc_f, const float* sigc_g, const float* mu2, const float* sig2, const float* m0, const float* sig0,
    const float* fm0, const float* fsig0, const float* obs, const float* eps, const float* X, const float* Fm,
    const float* Fs, const float* P1, const float* P1s, const float* logW, const float* lse, const int32_t* idx,
    const float* dlse, const float* dFm_ext, const float* dFs_ext, const float* dlogW_ext, float* dP, float* dPs, float* dF,
    float* dFs, float* dG, float* dGs, float* dmu2, float* dsig2, float* dm0, float* dsig0, float* dfm0, float* dfsig0,
    float* dsigc_q1, float* dsigc_f, float* dsigc_g, float* ws, void* stream) {
    using namespace psvo;
    using namespace psvo::cov;
    if (!desc_ok(desc)) return PSVO_ERR_INVALID;
    if (desc->layers > 1 || desc->layers < 0) return PSVO_ERR_UNSUPPORTED;
    if (!q1 || !g || !sigc_q1 || !sigc_g || !m0 || !sig0 || !fm0 || !fsig0 || !obs || !eps || !X || !Fm || !Fs || !logW ||
        !lse || !dP || !dPs || !dG || !dGs || !dm0 || !dsig0 || !dfm0 || !dfsig0 || !dsigc_q1 || !dsigc_g || !ws)
        return PSVO_ERR_INVALID;
    if (desc->two_q && (!mu2 || !sig2 || !dmu2 || !dsig2)) return PSVO_ERR_INVALID;
    if (!desc->bootstrap && (!f || !sigc_f || !P1 || !P1s || !dF || !dFs || !dsigc_f)) return PSVO_ERR_INVALID;
    if (desc->resample && !idx) return PSVO_ERR_INVALID;
    if (desc->N > 512) return PSVO_ERR_UNSUPPORTED;
    BwdArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N;
    a.resample = desc->resample; a.two_q = desc->two_q; a.bootstrap = desc->bootstrap; a.emission = desc->emission;
    a.q1 = *q1; a.f = desc->bootstrap ? *q1 : *f; a.g = *g;
    a.sc_q1 = sigc_q1; a.sc_f = sigc_f; a.sc_g = sigc_g;
    a.mu2 = mu2; a.sig2 = sig2; a.m0 = m0; a.sig0 = sig0; a.fm0 = fm0; a.fsig0 = fsig0; a.obs = obs; a.eps = eps;
    a.X = X; a.Fm = Fm; a.Fs = Fs; a.P1 = P1; a.P1s = P1s; a.logW = logW; a.lse = lse; a.idx = idx;
    a.dlse = dlse; a.dFm_ext = dFm_ext; a.dFs_ext = dFs_ext; a.dlogW_ext = dlogW_ext;
    a.dP = dP; a.dPs = dPs; a.dF = dF; a.dFs = dFs; a.dG = dG; a.dGs = dGs;
    a.dm0 = dm0; a.dsig0 = dsig0; a.dfm0 = dfm0; a.dfsig0 = dfsig0;
    a.sacc = ws;
    a.dm2_rows = ws + (size_t)desc->B * (2 * desc->Dx + desc->Dy);
    a.ds2_rows = a.dm2_rows + (size_t)desc->T * desc->B * desc->Dx * desc->N;
    {
        const size_t off = (size_t)desc->B * (2 * desc->Dx + desc->Dy) + 2 * (size_t)desc->T * desc->B * desc->Dx * desc->N;
        a.scanRec = ws + ((off + 3) & ~(size_t)3);
        a.scanPart = a.scanRec + (size_t)desc->T * desc->B * ((4 * desc->Dx * desc->Dx + 4 * desc->Dx + 1 + 3) / 4 * 4) * desc->N;
        if (reinterpret_cast<uintptr_t>(a.scanRec) & 15) a.scanRec = nullptr;
    }
    BwdOut o{dmu2, dsig2, dsigc_q1, dsigc_f, dsigc_g};
    hipStream_t s = static_cast<hipStream_t>(stream);
    PSVO_COV_DISPATCH(launch_bwd, a, o, s);
}
