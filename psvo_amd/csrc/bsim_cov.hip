// Backward simulation with proposal (PSVO.backward_simulation_w_proposal, reference src/SMC/PSVO.py:69-203) and its reverse
// pass with STATE-DEPENDENT diagonal scales: FLAGS.output_cov and FLAGS.diag_cov (src/runner_flag.py:67-70), every MLP with
// a second head (src/transformation/MLP.py:40-46,58-61), every scale sigma_con + 0.1 (exp(head) + 1e-6)
// (src/distribution/mvn.py:66-71).  Companion of filter_cov.hip; same mathematics as bsim_fwd.hip / bsim_bwd_impl.h with
// every scale a per-row value:
//   * the transition tile of forward step t-1 carries a scale per forward particle j: slot j = (F'_jd, R'_jd, W'_j) with
//     R'_jd = sqrt(log2(e) / 2) / sigma_f(X_j)_d, F'_jd = Fm_jd R'_jd and W'_j = (logW_j - lse) log2(e) - sum_d log2 sigma_jd,
//     so that a pair costs one fma per dimension more than with a constant scale:  v = W'_j - sum_d (x_d R'_jd - F'_jd)^2
//     (log2 domain; the (M, N, N, B) tile of PSVO.py:128-133 is never materialised);
//   * MLP_f / MLP_g are evaluated per sub-particle with both heads, MLP_q1inv per chain with both heads, and the product of
//     the two backward proposals on scales (SVO.py:186-197 as called from PSVO.py:120-122) is formed per chain and step;
//   * the hoisted distributions arrive as mean and scale per row: bmu2 / bsig2 (T,B,Dx), minit / sinit, imean / isig (B,Dx).
// Work decomposition: workgroup = 256 / M chains of ONE sequence, lane = (chain, sub-particle m), persistent over the steps;
// every lane walks ALL forward particles of the staged tile (broadcast LDS reads).  This is the plain mapping of bsim_fwd.hip
// without its quad blocking and half-split chains: the path is a non-default flag combination of the reference, built for
// parity first.  In the reverse pass the per-j sums (d Fm, d Fs, d logW of the forward filter: 2 Dx + 1 values per forward
// particle) are reduced over the lanes of a wave with DPP adds, over the waves of the workgroup with LDS float atomics and over
// the workgroups of a sequence with global float atomics (the summation order is not fixed: gradients are reproducible to
// rounding, not bit for bit).
// PSVOwR (src/SMC/PSVOwR.py:65-198) runs the same kernels in a with-resampling mode (template flag WR), one launch per step.
#include "common.h"

namespace psvo {
namespace covb {

__device__ __forceinline__ float head_sigma(float con, float raw) {
    return con + 0.1f * (exp2_fast(raw * kLog2e) + 1e-6f);
}
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float ln(float x) { return kLn2 * log2_fast(x); }

constexpr int kMaxStage = 4;      // forward-tile entries a thread stages per step (N <= 1024 with 256 threads)

struct FwdArgs {
    int B, T, N;
    int emission;
    psvo_mlp f, g, q1inv;
    const float *Fm, *Fs, *logW, *lse;
    const float *sc_f, *sc_g, *sc_q1inv;
    const float *bmu2, *bsig2, *minit, *sinit, *imean, *isig;
    const float *obs, *eps_b, *u_b;
    const int32_t* sel_in;
    float *bwX, *flp, *glp, *Omega;
    int32_t* sel_out;
    float* score;
    float *lam_all, *om_all, *mu1_all, *s1_all;
    // PSVOwR (WR = true): one launch per time step, t_hi == t_lo (or -1: only the cross-chain draw of step 0)
    int t_hi, t_lo;
    const float* u_r;          // (T,B,N) uniforms of the cross-chain draw
    const int32_t* anc_in;     // (T,B,N) teacher-forced ancestors or null
    float *bwXanc, *bwW, *omsel;   // (T,B,Dx,N), (T,B,N), (T,B,N): resampled chain states, per-step log-weights, drawn omegas
    int32_t* anc_out;          // (T,B,N)
};

// forward tile of one filter step in LDS: slot j = [F'_0.. | R'_0.. | W' | pad], TS floats
template <int DX>
struct Tile {
    static constexpr int TS = 2 * DX + 2;
    // raw = Fm[DX] | Fs[DX] | logW
    __device__ __forceinline__ static void put(float* buf, int j, int N, const float (&raw)[2 * DX + 1], float lse) {
        const float kap = sqrtf(0.5f * kLog2e);
        float W = j < N ? (raw[2 * DX] - lse) * kLog2e : -__builtin_huge_valf();
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float s = raw[DX + d];
            const float R = kap * rcp(s);
            buf[j * TS + d] = raw[d] * R;
            buf[j * TS + DX + d] = R;
            W -= log2_fast(s);
        }
        buf[j * TS + 2 * DX] = W;
    }
};

// WR = false: PSVO, persistent over t = T-1 .. 0.  WR = true: PSVOwR (src/SMC/PSVOwR.py:65-198) -- the chains of a sequence
// are resampled ACROSS the particle axis after every step (logits = the drawn sub-particles' normalised log-weights,
// PSVOwR.py:103,145,185), which couples all workgroups of a sequence once per step: the loop then runs one launch per time
// step (host loop in psvo_bsimwr_forward_cov, captured into the step's hipGraph like any other launch), and the cross-chain
// draw of step t+1 is made at the START of the launch of step t, by every workgroup for its own chains, from the N logits
// the previous launch left in HBM -- the kernel boundary is the exchange.
template <int DX, int DY, int H, int M, bool WR>
__global__ void __launch_bounds__(256) bsim_cov_fwd_kernel(const FwdArgs a) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    using TL = Tile<DX>;
    constexpr int TS = TL::TS;
    constexpr bool kRolled = true;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int NTB = blockDim.x;
    const int B = a.B, T = a.T, N = a.N;
    const int NP = (N + 7) & ~7;                 // tile padded (W' = -inf) to whole blocks of 8
    const int b = blockIdx.y;
    const int cpb = NTB / M;
    const int cl = tid / M, m = tid % M;
    const bool lead = (m == 0);
    const int n_raw = blockIdx.x * cpb + cl;
    const bool valid = n_raw < N;
    const int n = valid ? n_raw : N - 1;
    const int gbase = lane - m;

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;   // [2][NP][TS]
    float* cdf = tile + 2 * NP * TS; // WR: [1024 + 32] CDF of the cross-chain draw + scratch
    const int t_hi = WR ? a.t_hi : T - 1, t_lo = WR ? a.t_lo : 0;

    MQ::load(wf, a.f, tid, NTB);
    MG::load(wg, a.g, tid, NTB);
    MQ::load(wqi, a.q1inv, tid, NTB);

    float cf[DX], cq[DX], cg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        cf[d] = a.sc_f[d];
        cq[d] = a.sc_q1inv[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) cg[e] = a.sc_g[e];
    float xp[DX];   // x_{t+1} of this chain (same in all M lanes)
#pragma unroll
    for (int d = 0; d < DX; ++d) xp[d] = 0.f;
    if constexpr (WR) {
        // ---- cross-chain draw of step tr = t_hi + 1 (PSVOwR.py:103,145,185): ancestors of this workgroup's chains ----------
        const int tr = t_hi + 1;
        if (tr <= T - 1) {
            const size_t rb = (size_t)tr * B + b;
            float* red = cdf + 1024;
            int anc;
            if (a.anc_in) {
                anc = a.anc_in[rb * N + n];
            } else {
                // inclusive CDF of exp(omega_sel - max) over the N chains: thread i owns entries [4 i, 4 i + 4)
                float v[4], mx = -__builtin_huge_valf();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = 4 * tid + k;
                    v[k] = j < N ? a.omsel[rb * N + j] : -__builtin_huge_valf();
                    mx = fmaxf(mx, v[k]);
                }
                mx = wave_max(mx);
                if (lane == 0) red[tid >> 6] = mx;
                __syncthreads();
                mx = red[0];
                for (int w = 1; w < (NTB >> 6); ++w) mx = fmaxf(mx, red[w]);
                __syncthreads();
                float run = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    run += (4 * tid + k < N) ? exp2_fast((v[k] - mx) * kLog2e) : 0.f;
                    v[k] = run;
                }
                const float inc = wave_incl_scan(run, lane);
                if (lane == 63) red[tid >> 6] = inc;
                __syncthreads();
                float off = inc - run, total = 0.f;
                for (int w = 0; w < (NTB >> 6); ++w) {
                    if (w < (tid >> 6)) off += red[w];
                    total += red[w];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * tid + k < 1024) cdf[4 * tid + k] = v[k] + off;
                __syncthreads();
                const float target = a.u_r[rb * N + n] * total;
                int pos = 0;     // count of cdf entries <= target (the filter's search, SVO.py:266-300)
                for (int st = 1 << (31 - __clz(N)); st > 0; st >>= 1) {
                    const int q = pos + st;
                    if (q <= N && cdf[q - 1] <= target) pos = q;
                }
                anc = min(pos, N - 1);
            }
#pragma unroll
            for (int d = 0; d < DX; ++d) xp[d] = a.bwX[(rb * DX + d) * N + anc];
            if (valid && lead) {
                a.anc_out[rb * N + n] = anc;
#pragma unroll
                for (int d = 0; d < DX; ++d) a.bwXanc[(rb * DX + d) * N + n] = xp[d];
            }
        }
        if (t_hi < 0) return;      // (the launch behind step 0: only the draw)
    }
    float mi[DX], si[DX], im[DX], is[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        mi[d] = a.minit[b * DX + d];
        si[d] = a.sinit[b * DX + d];
        im[d] = a.imean[b * DX + d];
        is[d] = a.isig[b * DX + d];
    }
    const float logM = logf((float)M);
    const float ninf = -__builtin_huge_valf();

    // ---- forward-tile staging: global -> registers at the top of a step, registers -> LDS at its end ------------------------
    float st[kMaxStage][2 * DX + 1], st_l = 0.f;
    auto stage_load = [&](int tt) {
        const size_t tb = (size_t)tt * B + b;
        st_l = a.lse[tb];
#pragma unroll
        for (int r = 0; r < kMaxStage; ++r) {
            const int j = tid + r * NTB;
            if (j < NP) {
                const int jc = j < N ? j : N - 1;
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    st[r][d] = a.Fm[(tb * DX + d) * N + jc];
                    st[r][DX + d] = a.Fs[(tb * DX + d) * N + jc];
                }
                st[r][2 * DX] = a.logW[tb * N + jc];
            }
        }
    };
    auto stage_store = [&](float* buf) {
#pragma unroll
        for (int r = 0; r < kMaxStage; ++r) {
            const int j = tid + r * NTB;
            if (j < NP) TL::put(buf, j, N, st[r], st_l);
        }
    };
    if (t_hi >= 1) {
        stage_load(t_hi - 1);
        stage_store(tile);
    }

    float eps_c[DX], bmu_c[DX], bs_c[DX], obs_c[DY], u_c = 0.f;
    int sel_c = 0;
    auto load_inputs = [&](int t, float (&e)[DX], float (&bm)[DX], float (&bs)[DX], float (&o)[DY], float& uu, int& ss) {
        const size_t tb = (size_t)t * B + b;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            e[d] = a.eps_b[((tb * DX + d) * N + n) * M + m];
            bm[d] = a.bmu2[tb * DX + d];
            bs[d] = a.bsig2[tb * DX + d];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) o[k] = a.obs[tb * DY + k];
        if (a.sel_in) ss = a.sel_in[tb * N + n];
        else uu = a.u_b[tb * N + n];
    };
    load_inputs(t_hi, eps_c, bmu_c, bs_c, obs_c, u_c, sel_c);
    __syncthreads();

    float score = 0.f;

    for (int t = t_hi; t >= t_lo; --t) {
        const size_t tb = (size_t)t * B + b;
        const float* cur = tile + ((t_hi - t) & 1) * NP * TS;
        float* nxt = tile + ((t_hi - t + 1) & 1) * NP * TS;
        const bool last = (t == T - 1), tzero = (t == 0);
        const bool more = (t > t_lo);             // another step follows in this launch

        float eps_n[DX], bmu_n[DX], bs_n[DX], obs_n[DY], u_n = 0.f;
        int sel_n = 0;
        if (more) load_inputs(t - 1, eps_n, bmu_n, bs_n, obs_n, u_n, sel_n);
        if (more && t >= 2) stage_load(t - 2);

        // ---- proposal ---------------------------------------------------------------------------------------------------
        float x[DX], mu[DX], ic[DX];
        float kq = -DX * kHalfLog2Pi;
        if (last) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                mu[d] = mi[d];
                ic[d] = rcp(si[d]);
                x[d] = fmaf(si[d], eps_c[d], mu[d]);
                kq -= ln(si[d]);
            }
            if (a.mu1_all && valid && lead) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    a.mu1_all[(tb * DX + d) * N + n] = 0.f;
                    a.s1_all[(tb * DX + d) * N + n] = 1.f;
                }
            }
        } else {
            float qo[2 * DX];
            MQ::template eval<kRolled>(wqi, xp, qo);
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float s1 = head_sigma(cq[d], qo[DX + d]);
                if (a.mu1_all && valid && lead) {
                    a.mu1_all[(tb * DX + d) * N + n] = qo[d];
                    a.s1_all[(tb * DX + d) * N + n] = s1;
                }
                const float i1 = rcp(s1), i2 = rcp(bs_c[d]);
                ic[d] = i1 + i2;
                const float c = rcp(ic[d]);
                mu[d] = c * fmaf(i1, qo[d], i2 * bmu_c[d]);
                x[d] = fmaf(c, eps_c[d], mu[d]);
                kq -= ln(c);
            }
        }
        const float q_lp = diag_lp<DX>(x, mu, ic, kq);

        // ---- f(x_{t+1} | x~), g(y_t | x~): both heads of each MLP at the sub-particle ----------------------------------------
        float phi = 0.f;
        if (!last) {
            float fo[2 * DX], ifs[DX];
            MQ::template eval<kRolled>(wf, x, fo);
            float kf = -DX * kHalfLog2Pi;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float s = head_sigma(cf[d], fo[DX + d]);
                ifs[d] = rcp(s);
                kf -= ln(s);
            }
            float fmx[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) fmx[d] = fo[d];
            phi = diag_lp<DX>(xp, fmx, ifs, kf);
        }
        float go[2 * DY], gm[DY], isg[DY];
        MG::template eval<kRolled>(wg, x, go);
        float kg = -DY * kHalfLog2Pi;
#pragma unroll
        for (int k = 0; k < DY; ++k) {
            if (a.emission) {
                gm[k] = emis_mean(go[k]);
                isg[k] = 1.f;
            } else {
                gm[k] = go[k];
                const float s = head_sigma(cg[k], go[DY + k]);
                isg[k] = rcp(s);
                kg -= ln(s);
            }
        }
        const float g_lp = diag_lp<DY>(obs_c, gm, isg, kg);

        // ---- filter term: logsumexp_j( log f(x~ | X_{t-1}[j]) + W^_{t-1}[j] ), per-j scales ----------------------------------
        float lam;
        if (!tzero) {
            float mx = ninf, sm = 0.f;
            for (int j0 = 0; j0 < NP; j0 += 8) {
                float v[8], bm = ninf;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float* p = cur + (j0 + c) * TS;
                    float acc = p[2 * DX];
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        const float u = fmaf(x[d], p[DX + d], -p[d]);
                        acc = fmaf(-u, u, acc);
                    }
                    v[c] = acc;
                    bm = fmaxf(bm, acc);
                }
                const float nm = fmaxf(mx, bm);
                const float base = (nm == ninf) ? 0.f : nm;
                sm *= exp2_fast(mx - base);
#pragma unroll
                for (int c = 0; c < 8; ++c) sm += exp2_fast(v[c] - base);
                mx = nm;
            }
            const float lam2 = mx + log2_fast(sm);
            lam = fmaf(kLn2, lam2, -DX * kHalfLog2Pi);
            if (a.lam_all && valid) a.lam_all[(tb * N + n) * M + m] = lam2;
        } else {
            float iis[DX];
            float ki = -DX * kHalfLog2Pi;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                iis[d] = rcp(is[d]);
                ki -= ln(is[d]);
            }
            lam = diag_lp<DX>(x, im, iis, ki);   // t = 0: q0 / f density at mu_0 (PSVO.py:169-175)
        }

        // ---- omega, normalise over the M sub-particles, draw one -----------------------------------------------------------
        const float om_raw = lam + phi + g_lp - q_lp;
        const float omx = group_max<M>(om_raw);
        const float pw = exp2_fast((om_raw - omx) * kLog2e);
        const float cdfv = group_incl_scan<M>(pw, m);
        const float total = __shfl(cdfv, gbase + M - 1);
        const float omega = om_raw - fmaf(kLn2, log2_fast(total), omx);
        if (a.om_all && valid) a.om_all[(tb * N + n) * M + m] = omega;
        int sel;
        if (a.sel_in) {
            sel = sel_c;
        } else {
            const unsigned long long bal = __ballot(cdfv <= u_c * total);
            const unsigned long long mask = (M == 64) ? ~0ull : (((1ull << M) - 1ull) << gbase);
            sel = min((int)__popcll(bal & mask), M - 1);
        }
        const int src = gbase + sel;
        float xs[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) xs[d] = __shfl(x[d], src);
        const float om_s = __shfl(omega, src), phi_s = __shfl(phi, src), g_s = __shfl(g_lp, src), q_s = __shfl(q_lp, src),
                    lam_s = __shfl(lam, src);
        const float Om = om_s + q_s + logM;
        if constexpr (WR) {
            // per-step log-weight of the chain (PSVOwR.py:97-101, 134-141, 176-181): Lam (t >= 1) or the prior term (t = 0)
            // + g - (q + omega + log M) of the drawn sub-particle
            (void)phi_s;
            if (valid && lead) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.bwX[(tb * DX + d) * N + n] = xs[d];
                a.bwW[tb * N + n] = lam_s + g_s - Om;
                a.omsel[tb * N + n] = om_s;
                a.sel_out[tb * N + n] = sel;
            }
        } else {
            if (valid && lead) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.bwX[(tb * DX + d) * N + n] = xs[d];
                a.glp[tb * N + n] = g_s;
                a.Omega[tb * N + n] = Om;
                a.sel_out[tb * N + n] = sel;
                if (!last) a.flp[(tb + B) * N + n] = phi_s;     // f_log_probs[t+1]
                if (tzero) a.flp[(size_t)b * N + n] = lam_s;    // f_log_probs[0] = f_init
            }
            score += g_s - Om + (last ? 0.f : phi_s) + (tzero ? lam_s : 0.f);
        }

#pragma unroll
        for (int d = 0; d < DX; ++d) {
            xp[d] = xs[d];
            eps_c[d] = eps_n[d];
            bmu_c[d] = bmu_n[d];
            bs_c[d] = bs_n[d];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) obs_c[k] = obs_n[k];
        u_c = u_n;
        sel_c = sel_n;
        if (more && t >= 2) stage_store(nxt);
        __syncthreads();
    }
    if constexpr (!WR) {
        if (valid && lead) a.score[(size_t)b * N + n] = score;
    }
}

// lseW[t][b] = logsumexp_n bwW[t][b][:]  (PSVOwR.py:52-63; one wave per (t, b))
__global__ void __launch_bounds__(256) wr_lse_kernel(const float* __restrict__ W, long long rows, int N, float* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float mx = -__builtin_huge_valf();
    for (int l = lane; l < N; l += 64) mx = fmaxf(mx, W[r * N + l]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int l = lane; l < N; l += 64) s += exp2_fast((W[r * N + l] - mx) * kLog2e);
    s = wave_sum(s);
    if (lane == 0) out[r] = fmaf(kLn2, log2_fast(s), mx);
}

// ---------------------------------------------------------------------------------------------------------------------------
// reverse pass: t = 0 .. T-1, carrying d loss / d bwX[t] of each chain.  With aw = d loss / d score of the chain,
// p_m = exp(omega_m) the normalised sub-particle weights and s the drawn sub-particle (bsim_bwd_impl.h has the derivation):
//   d g_m = aw p_m,  d phi_m = aw p_m (t < T-1),  d q_m = -aw p_m,  d lam_m = aw (p_m - [m == s]) (t >= 1),  aw p_m (t = 0),
//   d x_m += [m == s] d bwX[t].
// ---------------------------------------------------------------------------------------------------------------------------
struct BwdArgs {
    int B, T, N;
    int emission;
    psvo_mlp f, g, q1inv;
    const float *Fm, *Fs, *logW, *lse;
    const float *sc_f, *sc_g, *sc_q1inv;
    const float *bmu2, *bsig2, *minit, *sinit, *imean, *isig;
    const float *obs, *eps_b, *bwX;
    const int32_t* sel;
    const float *lam_all, *om_all, *mu1_all, *s1_all, *dscore;
    // rows for psvo_mlp_wgrad: w.r.t. the mean head and w.r.t. the raw scale head of each MLP evaluation
    float *xt, *dFt, *dFts, *dGt, *dGts;    // (T,B,D,N,M)
    float *dmu1, *dmu1s;                    // (T,B,Dx,N)  (MLP_q1inv at bwX[t+1])
    // accumulated with float atomics (zero-filled by the caller)
    float *dFm, *dFs, *dlogW, *dlse;        // (T,B,Dx,N) x 2, (T,B,N), (T,B): d loss / d the forward filter's outputs
    float *dbmu2, *dbsig2;                  // (T,B,Dx)
    float *dminit, *dsinit, *dimean, *disig;   // (B,Dx)
    float *dsc_f, *dsc_g, *dsc_q1inv;       // (Dx), (Dy), (Dx)
    // PSVOwR (WR = true): one launch per time step t_lo == t_hi
    int t_lo, t_hi;
    const float *bwXanc, *bwW, *lseW, *dlseW;   // (T,B,Dx,N), (T,B,N), (T,B), (T,B)
    const int32_t* anc;                     // (T,B,N)
    float* dXs;                             // (T,B,Dx,N) d loss / d bwX (zero-filled by the caller; float atomics)
};

// WR = false: PSVO, persistent over t = 0 .. T-1 with d loss / d bwX[t] carried in registers.  WR = true: PSVOwR, one launch
// per time step; the loss is sum_t logsumexp_n W_t[n], so with aw = d loss / d W_t[n] = dlseW_t softmax_n(W_t):
//   W = logsumexp_m(omega_raw) - phi_s - log M   =>   d lam_m = d g_m = -d q_m = aw p_m,   d phi_m = aw (p_m - [m == s]),
// and the chain's x_{t+1} is the RESAMPLED state bwXanc[t+1][n] = bwX[t+1][anc], so its gradient is scatter-added into the
// ancestor chain's d bwX[t+1] (HBM float atomics; the next launch reads it).
template <int DX, int DY, int H, int M, bool WR>
__global__ void __launch_bounds__(256) bsim_cov_bwd_kernel(const BwdArgs a) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    using TL = Tile<DX>;
    constexpr int TS = TL::TS;
    constexpr int NJ = 2 * DX + 1;            // per-j sums: d F (DX), d sigma (DX), d W^
    constexpr int NI = DX + 2;                // floats per item of the pair phase
    constexpr bool kRolled = true;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int NTB = blockDim.x;
    const int B = a.B, T = a.T, N = a.N;
    const int NP = (N + 15) & ~15;               // tile padded (W' = -inf: zero weight) to whole tiles of 16
    const int b = blockIdx.y;
    const int cpb = NTB / M;
    const int cl = tid / M, m = tid % M;
    const int n_raw = blockIdx.x * cpb + cl;
    const bool valid = n_raw < N;
    const int n = valid ? n_raw : N - 1;
    const bool lead = valid && (m == 0);

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;            // [NP][TS]   tile of forward step t - 1
    float* jacc = tile + NP * TS;             // [NJ][NP]   per-j sums of the step over the workgroup
    float* xch = jacc + NJ * NP;              // [4][64][NI] items of each wave: x~ (DX), lam2, d lam
    float* cacc = xch + 4 * 64 * NI;             // [4 * DX]   per-step sums over the workgroup's chains: d bmu2, d bsig2 | t = T-1: d minit, d sinit | t = 0: d imean, d isig

    MQ::load(wf, a.f, tid, NTB);
    MG::load(wg, a.g, tid, NTB);
    MQ::load(wqi, a.q1inv, tid, NTB);

    float cf[DX], cq[DX], cg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        cf[d] = a.sc_f[d];
        cq[d] = a.sc_q1inv[d];
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) cg[e] = a.sc_g[e];
    float mi[DX], si[DX], im[DX], is[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        mi[d] = a.minit[b * DX + d];
        si[d] = a.sinit[b * DX + d];
        im[d] = a.imean[b * DX + d];
        is[d] = a.isig[b * DX + d];
    }
    float aw = (!WR && valid) ? a.dscore[(size_t)b * N + n] : 0.f;
    const float kap = sqrtf(0.5f * kLog2e);
    const int t_lo = WR ? a.t_lo : 0, t_hi = WR ? a.t_hi : T - 1;

    float acc_f[DX], acc_q[DX], acc_g[DY];    // sums of d sigma over this lane's rows of the f / q1inv / g heads
#pragma unroll
    for (int d = 0; d < DX; ++d) acc_f[d] = acc_q[d] = 0.f;
#pragma unroll
    for (int e = 0; e < DY; ++e) acc_g[e] = 0.f;
    float dX[DX];   // d loss / d bwX[t] of this chain (same value in its M lanes)
#pragma unroll
    for (int d = 0; d < DX; ++d) dX[d] = 0.f;
    __syncthreads();

    for (int t = t_lo; t <= t_hi; ++t) {
        const size_t tb = (size_t)t * B + b;
        const bool last = (t == T - 1), tzero = (t == 0);
        if constexpr (WR) {
            aw = valid ? a.dlseW[tb] * exp2_fast((a.bwW[tb * N + n] - a.lseW[tb]) * kLog2e) : 0.f;
#pragma unroll
            for (int d = 0; d < DX; ++d) dX[d] = a.dXs[(tb * DX + d) * N + n];
        }

        // ---- stage the tile of forward step t - 1, zero the step's accumulators ---------------------------------------------
        if (!tzero) {
            const size_t pb = tb - B;
            const float l = a.lse[pb];
            for (int j = tid; j < NP; j += NTB) {
                const int jc = j < N ? j : N - 1;
                float raw[2 * DX + 1];
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    raw[d] = a.Fm[(pb * DX + d) * N + jc];
                    raw[DX + d] = a.Fs[(pb * DX + d) * N + jc];
                }
                raw[2 * DX] = a.logW[pb * N + jc];
                TL::put(tile, j, N, raw, l);
            }
            for (int i = tid; i < NJ * NP; i += NTB) jacc[i] = 0.f;
        }
        if (tid < 4 * DX) cacc[tid] = 0.f;
        __syncthreads();

        // ---- recompute the step's forward quantities ------------------------------------------------------------------------------
        float eps[DX], xp[DX], y[DY];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            eps[d] = a.eps_b[((tb * DX + d) * N + n) * M + m];
            xp[d] = last ? 0.f : (WR ? a.bwXanc : a.bwX)[((tb + B) * DX + d) * N + n];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = a.obs[tb * DY + k];
        const int sel = a.sel[tb * N + n];
        const float p_m = exp2_fast(a.om_all[(tb * N + n) * M + m] * kLog2e);
        const float dsel = (m == sel) ? 1.f : 0.f;

        float x[DX], mu[DX], c[DX], i1[DX], i2[DX], m1[DX], s1[DX], bm[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            if (last) {
                mu[d] = mi[d];
                c[d] = si[d];
                i1[d] = i2[d] = m1[d] = bm[d] = 0.f;
                s1[d] = 1.f;
            } else {
                m1[d] = a.mu1_all[(tb * DX + d) * N + n];
                s1[d] = a.s1_all[(tb * DX + d) * N + n];
                bm[d] = a.bmu2[tb * DX + d];
                i1[d] = rcp(s1[d]);
                i2[d] = rcp(a.bsig2[tb * DX + d]);
                c[d] = rcp(i1[d] + i2[d]);
                mu[d] = c[d] * fmaf(i1[d], m1[d], i2[d] * bm[d]);
            }
            x[d] = fmaf(c[d], eps[d], mu[d]);
        }
        const float dgp = aw * p_m;                               // d g_m = -d q_m (PSVO: = d phi_m)
        const float dlam = (WR || tzero) ? dgp : aw * (p_m - dsel);
        const float dphi = WR ? aw * (p_m - dsel) : dgp;

        float dx[DX], dxp[DX];      // gradients w.r.t. this sub-particle and (its share of) w.r.t. the chain's x_{t+1}
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            dx[d] = dsel * dX[d];
            dxp[d] = 0.f;
        }
        if (valid) {
#pragma unroll
            for (int d = 0; d < DX; ++d) a.xt[((tb * DX + d) * N + n) * M + m] = x[d];
        }
        // ---- g(y_t | x~) -------------------------------------------------------------------------------------------------------
        {
            float go[2 * DY], dgo[2 * DY];
            MG::template eval<kRolled>(wg, x, go);
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                if (a.emission) {
                    dgo[k] = dgp * (y[k] - emis_mean(go[k])) * emis_dmean(go[k]);
                    dgo[DY + k] = 0.f;
                } else {
                    const float hx = 0.1f * exp2_fast(go[DY + k] * kLog2e);
                    const float isg = rcp(cg[k] + (hx + 1e-7f));
                    const float z = (y[k] - go[k]) * isg;
                    dgo[k] = dgp * z * isg;
                    const float ds = dgp * (z * z - 1.f) * isg;
                    dgo[DY + k] = ds * hx;
                    acc_g[k] += valid ? ds : 0.f;
                }
                if (valid) {
                    a.dGt[((tb * DY + k) * N + n) * M + m] = dgo[k];
                    a.dGts[((tb * DY + k) * N + n) * M + m] = dgo[DY + k];
                }
            }
            MG::template bwd_input<kRolled>(wg, x, dgo, dx);
        }
        // ---- f(x_{t+1} | x~) ---------------------------------------------------------------------------------------------------
        if (!last) {
            float fo[2 * DX], dfo[2 * DX];
            MQ::template eval<kRolled>(wf, x, fo);
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float hx = 0.1f * exp2_fast(fo[DX + d] * kLog2e);
                const float ifs = rcp(cf[d] + (hx + 1e-7f));
                const float z = (xp[d] - fo[d]) * ifs;
                dfo[d] = dphi * z * ifs;
                dxp[d] -= dfo[d];
                const float ds = dphi * (z * z - 1.f) * ifs;
                dfo[DX + d] = ds * hx;
                acc_f[d] += valid ? ds : 0.f;
            }
            if (valid) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    a.dFt[((tb * DX + d) * N + n) * M + m] = dfo[d];
                    a.dFts[((tb * DX + d) * N + n) * M + m] = dfo[DX + d];
                }
            }
            MQ::template bwd_input<kRolled>(wf, x, dfo, dx);
        } else if (valid) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                a.dFt[((tb * DX + d) * N + n) * M + m] = 0.f;
                a.dFts[((tb * DX + d) * N + n) * M + m] = 0.f;
            }
        }
        // ---- filter term ---------------------------------------------------------------------------------------------------------
        if (!tzero) {
            // Pair phase with j ON THE LANES of a 16-lane row: lane = (g, j16), row g = lane >> 4 works on the 16 items
            // (chain, m) that the lanes of its own row own, against forward particle j = 16 jt + j16 of tile jt.  The items
            // travel through wave-private LDS into registers once per step; per-ITEM sums (d x~) accumulate in registers over
            // the tiles and are reduced over the row's 16 lanes once per step (four DPP rotations per value); per-j sums
            // accumulate in registers over the row's items and are reduced over the four rows once per tile (two swap-adds).
            // (The first version kept lane = item and reduced 2 Dx + 1 per-j values over the wave for EVERY j: 15.8 ms at C*.)
            const float lam2 = a.lam_all[(tb * N + n) * M + m];
            float* const xw = xch + (tid >> 6) * 64 * NI;
            {
#pragma unroll
                for (int d = 0; d < DX; ++d) xw[lane * NI + d] = x[d];
                xw[lane * NI + DX] = lam2;
                xw[lane * NI + DX + 1] = valid ? dlam : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            const int g16 = lane & ~15, j16 = lane & 15;
            float ix[16][DX], il[16], idl[16], ixa[16][DX];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    ix[i][d] = xw[(g16 + i) * NI + d];
                    ixa[i][d] = 0.f;
                }
                il[i] = xw[(g16 + i) * NI + DX];
                idl[i] = xw[(g16 + i) * NI + DX + 1];
            }
            for (int jt = 0; jt < NP; jt += 16) {
                const int j = jt + j16;
                const float* p = tile + j * TS;
                float Fp[DX], Rp[DX], Rk[DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    Fp[d] = p[d];
                    Rp[d] = p[DX + d];
                    Rk[d] = Rp[d] * (1.f / kap);          // 1 / sigma_jd
                }
                const float Wp = p[2 * DX];
                float ja[NJ];
#pragma unroll
                for (int q = 0; q < NJ; ++q) ja[q] = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float acc = Wp, u[DX];
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        u[d] = fmaf(ix[i][d], Rp[d], -Fp[d]);       // (x - F_j) kappa / sigma_j
                        acc = fmaf(-u[d], u[d], acc);
                    }
                    const float cj = idl[i] * exp2_fast(acc - il[i]);   // d lam_k * p_kj
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        const float g1 = cj * (u[d] * Rp[d]) * (2.f * kLn2);          // c (x - F) / sigma^2
                        ixa[i][d] += g1;
                        ja[d] += g1;                                                  // d F_jd
                        ja[DX + d] += cj * fmaf(u[d] * u[d], 2.f * kLn2, -1.f) * Rk[d];   // d sigma_jd = c (z^2 - 1) / sigma
                    }
                    ja[2 * DX] += cj;                                                 // d W^_j
                }
#pragma unroll
                for (int q = 0; q < NJ; ++q) {
                    float v = ja[q];
                    v += xor_lane<16>(v);
                    v += xor_lane<32>(v);
                    if (lane < 16) atomicAdd(&jacc[q * NP + j], v);
                }
            }
            // item sums over the row's 16 lanes; item i of the row is owned by the row's lane i, which keeps the sum
            float mine[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) mine[d] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    float v = ixa[i][d];
                    v += dpp_mov<0x128, 0xF, 0xF, true>(v, v);   // row_ror:8
                    v += dpp_mov<0x124, 0xF, 0xF, true>(v, v);   // row_ror:4
                    v += dpp_mov<0x122, 0xF, 0xF, true>(v, v);   // row_ror:2
                    v += dpp_mov<0x121, 0xF, 0xF, true>(v, v);   // row_ror:1
                    mine[d] = (j16 == i) ? v : mine[d];
                }
            }
#pragma unroll
            for (int d = 0; d < DX; ++d) dx[d] -= mine[d];
        } else {
            float dmi = 0.f, dsi = 0.f;
#pragma unroll
            for (int d = 0; d < DX; ++d) {     // lam = log N(x; imean, isig): t = 0 prior term
                const float iis = rcp(is[d]);
                const float z = (x[d] - im[d]) * iis;
                const float g1 = (valid ? dlam : 0.f) * z * iis;
                dx[d] -= g1;
                dmi = group_sum<M>(g1);
                dsi = group_sum<M>((valid ? dlam : 0.f) * (z * z - 1.f) * iis);
                if (lead) {
                    atomicAdd(&cacc[2 * DX + d], dmi);
                    atomicAdd(&cacc[3 * DX + d], dsi);
                }
            }
        }
        // ---- x~ = mu + c eps;  -q_lp contributes + sum log c -------------------------------------------------------------------------------
        float dmu[DX], dc[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            dmu[d] = group_sum<M>(dx[d]);
            dc[d] = group_sum<M>(fmaf(dx[d], eps[d], dgp * rcp(c[d])));
        }
        float dq[2 * DX];
        if (!last) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float ic = i1[d] + i2[d];
                const float dA = dmu[d] * c[d];
                const float dic = -(dc[d] + dmu[d] * mu[d] * ic) * c[d] * c[d];
                const float di1 = fmaf(dA, m1[d], dic), di2 = fmaf(dA, bm[d], dic);
                const float ds1 = -di1 * i1[d] * i1[d];
                dq[d] = dA * i1[d];                                        // d MLP_q1inv mean head
                dq[DX + d] = ds1 * (s1[d] - cq[d] - 1e-7f);              // d raw scale head (0.1 exp(raw) = s1 - con - 1e-7)
                if (lead) {
                    acc_q[d] += ds1;
                    a.dmu1[(tb * DX + d) * N + n] = dq[d];
                    a.dmu1s[(tb * DX + d) * N + n] = dq[DX + d];
                    atomicAdd(&cacc[d], dA * i2[d]);
                    atomicAdd(&cacc[DX + d], -di2 * i2[d] * i2[d]);
                }
            }
            float dxq[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) dxq[d] = 0.f;
            MQ::template bwd_input<kRolled>(wqi, xp, dq, dxq);      // (the same in the chain's M lanes)
#pragma unroll
            for (int d = 0; d < DX; ++d) dX[d] = group_sum<M>(dxp[d]) + dxq[d];
            if constexpr (WR) {     // x_{t+1} = bwX[t+1][anc]: the gradient goes to the ancestor chain
                if (lead) {
                    const int an = a.anc[(tb + B) * N + n];
#pragma unroll
                    for (int d = 0; d < DX; ++d) atomicAdd(&a.dXs[((tb + B) * DX + d) * N + an], dX[d]);
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                if (lead) {
                    a.dmu1[(tb * DX + d) * N + n] = 0.f;
                    a.dmu1s[(tb * DX + d) * N + n] = 0.f;
                    atomicAdd(&cacc[d], dmu[d]);
                    atomicAdd(&cacc[DX + d], dc[d]);
                }
            }
        }
        __syncthreads();
        // ---- flush the step's workgroup sums ---------------------------------------------------------------------------------------
        if (!tzero) {
            const size_t pb = tb - B;
            for (int i = tid; i < NJ * N; i += NTB) {
                const int q = i / N, j = i - q * N;
                float v = jacc[q * NP + j];
                if (q < DX) atomicAdd(&a.dFm[(pb * DX + q) * N + j], v);
                else if (q < 2 * DX) atomicAdd(&a.dFs[(pb * DX + (q - DX)) * N + j], v);
                else atomicAdd(&a.dlogW[pb * N + j], v);
            }
            // d lse of forward step t - 1: -(sum_j d W^_j)  (the tile holds logW - lse)
            float part = 0.f;
            for (int j = tid; j < N; j += NTB) part += jacc[2 * DX * NP + j];
            part = wave_sum(part);
            if (lane == 0) atomicAdd(&a.dlse[pb], -part);
        }
        if (tid < DX) {
            if (last) {
                atomicAdd(&a.dminit[b * DX + tid], cacc[tid]);
                atomicAdd(&a.dsinit[b * DX + tid], cacc[DX + tid]);
            } else {
                atomicAdd(&a.dbmu2[tb * DX + tid], cacc[tid]);
                atomicAdd(&a.dbsig2[tb * DX + tid], cacc[DX + tid]);
            }
            if (tzero) {
                atomicAdd(&a.dimean[b * DX + tid], cacc[2 * DX + tid]);
                atomicAdd(&a.disig[b * DX + tid], cacc[3 * DX + tid]);
            }
        }
        __syncthreads();
    }
    // ---- d sigma_con of the three heads ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float sf = wave_sum(acc_f[d]), sq = wave_sum(acc_q[d]);
        if (lane == 0) {
            atomicAdd(&a.dsc_f[d], sf);
            atomicAdd(&a.dsc_q1inv[d], sq);
        }
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) {
        const float sg = wave_sum(acc_g[e]);
        if (lane == 0) atomicAdd(&a.dsc_g[e], sg);
    }
}

template <int DX, int DY, int H, int M>
static int launch_fwd(const FwdArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    const int NP = (a.N + 7) & ~7;
    int NTB = ((a.N * M + 63) / 64) * 64;
    if (NTB > 256) NTB = 256;
    const int cpb = NTB / M;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * Tile<DX>::TS);
    dim3 grid((a.N + cpb - 1) / cpb, a.B);
    clear_hip_error();
    hipLaunchKernelGGL((bsim_cov_fwd_kernel<DX, DY, H, M, false>), grid, dim3(NTB), lds, stream, a);
    return launch_status();
}

// PSVOwR: one launch per time step T-1 .. 0, one more for the cross-chain draw of step 0, then the per-step logsumexp
template <int DX, int DY, int H, int M>
static int launch_fwd_wr(const FwdArgs& a0, float* lseW, hipStream_t stream) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    const int NP = (a0.N + 7) & ~7;
    int NTB = ((a0.N * M + 63) / 64) * 64;
    if (NTB > 256) NTB = 256;
    const int cpb = NTB / M;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * Tile<DX>::TS + 1024 + 32);
    dim3 grid((a0.N + cpb - 1) / cpb, a0.B);
    clear_hip_error();
    FwdArgs a = a0;
    for (int t = a0.T - 1; t >= -1; --t) {
        a.t_hi = a.t_lo = t;
        hipLaunchKernelGGL((bsim_cov_fwd_kernel<DX, DY, H, M, true>), grid, dim3(NTB), lds, stream, a);
    }
    const long long rows = (long long)a0.T * a0.B;
    hipLaunchKernelGGL(wr_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, a0.bwW, rows, a0.N, lseW);
    return launch_status();
}

template <int DX, int DY, int H, int M>
static int launch_bwd(const BwdArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, 2 * DX, 1>;
    using MG = MlpLds<DX, H, 2 * DY, 1>;
    const int NP = (a.N + 15) & ~15;
    int NTB = ((a.N * M + 63) / 64) * 64;
    if (NTB > 256) NTB = 256;
    const int cpb = NTB / M;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + (size_t)NP * (Tile<DX>::TS + 2 * DX + 1) +
                                        4 * 64 * (DX + 2) + 4 * DX + 16);
    dim3 grid((a.N + cpb - 1) / cpb, a.B);
    clear_hip_error();
    if (a.t_lo < 0) {       // PSVO: persistent over all steps
        hipLaunchKernelGGL((bsim_cov_bwd_kernel<DX, DY, H, M, false>), grid, dim3(NTB), lds, stream, a);
    } else {                // PSVOwR: one launch per time step, 0 .. T-1
        BwdArgs w = a;
        for (int t = 0; t < a.T; ++t) {
            w.t_lo = w.t_hi = t;
            hipLaunchKernelGGL((bsim_cov_bwd_kernel<DX, DY, H, M, true>), grid, dim3(NTB), lds, stream, w);
        }
    }
    return launch_status();
}

#define PSVO_COVB_M(LAUNCH, DX_, DY_, H_, ...)                                      \
    switch (desc->M) {                                                              \
        case 4: return LAUNCH<DX_, DY_, H_, 4>(__VA_ARGS__);                        \
        case 8: return LAUNCH<DX_, DY_, H_, 8>(__VA_ARGS__);                        \
        case 16: return LAUNCH<DX_, DY_, H_, 16>(__VA_ARGS__);                      \
        case 32: return LAUNCH<DX_, DY_, H_, 32>(__VA_ARGS__);                      \
        default: return PSVO_ERR_UNSUPPORTED;                                       \
    }
#define PSVO_COVB_DISPATCH(LAUNCH, ...)                                             \
    do {                                                                            \
        const int key = desc->Dx * 1000 + desc->Dy * 100 + desc->H;                 \
        switch (key) {                                                              \
            case 2116: PSVO_COVB_M(LAUNCH, 2, 1, 16, __VA_ARGS__)                   \
            case 2132: PSVO_COVB_M(LAUNCH, 2, 1, 32, __VA_ARGS__)                   \
            case 2164: PSVO_COVB_M(LAUNCH, 2, 1, 64, __VA_ARGS__)                   \
            case 2216: PSVO_COVB_M(LAUNCH, 2, 2, 16, __VA_ARGS__)                   \
            case 2232: PSVO_COVB_M(LAUNCH, 2, 2, 32, __VA_ARGS__)                   \
            case 2264: PSVO_COVB_M(LAUNCH, 2, 2, 64, __VA_ARGS__)                   \
            case 3116: PSVO_COVB_M(LAUNCH, 3, 1, 16, __VA_ARGS__)                   \
            case 3132: PSVO_COVB_M(LAUNCH, 3, 1, 32, __VA_ARGS__)                   \
            case 3164: PSVO_COVB_M(LAUNCH, 3, 1, 64, __VA_ARGS__)                   \
            case 3216: PSVO_COVB_M(LAUNCH, 3, 2, 16, __VA_ARGS__)                   \
            case 3232: PSVO_COVB_M(LAUNCH, 3, 2, 32, __VA_ARGS__)                   \
            case 3264: PSVO_COVB_M(LAUNCH, 3, 2, 64, __VA_ARGS__)                   \
            case 4116: PSVO_COVB_M(LAUNCH, 4, 1, 16, __VA_ARGS__)                   \
            case 4132: PSVO_COVB_M(LAUNCH, 4, 1, 32, __VA_ARGS__)                   \
            case 4164: PSVO_COVB_M(LAUNCH, 4, 1, 64, __VA_ARGS__)                   \
            case 4216: PSVO_COVB_M(LAUNCH, 4, 2, 16, __VA_ARGS__)                   \
            case 4232: PSVO_COVB_M(LAUNCH, 4, 2, 32, __VA_ARGS__)                   \
            case 4264: PSVO_COVB_M(LAUNCH, 4, 2, 64, __VA_ARGS__)                   \
            default: return PSVO_ERR_UNSUPPORTED;                                   \
        }                                                                           \
    } while (0)

static bool desc_ok(const psvo_desc* d) { return d && d->B > 0 && d->T >= 2 && d->N > 0 && d->M > 0; }

}  // namespace covb
}  // namespace psvo

extern "C" int psvo_bsim_forward_cov(const psvo_desc* desc, const float* Fm, const float* Fs, const float* logW,
                                     const float* lse, const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                                     const float* sigc_f, const float* sigc_g, const float* sigc_q1inv, const float* bmu2,
                                     const float* bsig2, const float* minit, const float* sinit, const float* imean,
                                     const float* isig, const float* obs, const float* eps_b, const float* u_b,
                                     const int32_t* sel_in, float* bwX, float* flp, float* glp, float* Omega,
                                     int32_t* sel_out, float* score, float* lam_all, float* om_all, float* mu1_all,
                                     float* s1_all, void* stream) {
    using namespace psvo;
    using namespace psvo::covb;
    if (!desc) return PSVO_ERR_INVALID;
    if (desc->layers > 1 || desc->layers < 0) return PSVO_ERR_UNSUPPORTED;
    if (!desc_ok(desc)) return PSVO_ERR_INVALID;
    if (!Fm || !Fs || !logW || !lse || !f || !g || !q1_inv || !sigc_f || !sigc_g || !sigc_q1inv || !bmu2 || !bsig2 || !minit ||
        !sinit || !imean || !isig || !obs || !eps_b || !bwX || !flp || !glp || !Omega || !sel_out || !score)
        return PSVO_ERR_INVALID;
    if (!u_b && !sel_in) return PSVO_ERR_INVALID;
    if ((mu1_all != nullptr) != (s1_all != nullptr)) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    FwdArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.Fs = Fs; a.logW = logW; a.lse = lse;
    a.sc_f = sigc_f; a.sc_g = sigc_g; a.sc_q1inv = sigc_q1inv;
    a.bmu2 = bmu2; a.bsig2 = bsig2; a.minit = minit; a.sinit = sinit; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.u_b = u_b; a.sel_in = sel_in;
    a.bwX = bwX; a.flp = flp; a.glp = glp; a.Omega = Omega; a.sel_out = sel_out; a.score = score;
    a.lam_all = lam_all; a.om_all = om_all; a.mu1_all = mu1_all; a.s1_all = s1_all;
    a.t_hi = desc->T - 1; a.t_lo = 0; a.u_r = nullptr; a.anc_in = nullptr;
    a.bwXanc = a.bwW = a.omsel = nullptr; a.anc_out = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PSVO_COVB_DISPATCH(launch_fwd, a, s);
}

extern "C" int psvo_bsim_backward_cov(
    const psvo_desc* desc, const float* Fm, const float* Fs, const float* logW, const float* lse, const psvo_mlp* f,
    const psvo_mlp* g, const psvo_mlp* q1_inv, const float* sigc_f, const float* sigc_g, const float* sigc_q1inv,
    const float* bmu2, const float* bsig2, const float* minit, const float* sinit, const float* imean, const float* isig,
    const float* obs, const float* eps_b, const float* bwX, const int32_t* sel, const float* lam_all, const float* om_all,
    const float* mu1_all, const float* s1_all, const float* dscore, float* xt, float* dFt, float* dFts, float* dGt,
    float* dGts, float* dmu1, float* dmu1s, float* dFm, float* dFs, float* dlogW, float* dlse, float* dbmu2, float* dbsig2,
    float* dminit, float* dsinit, float* dimean, float* disig, float* dsigc_f, float* dsigc_g, float* dsigc_q1inv,
    void* stream) {
    using namespace psvo;
    using namespace psvo::covb;
    if (!desc) return PSVO_ERR_INVALID;
    if (desc->layers > 1 || desc->layers < 0) return PSVO_ERR_UNSUPPORTED;
    if (!desc_ok(desc)) return PSVO_ERR_INVALID;
    if (!Fm || !Fs || !logW || !lse || !f || !g || !q1_inv || !sigc_f || !sigc_g || !sigc_q1inv || !bmu2 || !bsig2 || !minit ||
        !sinit || !imean || !isig || !obs || !eps_b || !bwX || !sel || !lam_all || !om_all || !mu1_all || !s1_all || !dscore ||
        !xt || !dFt || !dFts || !dGt || !dGts || !dmu1 || !dmu1s || !dFm || !dFs || !dlogW || !dlse || !dbmu2 || !dbsig2 ||
        !dminit || !dsinit || !dimean || !disig || !dsigc_f || !dsigc_g || !dsigc_q1inv)
        return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    BwdArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.Fs = Fs; a.logW = logW; a.lse = lse;
    a.sc_f = sigc_f; a.sc_g = sigc_g; a.sc_q1inv = sigc_q1inv;
    a.bmu2 = bmu2; a.bsig2 = bsig2; a.minit = minit; a.sinit = sinit; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.bwX = bwX; a.sel = sel;
    a.lam_all = lam_all; a.om_all = om_all; a.mu1_all = mu1_all; a.s1_all = s1_all; a.dscore = dscore;
    a.xt = xt; a.dFt = dFt; a.dFts = dFts; a.dGt = dGt; a.dGts = dGts; a.dmu1 = dmu1; a.dmu1s = dmu1s;
    a.dFm = dFm; a.dFs = dFs; a.dlogW = dlogW; a.dlse = dlse; a.dbmu2 = dbmu2; a.dbsig2 = dbsig2;
    a.dminit = dminit; a.dsinit = dsinit; a.dimean = dimean; a.disig = disig;
    a.dsc_f = dsigc_f; a.dsc_g = dsigc_g; a.dsc_q1inv = dsigc_q1inv;
    a.t_lo = a.t_hi = -1;       // persistent over all steps
    a.bwXanc = a.bwW = a.lseW = a.dlseW = nullptr; a.anc = nullptr; a.dXs = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PSVO_COVB_DISPATCH(launch_bwd, a, s);
}

/* PSVOwR with state-dependent scales: see include/psvo_hip.h */
extern "C" int psvo_bsimwr_forward_cov(
    const psvo_desc* desc, const float* Fm, const float* Fs, const float* logW, const float* lse, const psvo_mlp* f,
    const psvo_mlp* g, const psvo_mlp* q1_inv, const float* sigc_f, const float* sigc_g, const float* sigc_q1inv,
    const float* bmu2, const float* bsig2, const float* minit, const float* sinit, const float* imean, const float* isig,
    const float* obs, const float* eps_b, const float* u_b, const float* u_r, const int32_t* sel_in, const int32_t* anc_in,
    float* bwX, float* bwXanc, float* bwW, float* lseW, int32_t* sel_out, int32_t* anc_out, float* omsel, float* lam_all,
    float* om_all, float* mu1_all, float* s1_all, void* stream) {
    using namespace psvo;
    using namespace psvo::covb;
    if (!desc) return PSVO_ERR_INVALID;
    if (desc->layers > 1 || desc->layers < 0) return PSVO_ERR_UNSUPPORTED;
    if (!desc_ok(desc)) return PSVO_ERR_INVALID;
    if (!Fm || !Fs || !logW || !lse || !f || !g || !q1_inv || !sigc_f || !sigc_g || !sigc_q1inv || !bmu2 || !bsig2 || !minit ||
        !sinit || !imean || !isig || !obs || !eps_b || !bwX || !bwXanc || !bwW || !lseW || !sel_out || !anc_out || !omsel)
        return PSVO_ERR_INVALID;
    if ((!u_b && !sel_in) || (!u_r && !anc_in)) return PSVO_ERR_INVALID;
    if ((mu1_all != nullptr) != (s1_all != nullptr)) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    FwdArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.Fs = Fs; a.logW = logW; a.lse = lse;
    a.sc_f = sigc_f; a.sc_g = sigc_g; a.sc_q1inv = sigc_q1inv;
    a.bmu2 = bmu2; a.bsig2 = bsig2; a.minit = minit; a.sinit = sinit; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.u_b = u_b; a.sel_in = sel_in;
    a.bwX = bwX; a.flp = nullptr; a.glp = nullptr; a.Omega = nullptr; a.sel_out = sel_out; a.score = nullptr;
    a.lam_all = lam_all; a.om_all = om_all; a.mu1_all = mu1_all; a.s1_all = s1_all;
    a.t_hi = a.t_lo = 0; a.u_r = u_r; a.anc_in = anc_in;
    a.bwXanc = bwXanc; a.bwW = bwW; a.omsel = omsel; a.anc_out = anc_out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PSVO_COVB_DISPATCH(launch_fwd_wr, a, lseW, s);
}

extern "C" int psvo_bsimwr_backward_cov(
    const psvo_desc* desc, const float* Fm, const float* Fs, const float* logW, const float* lse, const psvo_mlp* f,
    const psvo_mlp* g, const psvo_mlp* q1_inv, const float* sigc_f, const float* sigc_g, const float* sigc_q1inv,
    const float* bmu2, const float* bsig2, const float* minit, const float* sinit, const float* imean, const float* isig,
    const float* obs, const float* eps_b, const float* bwXanc, const float* bwW, const float* lseW, const int32_t* sel,
    const int32_t* anc, const float* lam_all, const float* om_all, const float* mu1_all, const float* s1_all,
    const float* dlseW, float* xt, float* dFt, float* dFts, float* dGt, float* dGts, float* dmu1, float* dmu1s, float* dFm,
    float* dFs, float* dlogW, float* dlse, float* dbmu2, float* dbsig2, float* dminit, float* dsinit, float* dimean,
    float* disig, float* dsigc_f, float* dsigc_g, float* dsigc_q1inv, float* dXs, void* stream) {
    using namespace psvo;
    using namespace psvo::covb;
    if (!desc) return PSVO_ERR_INVALID;
    if (desc->layers > 1 || desc->layers < 0) return PSVO_ERR_UNSUPPORTED;
    if (!desc_ok(desc)) return PSVO_ERR_INVALID;
    if (!Fm || !Fs || !logW || !lse || !f || !g || !q1_inv || !sigc_f || !sigc_g || !sigc_q1inv || !bmu2 || !bsig2 || !minit ||
        !sinit || !imean || !isig || !obs || !eps_b || !bwXanc || !bwW || !lseW || !sel || !anc || !lam_all || !om_all ||
        !mu1_all || !s1_all || !dlseW || !xt || !dFt || !dFts || !dGt || !dGts || !dmu1 || !dmu1s || !dFm || !dFs || !dlogW ||
        !dlse || !dbmu2 || !dbsig2 || !dminit || !dsinit || !dimean || !disig || !dsigc_f || !dsigc_g || !dsigc_q1inv || !dXs)
        return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    BwdArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.Fs = Fs; a.logW = logW; a.lse = lse;
    a.sc_f = sigc_f; a.sc_g = sigc_g; a.sc_q1inv = sigc_q1inv;
    a.bmu2 = bmu2; a.bsig2 = bsig2; a.minit = minit; a.sinit = sinit; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.bwX = nullptr; a.sel = sel;
    a.lam_all = lam_all; a.om_all = om_all; a.mu1_all = mu1_all; a.s1_all = s1_all; a.dscore = nullptr;
    a.xt = xt; a.dFt = dFt; a.dFts = dFts; a.dGt = dGt; a.dGts = dGts; a.dmu1 = dmu1; a.dmu1s = dmu1s;
    a.dFm = dFm; a.dFs = dFs; a.dlogW = dlogW; a.dlse = dlse; a.dbmu2 = dbmu2; a.dbsig2 = dbsig2;
    a.dminit = dminit; a.dsinit = dsinit; a.dimean = dimean; a.disig = disig;
    a.dsc_f = dsigc_f; a.dsc_g = dsigc_g; a.dsc_q1inv = dsigc_q1inv;
    a.t_lo = a.t_hi = 0;        // (>= 0: one launch per time step)
    a.bwXanc = bwXanc; a.bwW = bwW; a.lseW = lseW; a.dlseW = dlseW; a.anc = anc; a.dXs = dXs;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PSVO_COVB_DISPATCH(launch_bwd, a, s);
}
