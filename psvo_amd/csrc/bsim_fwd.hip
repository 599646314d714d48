// Backward simulation with proposal (PSVO): restates PSVO.backward_simulation_w_proposal
// (reference src/SMC/PSVO.py:69-203) -- see include/psvo_hip.h for the contract.
//
// Work decomposition (MI355X).  The B*N backward chains never interact (PSVO.py:116-151); each
// chain step draws M sub-particles and needs, for each, logsumexp_j over ALL N forward particles
// of the transition log-density (the reference's (M, N, N, B) tile, PSVO.py:128-133).
//   * workgroup = 256 lanes = 256/(M*HS) chains of ONE sequence b, persistent over t = T-1 .. 0;
//   * lane = (chain, [half,] sub-particle m): the lane owns proposal m (sampling, MLP_f, MLP_g); with HS = 2 the two
//     halves of a chain split the forward particles and the hidden units of every MLP (two waves per SIMD at C*);
//   * the pair loop is register-blocked over the lane's QUAD: the four lanes of a quad hold four
//     consecutive m; each lane walks the forward particles j = q, q+4, ... (q = quad lane) and
//     evaluates all four m of its quad against each j, so one 16-byte LDS broadcast read feeds
//     four pairs.  Partial (max, sum) pairs are merged across the quad with two xor-shuffles.
//   * the forward tile of step t-1 -- F'_j = MLP_f(X_{t-1}[j]) * r and W'_j = normalised log
//     weight, both pre-scaled into the log2 domain -- is staged in LDS (N * 16 B; 2 KB at N=128),
//     double-buffered, prefetched from HBM one step ahead.  The tile is never materialised.
//   * pair arithmetic: v = W'_j - sum_d (x'_d - F'_jd)^2  (log2 domain) in packed f32 (two sub-particles per
//     instruction); log-sum-exp in blocks of 16 tile entries: all v of a block and its maximum first, then one
//     rescale of the running sums and the block's exponentials (the tile is padded with W' = -inf to whole blocks).
#include "common.h"

namespace psvo {
inline namespace PSVO_LNS {   // l1 / l2: hidden layers of the per-particle MLPs (common.h)
PSVO_TIMERS_DEFINE(bsim_fwd)


struct BsimArgs {
    int B, T, N;
    int emission;
    psvo_mlp f, g, q1inv;
    const float *X, *Fm, *logW, *lse;
    const float *sig_f, *sig_g, *sig_q1inv, *sig_bq2;
    const float *bmu2, *minit, *sig_init, *imean, *isig;
    const float *obs, *eps_b, *u_b;
    const int32_t* sel_in;
    float *bwX, *flp, *glp, *Omega;
    int32_t* sel_out;
    float* score;
    float *lam2_all, *om_all, *mu1_all;  // optional saves for psvo_bsim_backward (may be null)
};

template <int DX>
struct TileSlot {
    static constexpr int kFloats = (DX <= 3) ? 4 : 8;
};

__device__ __forceinline__ float quad_bcast(float v, int lane, int i) { return __shfl(v, (lane & ~3) | i); }

// merge two online-lse states kept in the log2 domain
__device__ __forceinline__ void lse2_merge(float& m, float& s, float m2, float s2) {
    const float nm = fmaxf(m, m2);
    // guard -inf - -inf
    const float a = (m == nm) ? 1.f : exp2_fast(m - nm);
    const float b = (m2 == nm) ? 1.f : exp2_fast(m2 - nm);
    s = s * a + s2 * b;
    m = nm;
}

// HS = 1: lane = (chain, m).  HS = 2: lane = (chain, half, m) -- the two halves of a chain split the
// walk over the forward particles and the hidden units of every MLP evaluation, so a chain step is
// spread over 2*M lanes.  At C* (65536 (chain, m) pairs = one wave per SIMD, where a lone wave issues
// one VALU op per 4 cycles) this doubles the resident waves per SIMD.
template <int DX, int DY, int H, int M, int HS>
__global__ void __launch_bounds__(256) bsim_fwd_kernel(const BsimArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    constexpr int PS = TileSlot<DX>::kFloats;
    constexpr int kMaxStage = 4;
    constexpr bool kRolled = (2 * MQ::kSize + MG::kSize) > 330;
    constexpr bool kPeel = (DX <= 3);   // (the Dx = 4 kernel measured slower peeled: C5 14.4 -> 17.6 ms)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int NTB = blockDim.x;
    const int B = a.B, T = a.T, N = a.N;
    // forward tile padded (W' = -inf) so that every half-slice of a quad lane is a whole number of 4-entry blocks
    const int NP = ((N + 16 * HS - 1) / (16 * HS)) * (16 * HS);
    const int b = blockIdx.y;
    constexpr int G = M * HS;  // lanes per chain
    const int cpb = NTB / G;
    const int cl = tid / G, hpart = (tid % G) / M, m = tid % M, q = m & 3;
    const bool lead = (m == 0) && (hpart == 0);
    const int n_raw = blockIdx.x * cpb + cl;
    const bool valid = n_raw < N;
    const int n = valid ? n_raw : N - 1;
    const int gbase = lane - m;  // first lane of this chain's M-lane group inside the wave

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;  // 2 buffers of NP * PS floats

    MQ::load(wf, a.f, tid, NTB);
    MG::load(wg, a.g, tid, NTB);
    MQ::load(wqi, a.q1inv, tid, NTB);

    // ---- constants --------------------------------------------------------------------------
    float isf[DX], rp[DX], isg[DY];
    float kf = -DX * kHalfLog2Pi, kg = -DY * kHalfLog2Pi;
    const float rscale = sqrtf(0.5f * kLog2e);
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float s = a.sig_f[d];
        isf[d] = 1.f / s;
        rp[d] = isf[d] * rscale;  // (x' - F')^2 summed == 0.5*log2e*sum(((x-F)/s)^2)
        kf -= logf(s);
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) {
        const float s = a.sig_g[e];
        isg[e] = 1.f / s;
        kg -= logf(s);
    }
    // PoG(q1_inv, BSim_q2) on scales (SVO.py:186-197 as called from PSVO.py:120-122)
    float pc[DX], pic[DX], pi1[DX], pi2[DX];
    float kq = -DX * kHalfLog2Pi;
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        pi1[d] = 1.f / a.sig_q1inv[d];
        pi2[d] = 1.f / a.sig_bq2[d];
        pic[d] = pi1[d] + pi2[d];
        pc[d] = 1.f / pic[d];
        kq -= logf(pc[d]);
    }
    // q_init (t = T-1) and the t = 0 "filter" term
    float s_init[DX], is_init[DX], i_isig[DX], im[DX], mi[DX];
    float kinit = -DX * kHalfLog2Pi, kiota = -DX * kHalfLog2Pi;
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        s_init[d] = a.sig_init[d];
        is_init[d] = 1.f / s_init[d];
        kinit -= logf(s_init[d]);
        i_isig[d] = 1.f / a.isig[d];
        kiota -= logf(a.isig[d]);
        im[d] = a.imean[b * DX + d];
        mi[d] = a.minit[b * DX + d];
    }
    const float logM = logf((float)M);
    const float ninf = -__builtin_huge_valf();

    // ---- forward-tile staging -----------------------------------------------------------------
    // The global loads of step t-2's tile are issued at the top of step t and only *consumed* (scaled and
    // written to LDS) at its end: nothing between depends on them, so their latency hides behind the step.
    float st[kMaxStage][DX + 1], st_l = 0.f;
    auto stage_load = [&](int tt) {  // global -> registers (raw), forward step tt
        const size_t tb = (size_t)tt * B + b;
        st_l = a.lse[tb];
#pragma unroll
        for (int r = 0; r < kMaxStage; ++r) {
            const int j = tid + r * NTB;
            if (j < NP) {
                const int jc = j < N ? j : N - 1;
#pragma unroll
                for (int d = 0; d < DX; ++d) st[r][d] = a.Fm[(tb * DX + d) * N + jc];
                st[r][DX] = a.logW[tb * N + jc];
            }
        }
    };
    auto stage_store = [&](float* buf) {  // registers -> LDS: F' = Fm * rho, W' = (logW - lse) * log2(e)
#pragma unroll
        for (int r = 0; r < kMaxStage; ++r) {
            const int j = tid + r * NTB;
            if (j < NP) {
                float F[DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) F[d] = st[r][d] * rp[d];
                const float W = j < N ? (st[r][DX] - st_l) * kLog2e : ninf;
                if constexpr (DX <= 3) {
                    float4 v;
                    v.x = F[0];
                    v.y = DX > 1 ? F[DX > 1 ? 1 : 0] : 0.f;
                    v.z = DX > 2 ? F[DX > 2 ? 2 : 0] : 0.f;
                    v.w = W;
                    *reinterpret_cast<float4*>(buf + j * PS) = v;
                } else {
                    *reinterpret_cast<float4*>(buf + j * PS) = make_float4(F[0], F[1], F[2], F[3]);
                    *reinterpret_cast<float4*>(buf + j * PS + 4) = make_float4(W, 0.f, 0.f, 0.f);
                }
            }
        }
    };

    if (T >= 2) {
        stage_load(T - 2);
        stage_store(tile);
    }

    // ---- per-step inputs, prefetched one step ahead ----------------------------------------------
    float eps_c[DX], bmu_c[DX], obs_c[DY], u_c = 0.f;
    int sel_c = 0;
    auto load_inputs = [&](int t, float (&e)[DX], float (&bm)[DX], float (&o)[DY], float& uu, int& ss) {
        const size_t tb = (size_t)t * B + b;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            e[d] = a.eps_b[((tb * DX + d) * N + n) * M + m];
            bm[d] = a.bmu2[tb * DX + d];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) o[k] = a.obs[tb * DY + k];
        if (a.sel_in) ss = a.sel_in[tb * N + n];
        else uu = a.u_b[tb * N + n];
    };
    load_inputs(T - 1, eps_c, bmu_c, obs_c, u_c, sel_c);
    __syncthreads();

    float xp[DX];  // x_{t+1} of this chain (same in all M lanes)
#pragma unroll
    for (int d = 0; d < DX; ++d) xp[d] = 0.f;
    float score = 0.f;

    SEC_INIT(bsim_fwd)
    // the steps t = T-1 (initial proposal, no successor) and t = 0 (prior term instead of the tile pass) are peeled:
    // the T-2 steps in between carry none of their branches
    auto step = [&](auto last_tag, auto zero_tag, const int t) {
        SEC(0);   // loop overhead / barrier tail
        const size_t tb = (size_t)t * B + b;
        const float* cur = tile + ((T - 1 - t) & 1) * NP * PS;
        float* nxt = tile + ((T - t) & 1) * NP * PS;
        const bool last = last_tag;                 // (compile-time constants in the peeled instantiations)
        const bool tzero = zero_tag;

        float eps_n[DX], bmu_n[DX], obs_n[DY], u_n = 0.f;
        int sel_n = 0;
        if (!tzero) load_inputs(t - 1, eps_n, bmu_n, obs_n, u_n, sel_n);
        if (t >= 2) stage_load(t - 2);

        SEC(1);   // issue of the prefetch loads
        // ---- proposal ---------------------------------------------------------------------------
        float x[DX], q_lp;
        if (last) {
            float mu[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                mu[d] = mi[d];
                x[d] = fmaf(s_init[d], eps_c[d], mu[d]);
            }
            q_lp = diag_lp<DX>(x, mu, is_init, kinit);
            if (a.mu1_all && valid && lead) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.mu1_all[(tb * DX + d) * N + n] = 0.f;
            }
        } else {
            float m1[DX], mu[DX];
            {   // x_{t+1} is the same in all G lanes of the chain: spread MLP_q1inv's hidden units over kQS of them
                constexpr int kQS = (H / 4 < G) ? H / 4 : G;
                MQ::template eval_part<kQS>(wqi, (tid % G) & (kQS - 1), xp, m1);
#pragma unroll
                for (int d = 0; d < DX; ++d) m1[d] = group_sum<kQS>(m1[d]);
            }
            if (a.mu1_all && valid && lead) {
#pragma unroll
                for (int d = 0; d < DX; ++d) a.mu1_all[(tb * DX + d) * N + n] = m1[d];
            }
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                mu[d] = pc[d] * fmaf(pi1[d], m1[d], pi2[d] * bmu_c[d]);
                x[d] = fmaf(pc[d], eps_c[d], mu[d]);
            }
            q_lp = diag_lp<DX>(x, mu, pic, kq);
        }

        SEC(2);   // MLP_q1inv + proposal
        // ---- f(x_{t+1} | x~), g(y_t | x~) ------------------------------------------------------------
        float phi = 0.f;
        if (!last) {
            float fmx[DX];
            if constexpr (HS == 1) {
                MQ::template eval<kRolled>(wf, x, fmx);
            } else {
                MQ::template eval_part<HS, ilog2(M)>(wf, hpart, x, fmx);
#pragma unroll
                for (int d = 0; d < DX; ++d) fmx[d] += xor_lane<M>(fmx[d]);
            }
            phi = diag_lp<DX>(xp, fmx, isf, kf);
        }
        float gm[DY];
        if constexpr (HS == 1) {
            MG::template eval<kRolled>(wg, x, gm);
        } else {
            MG::template eval_part<HS, ilog2(M)>(wg, hpart, x, gm);
#pragma unroll
            for (int k = 0; k < DY; ++k) gm[k] += xor_lane<M>(gm[k]);
        }
        if (a.emission) {
#pragma unroll
            for (int k = 0; k < DY; ++k) gm[k] = emis_mean(gm[k]);
        }
        const float g_lp = diag_lp<DY>(obs_c, gm, isg, kg);

        SEC(3);   // MLP_f, MLP_g, densities
        // ---- filter term: logsumexp_j( log f(x~ | X_{t-1}[j]) + W^_{t-1}[j] ) -----------------------
        float lam;
        if (!tzero) {
            float xq[4][DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                float t4[4];
                quad_bcast4(x[d] * rp[d], t4);
#pragma unroll
                for (int i = 0; i < 4; ++i) xq[i][d] = t4[i];
            }
            // The lane walks its quad-slice of the forward tile in blocks of JB entries (4 pairs per entry:
            // the quad's four sub-particles).  Phase A forms every v of the block with packed f32 math
            // (v_pk_add_f32 / v_pk_fma_f32 / v_pk_max_f32: sub-particles (0,1) and (2,3) share an instruction)
            // and the block maximum; phase B rescales the running sums ONCE per block and adds the block's
            // exponentials.  At C* one block covers the lane's whole slice, so no rescaling remains.
            f2 xa[DX], xb[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                xa[d] = f2{xq[0][d], xq[1][d]};
                xb[d] = f2{xq[2][d], xq[3][d]};
            }
            f2 mxa = f2{ninf, ninf}, mxb = mxa, sma = f2{0.f, 0.f}, smb = sma;
            const int nqh = (NP >> 2) / HS;       // forward-tile entries per quad lane and half (multiple of 4)
            const int j0 = hpart * nqh;
            auto block = [&](auto jb_tag, int k0) {
                constexpr int JB = decltype(jb_tag)::value;
                f2 va[JB], vb[JB];
                f2 bma = f2{ninf, ninf}, bmb = bma;
#pragma unroll
                for (int c = 0; c < JB; ++c) {
                    const float* p = cur + ((j0 + k0 + c) * 4 + q) * PS;
                    float F[DX], W;
                    if constexpr (DX <= 3) {
                        const float4 sl = *reinterpret_cast<const float4*>(p);
                        F[0] = sl.x;
                        if (DX > 1) F[DX > 1 ? 1 : 0] = sl.y;
                        if (DX > 2) F[DX > 2 ? 2 : 0] = sl.z;
                        W = sl.w;
                    } else {
                        const float4 sl = *reinterpret_cast<const float4*>(p);
                        F[0] = sl.x; F[1] = sl.y; F[2] = sl.z; F[3] = sl.w;
                        W = p[4];
                    }
                    f2 la = f2{W, W}, lb = la;
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        const f2 Fd = f2{F[d], F[d]};
                        const f2 ua = xa[d] - Fd, ub = xb[d] - Fd;
                        la = pk_fma(-ua, ua, la);
                        lb = pk_fma(-ub, ub, lb);
                    }
                    va[c] = la;
                    vb[c] = lb;
                    bma = pk_max(bma, la);
                    bmb = pk_max(bmb, lb);
                }
                const f2 nma = pk_max(mxa, bma), nmb = pk_max(mxb, bmb);
                // nm == -inf only if every term so far is -inf: keep s = 0 without NaNs
                const f2 ba = f2{nma.x == ninf ? 0.f : nma.x, nma.y == ninf ? 0.f : nma.y};
                const f2 bb = f2{nmb.x == ninf ? 0.f : nmb.x, nmb.y == ninf ? 0.f : nmb.y};
                const f2 ra = mxa - ba, rb = mxb - bb;
                sma = sma * f2{exp2_fast(ra.x), exp2_fast(ra.y)};
                smb = smb * f2{exp2_fast(rb.x), exp2_fast(rb.y)};
#pragma unroll
                for (int c = 0; c < JB; ++c) {
                    const f2 da = va[c] - ba, db = vb[c] - bb;
                    sma += f2{exp2_fast(da.x), exp2_fast(da.y)};
                    smb += f2{exp2_fast(db.x), exp2_fast(db.y)};
                }
                mxa = nma;
                mxb = nmb;
            };
            int k0 = 0;                           // (nqh is wave-uniform and a multiple of 4)
            for (; k0 + 16 <= nqh; k0 += 16) block(std::integral_constant<int, 16>{}, k0);
            if (nqh - k0 >= 8) {
                block(std::integral_constant<int, 8>{}, k0);
                k0 += 8;
            }
            if (nqh - k0 >= 4) block(std::integral_constant<int, 4>{}, k0);
            SEC(4);   // pair loop
            const float mx[4] = {mxa.x, mxa.y, mxb.x, mxb.y};
            const float sm[4] = {sma.x, sma.y, smb.x, smb.y};
            // merge the four j-slices of the quad (lane q keeps sub-particle i == q) and the chain's halves:
            // common maximum first, then ONE rescale of the lane's own sum and plain adds
            float lm = ninf, ls = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float gm = fmaxf(mx[i], xor_lane<1>(mx[i]));
                gm = fmaxf(gm, xor_lane<2>(gm));
                if constexpr (HS == 2) gm = fmaxf(gm, xor_lane<M>(gm));
                const float base = (gm == ninf) ? 0.f : gm;
                float sc = sm[i] * exp2_fast(mx[i] - base);    // (mx = -inf: sm = 0 and exp2(-inf) = 0)
                sc += xor_lane<1>(sc);
                sc += xor_lane<2>(sc);
                if constexpr (HS == 2) sc += xor_lane<M>(sc);
                if (i == q) {
                    lm = gm;
                    ls = sc;
                }
            }
            const float lam2 = lm + log2_fast(ls);
            lam = fmaf(kLn2, lam2, kf);
            if (a.lam2_all && valid && hpart == 0) a.lam2_all[(tb * N + n) * M + m] = lam2;
        } else {
            lam = diag_lp<DX>(x, im, i_isig, kiota);  // t = 0: q0 / f density at mu_0 (PSVO.py:169-175)
        }

        SEC(5);   // quad / half merges, lam2
        // ---- omega, normalise over the M sub-particles, draw one ---------------------------------------
        const float om_raw = lam + phi + g_lp - q_lp;
        const float omx = group_max<M>(om_raw);
        const float pw = exp2_fast((om_raw - omx) * kLog2e);
        const float cdfv = group_incl_scan<M>(pw, m);  // inclusive scan across the chain's M lanes
        const float total = __shfl(cdfv, gbase + M - 1);
        const float omega = om_raw - fmaf(kLn2, log2_fast(total), omx);
        if (a.om_all && valid && hpart == 0) a.om_all[(tb * N + n) * M + m] = omega;
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
        const float om_s = __shfl(omega, src);
        const float phi_s = __shfl(phi, src);
        SEC(6);   // normalise over M, draw
        const float g_s = __shfl(g_lp, src);
        const float q_s = __shfl(q_lp, src);
        const float lam_s = __shfl(lam, src);
        const float Om = om_s + q_s + logM;

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

#pragma unroll
        for (int d = 0; d < DX; ++d) {
            xp[d] = xs[d];
            eps_c[d] = eps_n[d];
            bmu_c[d] = bmu_n[d];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) obs_c[k] = obs_n[k];
        u_c = u_n;
        sel_c = sel_n;

        SEC(7);   // gather of the selected sub-particle, output stores
        if (t >= 2) stage_store(nxt);
        __syncthreads();
        SEC(8);   // tile store + barrier
    };
    if constexpr (kPeel) {
        step(std::true_type{}, std::false_type{}, T - 1);
        for (int t = T - 2; t >= 1; --t) step(std::false_type{}, std::false_type{}, t);
        step(std::false_type{}, std::true_type{}, 0);
    } else {
        for (int t = T - 1; t >= 0; --t) step(t == T - 1, t == 0, t);
    }
    if (valid && lead) a.score[(size_t)b * N + n] = score;
}

template <int DX, int DY, int H, int M>
static int launch_bsim(const BsimArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    constexpr int PS = TileSlot<DX>::kFloats;
    // fewer than two waves per SIMD (1024 SIMDs) with one lane per (chain, m): spread each chain over 2M lanes
    const long long waves1 = ((long long)a.B * a.N * M + 63) / 64;
    const int HS = (waves1 < 2048 && 2 * M <= 64 && (H / 2) % 4 == 0 && (PSVO_L == 1 || g_tune_l2_split)) ? 2 : 1;
    const int NP = ((a.N + 16 * HS - 1) / (16 * HS)) * (16 * HS);
    int NTB = ((a.N * M * HS + 63) / 64) * 64;
    if (NTB > 256) NTB = 256;
    const int cpb = NTB / (M * HS);
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * PS);
    dim3 grid((a.N + cpb - 1) / cpb, a.B);
    clear_hip_error();
    if (HS == 2) {
        if constexpr (2 * M <= 64 && (H / 2) % 4 == 0)
            hipLaunchKernelGGL((bsim_fwd_kernel<DX, DY, H, M, 2>), grid, dim3(NTB), lds, stream, a);
    } else {
        hipLaunchKernelGGL((bsim_fwd_kernel<DX, DY, H, M, 1>), grid, dim3(NTB), lds, stream, a);
    }
    return launch_status();
}

template <int DX, int DY, int H>
static int bsim_dispatch_m(const BsimArgs& a, int M, hipStream_t s) {
    switch (M) {
        case 4: return launch_bsim<DX, DY, H, 4>(a, s);
        case 8: return launch_bsim<DX, DY, H, 8>(a, s);
        case 16: return launch_bsim<DX, DY, H, 16>(a, s);
        case 32: return launch_bsim<DX, DY, H, 32>(a, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX, int DY>
static int bsim_dispatch_h(const BsimArgs& a, int H, int M, hipStream_t s) {
    switch (H) {
#if PSVO_L == 1   // (two hidden layers: widths 32 and 64; narrower ones are zero-padded upstream)
        case 16: return bsim_dispatch_m<DX, DY, 16>(a, M, s);
#endif
        case 32: return bsim_dispatch_m<DX, DY, 32>(a, M, s);
        case 64: return bsim_dispatch_m<DX, DY, 64>(a, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX>
static int bsim_dispatch_dy(const BsimArgs& a, int Dy, int H, int M, hipStream_t s) {
    switch (Dy) {
        case 1: return bsim_dispatch_h<DX, 1>(a, H, M, s);
        case 2: return bsim_dispatch_h<DX, 2>(a, H, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // inline namespace PSVO_LNS
}  // namespace psvo

PSVO_L2_DECL(psvo_bsim_forward)
PSVO_ENTRY(psvo_bsim_forward)(const psvo_desc* desc, const float* X, const float* Fm, const float* logW,
                                 const float* lse, const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv,
                                 const float* sig_f, const float* sig_g, const float* sig_q1inv,
                                 const float* sig_bq2, const float* bmu2, const float* minit,
                                 const float* sig_init, const float* imean, const float* isig, const float* obs,
                                 const float* eps_b, const float* u_b, const int32_t* sel_in, float* bwX,
                                 float* flp, float* glp, float* Omega, int32_t* sel_out, float* score,
                                 float* lam2_all, float* om_all, float* mu1_all,
                                 void* stream) {
    using namespace psvo;
    if (!desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;
#if PSVO_L == 1
    if (desc && desc->layers == 2)
        return psvo_bsim_forward_l2(desc, X, Fm, logW, lse, f, g, q1_inv, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2,
            minit, sig_init, imean, isig, obs, eps_b, u_b, sel_in, bwX, flp, glp, Omega, sel_out, score, lam2_all,
            om_all, mu1_all, stream);
#endif
    if (!mlp_layers_ok(f) || !mlp_layers_ok(g) || !mlp_layers_ok(q1_inv)) return PSVO_ERR_INVALID;
    if (!desc || !Fm || !logW || !lse || !f || !g || !q1_inv || !sig_f || !sig_g || !sig_q1inv || !sig_bq2 ||
        !bmu2 || !minit || !sig_init || !imean || !isig || !obs || !eps_b || !bwX || !flp || !glp || !Omega ||
        !sel_out || !score)
        return PSVO_ERR_INVALID;
    if (!u_b && !sel_in) return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T < 2 || desc->N <= 0 || desc->M <= 0) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    (void)X;

    BsimArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.X = X; a.Fm = Fm; a.logW = logW; a.lse = lse;
    a.sig_f = sig_f; a.sig_g = sig_g; a.sig_q1inv = sig_q1inv; a.sig_bq2 = sig_bq2;
    a.bmu2 = bmu2; a.minit = minit; a.sig_init = sig_init; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.u_b = u_b; a.sel_in = sel_in;
    a.bwX = bwX; a.flp = flp; a.glp = glp; a.Omega = Omega; a.sel_out = sel_out; a.score = score;
    a.lam2_all = lam2_all; a.om_all = om_all; a.mu1_all = mu1_all;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return bsim_dispatch_dy<2>(a, desc->Dy, desc->H, desc->M, s);
        case 3: return bsim_dispatch_dy<3>(a, desc->Dy, desc->H, desc->M, s);
        case 4: return bsim_dispatch_dy<4>(a, desc->Dy, desc->H, desc->M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
