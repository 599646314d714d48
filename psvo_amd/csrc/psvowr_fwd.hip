// PSVOwR backward simulation: PSVO's backward simulation with an additional multinomial resampling
// ACROSS the chains of a sequence after every step and a per-step ELBO (reference
// src/SMC/PSVOwR.py:65-198).
//
// The cross-chain draw couples all N chains of a sequence once per step.  A sequence is therefore owned by a
// CLUSTER of K persistent workgroups (K = 1, 2, 4 or 8; cooperative launch, so that all of them are resident):
// workgroup k walks the (chain, sub-particle) items of chains [k Nc, (k+1) Nc), publishes the selected
// sub-particle, its normalised log-weight omega_sel and the step weight of each of its chains as TAGGED 64-bit
// words {value, step tag} in a two-step ring in HBM; every workgroup then polls the N chains' words until the
// tags match (bounded spin), rebuilds the N-entry CDF and draws the ancestors of ITS OWN chains.  The data words
// are the synchronisation: no counter barrier, no store acknowledgement to wait for and no second round trip for
// the gathered states -- one one-way HBM hop per step instead of five serialised ones.
// The per-item work is the arithmetic of psvo_bsim_forward (packed f32 pair loop over the LDS-staged tile).
//
// With omega_raw = Lambda + phi + g - q the reference's per-step weight
//     bw_log_W = (Lambda + g)_sel - q_sel - omega_sel - log M          (PSVOwR.py:135-142)
// equals  logsumexp_m(omega_raw) - phi_sel - log M, which is what the kernel writes.
#include "common.h"

namespace psvo {
inline namespace PSVO_LNS {   // l1 / l2: hidden layers of the per-particle MLPs (common.h)

struct WrArgs {
    int B, T, N, emission;
    psvo_mlp f, g, q1inv;
    const float *Fm, *logW, *lse;
    const float *sig_f, *sig_g, *sig_q1inv, *sig_bq2;
    const float *bmu2, *minit, *sig_init, *imean, *isig;
    const float *obs, *eps_b, *u_b, *u_r;
    const int32_t *sel_in, *anc_in;
    float *bwX, *bwXanc, *bwW, *lseW;
    int32_t *sel_out, *anc_out;
    float *lam2_all, *om_all, *mu1_all;
    unsigned long long* ring;   // [2][B][N][kWrWords] tagged words {bits(value), t + 1} (zeroed before the launch)
    unsigned* err;              // error flag: a poll timed out (zeroed before the launch)
};

// words a chain publishes per step: Dx selected states, omega_sel, bw_log_W (sized for Dx <= 4)
constexpr int kWrWords = 6;

__device__ __forceinline__ void st_tagged(unsigned long long* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// workspace: the ring [2][B][N][kWrWords] of 64-bit words, then two 32-bit words whose last one is the error flag
static inline long long wr_ws_floats(int B, int N) { return 2ll * (2ll * B * N * kWrWords) + 2; }

template <int DX>
struct WrSlot {
    static constexpr int kFloats = (DX <= 3) ? 4 : 8;
};

template <int DX>
__device__ __forceinline__ void wr_read_slot(const float* p, float (&F)[DX], float& W) {
    const float4 e = *reinterpret_cast<const float4*>(p);
    if constexpr (DX <= 3) {
        F[0] = e.x;
        if constexpr (DX > 1) F[1] = e.y;
        if constexpr (DX > 2) F[2] = e.z;
        W = e.w;
    } else {
        F[0] = e.x; F[1] = e.y; F[2] = e.z; F[3] = e.w;
        W = p[4];
    }
}

template <int DX, int DY, int H, int M, int MAXT>
__global__ void __launch_bounds__(MAXT) psvowr_fwd_kernel(const WrArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    constexpr int PS = WrSlot<DX>::kFloats;
    constexpr bool kRolled = true;
    constexpr int JB = (MAXT > 256) ? 4 : 16;      // pair-loop block (entries held in registers; 128 VGPRs at 1024 lanes)
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTB = blockDim.x, nw = NTB >> 6;
    const int B = a.B, T = a.T, N = a.N;
    const int NP = ((N + 4 * JB - 1) / (4 * JB)) * (4 * JB);   // tile padded to whole blocks of JB entries
    const int b = blockIdx.y, kb = blockIdx.x, K = gridDim.x;
    const int Nc = (N + K - 1) / K;                // chains per workgroup
    const int c0 = kb * Nc, c1 = min(N, c0 + Nc);  // this workgroup's chains
    const int cpr = NTB / M;                       // chains per round
    const int rounds = (Nc + cpr - 1) / cpr;
    const int cl = tid / M, m = tid % M, q = m & 3;
    const int gbase = lane - m;
    unsigned* const err = a.err;

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;                 // [2][NP][PS]
    float* xanc = tile + 2 * NP * PS;              // [DX][Nc] resampled states (x_{t+1}) of this workgroup's chains
    float* cdf = xanc + DX * Nc;                   // [N]      CDF of the cross-chain draw
    float* xall = cdf + N;                         // [DX][N]  selected states of all chains (polled from the ring)
    float* red = xall + DX * N;                    // 64 floats scratch

    MQ::load(wf, a.f, tid, NTB);
    MG::load(wg, a.g, tid, NTB);
    MQ::load(wqi, a.q1inv, tid, NTB);

    float isf[DX], rp[DX], isg[DY];
    float kf = -DX * kHalfLog2Pi, kg = -DY * kHalfLog2Pi;
    const float rscale = sqrtf(0.5f * kLog2e);
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        const float s = a.sig_f[d];
        isf[d] = 1.f / s;
        rp[d] = isf[d] * rscale;
        kf -= logf(s);
    }
#pragma unroll
    for (int e = 0; e < DY; ++e) {
        const float s = a.sig_g[e];
        isg[e] = 1.f / s;
        kg -= logf(s);
    }
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

    auto stage = [&](int tt, float* buf) {  // forward tile of step tt: (F', W') slots
        const size_t tb = (size_t)tt * B + b;
        const float l = a.lse[tb];
        for (int j = tid; j < NP; j += NTB) {
            const int jc = j < N ? j : N - 1;
            float v[DX + 1];
#pragma unroll
            for (int d = 0; d < DX; ++d) v[d] = a.Fm[(tb * DX + d) * N + jc] * rp[d];
            v[DX] = j < N ? (a.logW[tb * N + jc] - l) * kLog2e : ninf;
            if constexpr (DX <= 3) {
                float4 o;
                o.x = v[0];
                o.y = DX > 1 ? v[DX > 1 ? 1 : 0] : 0.f;
                o.z = DX > 2 ? v[DX > 2 ? 2 : 0] : 0.f;
                o.w = v[DX];
                *reinterpret_cast<float4*>(buf + j * PS) = o;
            } else {
                *reinterpret_cast<float4*>(buf + j * PS) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(buf + j * PS + 4) = make_float4(v[4], 0.f, 0.f, 0.f);
            }
        }
    };
    // One tile entry per thread (the usual case, NP <= workgroup size): the entry's raw values are only REQUESTED at the top
    // of a step and scaled / stored at its end, so that no step opens with an exposed HBM round trip (stage() above
    // does both at once and remains for larger N).
    const bool one_entry = NP <= NTB;
    float raw[DX + 1], rawl = 0.f;
#pragma unroll
    for (int d = 0; d <= DX; ++d) raw[d] = 0.f;
    auto stage_load = [&](int tt) {
        const size_t tb = (size_t)tt * B + b;
        if (tid < NP) {
            const int jc = tid < N ? tid : N - 1;
#pragma unroll
            for (int d = 0; d < DX; ++d) raw[d] = a.Fm[(tb * DX + d) * N + jc];
            raw[DX] = a.logW[tb * N + jc];
            rawl = a.lse[tb];
        }
    };
    auto stage_store = [&](float* buf) {
        if (tid < NP) {
            float v[DX + 1];
#pragma unroll
            for (int d = 0; d < DX; ++d) v[d] = raw[d] * rp[d];
            v[DX] = tid < N ? (raw[DX] - rawl) * kLog2e : ninf;
            if constexpr (DX <= 3) {
                float4 o;
                o.x = v[0];
                o.y = DX > 1 ? v[DX > 1 ? 1 : 0] : 0.f;
                o.z = DX > 2 ? v[DX > 2 ? 2 : 0] : 0.f;
                o.w = v[DX];
                *reinterpret_cast<float4*>(buf + tid * PS) = o;
            } else {
                *reinterpret_cast<float4*>(buf + tid * PS) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(buf + tid * PS + 4) = make_float4(v[4], 0.f, 0.f, 0.f);
            }
        }
    };
    if (T >= 2) stage(T - 2, tile);
    __syncthreads();

    // per-step inputs and round 0's noise, loaded one exchange ahead
    float bm_n[DX], y_n[DY], eps_n[DX];
    auto load_step = [&](size_t tb) {
        const bool valid0 = (c0 + cl) < c1 && cl < Nc;
        const int n0 = valid0 ? c0 + cl : max(c1 - 1, 0);
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            bm_n[d] = a.bmu2[tb * DX + d];
            eps_n[d] = a.eps_b[((tb * DX + d) * N + n0) * M + m];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y_n[k] = a.obs[tb * DY + k];
    };
    load_step((size_t)(T - 1) * B + b);

    for (int t = T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * B + b;
        // ring slot of the step: a workgroup can run at most one step ahead of the slowest one of its cluster (it needs
        // everybody's step-t words to get past step t), so two slots suffice; the tag tells the steps apart
        unsigned long long* const slot = a.ring + ((size_t)(t & 1) * B + b) * N * kWrWords;
        const unsigned tag = (unsigned)(t + 1);
        const float* cur = tile + ((T - 1 - t) & 1) * NP * PS;
        float* nxt = tile + ((T - t) & 1) * NP * PS;
        const bool last = (t == T - 1);
        const bool staging = (t >= 2);     // tile of step t-2, read during step t-1
        if (staging) {
            if (one_entry) stage_load(t - 2);
            else stage(t - 2, nxt);
        }

        float bm[DX], y[DY];
#pragma unroll
        for (int d = 0; d < DX; ++d) bm[d] = bm_n[d];
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = y_n[k];

        for (int r = 0; r < rounds; ++r) {
            const int n_raw = c0 + r * cpr + cl;
            const bool valid = n_raw < c1 && (r * cpr + cl) < Nc;
            const int n = valid ? n_raw : max(c1 - 1, 0);
            const int nl = n - c0;                 // index inside this workgroup's chains
            float xp[DX], eps[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                xp[d] = last ? 0.f : xanc[d * Nc + nl];
                eps[d] = (r == 0) ? eps_n[d] : a.eps_b[((tb * DX + d) * N + n) * M + m];
            }
            // ---- proposal ------------------------------------------------------------------------
            float x[DX], q_lp;
            if (last) {
                float mu[DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    mu[d] = mi[d];
                    x[d] = fmaf(s_init[d], eps[d], mu[d]);
                }
                q_lp = diag_lp<DX>(x, mu, is_init, kinit);
                if (a.mu1_all && valid && m == 0) {
#pragma unroll
                    for (int d = 0; d < DX; ++d) a.mu1_all[(tb * DX + d) * N + n] = 0.f;
                }
            } else {
                float m1[DX], mu[DX];
                MQ::template eval<kRolled>(wqi, xp, m1);
                if (a.mu1_all && valid && m == 0) {
#pragma unroll
                    for (int d = 0; d < DX; ++d) a.mu1_all[(tb * DX + d) * N + n] = m1[d];
                }
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    mu[d] = pc[d] * fmaf(pi1[d], m1[d], pi2[d] * bm[d]);
                    x[d] = fmaf(pc[d], eps[d], mu[d]);
                }
                q_lp = diag_lp<DX>(x, mu, pic, kq);
            }
            float phi = 0.f;
            if (!last) {
                float fmx[DX];
                MQ::template eval<kRolled>(wf, x, fmx);
                phi = diag_lp<DX>(xp, fmx, isf, kf);
            }
            float gm[DY];
            MG::template eval<kRolled>(wg, x, gm);
            if (a.emission) {
#pragma unroll
                for (int k = 0; k < DY; ++k) gm[k] = emis_mean(gm[k]);
            }
            const float g_lp = diag_lp<DY>(y, gm, isg, kg);

            // ---- filter term over the LDS tile: quad register blocking, packed f32, block-wise log-sum-exp --------
            float lam;
            if (t >= 1) {
                f2 xa[DX], xb[DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    float t4[4];
                    quad_bcast4(x[d] * rp[d], t4);
                    xa[d] = f2{t4[0], t4[1]};
                    xb[d] = f2{t4[2], t4[3]};
                }
                f2 mxa = f2{ninf, ninf}, mxb = mxa, sma = f2{0.f, 0.f}, smb = sma;
                const int nq = NP >> 2;            // a multiple of JB
                for (int k0 = 0; k0 < nq; k0 += JB) {
                    f2 va[JB], vb[JB];
                    f2 bma = f2{ninf, ninf}, bmb = bma;
#pragma unroll
                    for (int c = 0; c < JB; ++c) {
                        float F[DX], W;
                        wr_read_slot<DX>(cur + ((k0 + c) * 4 + q) * PS, F, W);
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
                }
                const float mx[4] = {mxa.x, mxa.y, mxb.x, mxb.y};
                const float sm[4] = {sma.x, sma.y, smb.x, smb.y};
                float lm = ninf, ls = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // merge the quad's four j-slices; lane q keeps sub-particle i == q
                    float gmx = fmaxf(mx[i], xor_lane<1>(mx[i]));
                    gmx = fmaxf(gmx, xor_lane<2>(gmx));
                    const float base = (gmx == ninf) ? 0.f : gmx;
                    float sc = sm[i] * exp2_fast(mx[i] - base);
                    sc += xor_lane<1>(sc);
                    sc += xor_lane<2>(sc);
                    if (i == q) {
                        lm = gmx;
                        ls = sc;
                    }
                }
                const float lam2 = lm + log2_fast(ls);
                lam = fmaf(kLn2, lam2, kf);
                if (a.lam2_all && valid) a.lam2_all[(tb * N + n) * M + m] = lam2;
            } else {
                lam = diag_lp<DX>(x, im, i_isig, kiota);
            }

            // ---- omega over the M sub-particles, draw one ------------------------------------------------------
            const float om_raw = lam + phi + g_lp - q_lp;
            const float omx = group_max<M>(om_raw);
            const float pw = expf(om_raw - omx);
            const float cdfv = group_incl_scan<M>(pw, m);
            const float total = __shfl(cdfv, gbase + M - 1);
            const float lse_m = omx + logf(total);
            const float omega = om_raw - lse_m;
            if (a.om_all && valid) a.om_all[(tb * N + n) * M + m] = omega;
            int sel;
            if (a.sel_in) {
                sel = a.sel_in[tb * N + n];
            } else {
                const float u = a.u_b[tb * N + n];
                const unsigned long long bal = __ballot(cdfv <= u * total);
                const unsigned long long mask = (M == 64) ? ~0ull : (((1ull << M) - 1ull) << gbase);
                sel = min((int)__popcll(bal & mask), M - 1);
            }
            const int src = gbase + sel;
            float xs[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) xs[d] = __shfl(x[d], src);
            const float om_s = __shfl(omega, src);
            const float phi_s = __shfl(phi, src);
            const float bw = lse_m - phi_s - logM;   // == (Lambda + g)_sel - q_sel - omega_sel - log M
            if (valid && m == 0) {                   // published to the cluster through the ring
                unsigned long long* const w = slot + (size_t)n * kWrWords;
#pragma unroll
                for (int d = 0; d < DX; ++d) st_tagged(w + d, xs[d], tag);
                st_tagged(w + DX, om_s, tag);
                st_tagged(w + DX + 1, bw, tag);
#pragma unroll
                for (int d = 0; d < DX; ++d) a.bwX[(tb * DX + d) * N + n] = xs[d];
                a.bwW[tb * N + n] = bw;
                a.sel_out[tb * N + n] = sel;
            }
        }

        // ---- resample the chains: a[k] ~ Categorical(softmax_n omega_sel[n]) (PSVOwR.py:103,145,185); every workgroup
        //      rebuilds the CDF over all N chains and draws the ancestors of its own ----------------------------------
        {
            // the next step's inputs are requested before the poll so that their HBM latency hides behind it (issue only)
            if (t > 0) load_step(tb - B);
            const bool act = tid < N;
            float lo = ninf, lw = ninf;
            if (act) {   // poll chain `tid`'s words of this step (bounded: a timeout raises the error flag and drains)
                const unsigned long long* const w = slot + (size_t)tid * kWrWords;
                unsigned spins = 0;
                for (;;) {
                    unsigned long long v[DX + 2];
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < DX + 2; ++i) {
                        v[i] = __hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned)(v[i] >> 32) == tag;
                    }
                    if (ok) {
#pragma unroll
                        for (int d = 0; d < DX; ++d) xall[d * N + tid] = __uint_as_float((unsigned)v[d]);
                        lo = __uint_as_float((unsigned)v[DX]);
                        if (kb == 0) lw = __uint_as_float((unsigned)v[DX + 1]);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 21) ||
                        ((spins & 63u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            float m1 = wave_max(lo), m2 = wave_max(lw);
            if (lane == 0) {
                red[wave] = m1;
                red[16 + wave] = m2;
            }
            __syncthreads();
            float mo = ninf, mw = ninf;
            for (int i = 0; i < nw; ++i) {
                mo = fmaxf(mo, red[i]);
                mw = fmaxf(mw, red[16 + i]);
            }
            __syncthreads();
            const float w = act ? expf(lo - mo) : 0.f;
            const float ew = (act && kb == 0) ? expf(lw - mw) : 0.f;
            float sc = wave_incl_scan(w, lane);
            const float sw = wave_sum(ew);
            if (lane == 63) red[wave] = sc;
            if (lane == 0) red[16 + wave] = sw;
            __syncthreads();
            float off = 0.f, tot = 0.f, totw = 0.f;
            for (int i = 0; i < nw; ++i) {
                const float v = red[i];
                if (i < wave) off += v;
                tot += v;
                totw += red[16 + i];
            }
            sc += off;
            if (act) cdf[tid] = sc;
            if (tid == 0 && kb == 0) a.lseW[tb] = mw + logf(totw);   // logsumexp_n bw_log_W[t, :, b]
            __syncthreads();
            const int k = c0 + tid;                 // own chain whose ancestor this lane draws
            if (tid < Nc && k < c1) {
                int anc;
                if (a.anc_in) {
                    anc = a.anc_in[tb * N + k];
                } else {
                    const float target = a.u_r[tb * N + k] * tot;
                    int pos = 0;
                    for (int s = 1 << (31 - __clz(N)); s > 0; s >>= 1) {
                        const int p = pos + s;
                        if (p <= N && cdf[p - 1] <= target) pos = p;
                    }
                    anc = min(pos, N - 1);
                }
                a.anc_out[tb * N + k] = anc;
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    const float v = xall[d * N + anc];   // (written before the CDF's workgroup barriers)
                    xanc[d * Nc + tid] = v;         // (xanc is only read in the item rounds, after the closing barrier)
                    a.bwXanc[(tb * DX + d) * N + k] = v;
                }
            }
        }
        if (staging && one_entry) stage_store(nxt);
        __syncthreads();
    }
}

template <int DX, int DY, int H, int M>
static int launch_wr_fwd(const WrArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    constexpr int PS = WrSlot<DX>::kFloats;
    const int K = wr_cluster(a.B, a.N, M);
    const int Nc = (a.N + K - 1) / K;
    long long items = (long long)Nc * M;
    int NTB = (int)(((items + 63) / 64) * 64);
    const int NN = (a.N + 63) & ~63;               // the cross-chain draw uses one lane per chain
    if (NTB < NN) NTB = NN;
    if (NTB > 1024) NTB = 1024;
    if (a.N > NTB) return PSVO_ERR_UNSUPPORTED;
    const int JB = NTB > 256 ? 4 : 16;
    const int NP = ((a.N + 4 * JB - 1) / (4 * JB)) * (4 * JB);
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * PS + (size_t)DX * Nc + a.N +
                                        (size_t)DX * a.N + 64);
    if (lds > 160 * 1024) return PSVO_ERR_UNSUPPORTED;
    clear_hip_error();
    // tags of an earlier launch must not be mistaken for this one's: clear the ring and the error flag
    if (hipMemsetAsync(a.ring, 0, sizeof(float) * (size_t)wr_ws_floats(a.B, a.N), stream) != hipSuccess)
        return launch_status();
    WrArgs args = a;
    void* kargs[] = {(void*)&args};
    const dim3 grid(K, a.B), block(NTB);
    hipError_t e;
    const void* fn = NTB > 256 ? (const void*)psvowr_fwd_kernel<DX, DY, H, M, 1024>
                               : (const void*)psvowr_fwd_kernel<DX, DY, H, M, 256>;
    // a cluster (K > 1) needs all its workgroups resident: cooperative launch.  K == 1 (many sequences, or few chains)
    // has no cross-workgroup exchange -- a workgroup only reads back its own words -- so the grid may exceed what fits
    // on the device at once, which a cooperative launch would refuse.
    if (K > 1) e = hipLaunchCooperativeKernel(fn, grid, block, kargs, lds, stream);
    else e = hipLaunchKernel(fn, grid, block, kargs, lds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = e;
        (void)hipGetLastError();
        return PSVO_ERR_HIP;
    }
    return launch_status();
}

template <int DX, int DY, int H>
static int wr_dispatch_m(const WrArgs& a, int M, hipStream_t s) {
    switch (M) {
        case 4: return launch_wr_fwd<DX, DY, H, 4>(a, s);
        case 8: return launch_wr_fwd<DX, DY, H, 8>(a, s);
        case 16: return launch_wr_fwd<DX, DY, H, 16>(a, s);
        case 32: return launch_wr_fwd<DX, DY, H, 32>(a, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX, int DY>
static int wr_dispatch_h(const WrArgs& a, int H, int M, hipStream_t s) {
    switch (H) {
#if PSVO_L == 1   // (two hidden layers: widths 32 and 64; narrower ones are zero-padded upstream)
        case 16: return wr_dispatch_m<DX, DY, 16>(a, M, s);
#endif
        case 32: return wr_dispatch_m<DX, DY, 32>(a, M, s);
        case 64: return wr_dispatch_m<DX, DY, 64>(a, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX>
static int wr_dispatch_dy(const WrArgs& a, int Dy, int H, int M, hipStream_t s) {
    switch (Dy) {
        case 1: return wr_dispatch_h<DX, 1>(a, H, M, s);
        case 2: return wr_dispatch_h<DX, 2>(a, H, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // inline namespace PSVO_LNS
}  // namespace psvo

#if PSVO_L == 1   // (sizing helpers: independent of the number of hidden layers)
extern "C" int psvo_bsimwr_blocks(int B, int N, int M) { return psvo::wr_cluster(B, N, M); }

extern "C" long long psvo_bsimwr_ws_floats(int B, int T, int N) {
    (void)T;
    return psvo::wr_ws_floats(B, N);
}
#endif

PSVO_L2_DECL(psvo_bsimwr_forward)
PSVO_ENTRY(psvo_bsimwr_forward)(const psvo_desc* desc, const float* Fm, const float* logW, const float* lse,
                                   const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv, const float* sig_f,
                                   const float* sig_g, const float* sig_q1inv, const float* sig_bq2, const float* bmu2,
                                   const float* minit, const float* sig_init, const float* imean, const float* isig,
                                   const float* obs, const float* eps_b, const float* u_b, const float* u_r,
                                   const int32_t* sel_in, const int32_t* anc_in, float* bwX, float* bwXanc, float* bwW,
                                   float* lseW, int32_t* sel_out, int32_t* anc_out, float* lam2_all, float* om_all,
                                   float* mu1_all, float* ws, void* stream) {
    using namespace psvo;
    if (!desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;
#if PSVO_L == 1
    if (desc && desc->layers == 2)
        return psvo_bsimwr_forward_l2(desc, Fm, logW, lse, f, g, q1_inv, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2,
            minit, sig_init, imean, isig, obs, eps_b, u_b, u_r, sel_in, anc_in, bwX, bwXanc, bwW, lseW, sel_out,
            anc_out, lam2_all, om_all, mu1_all, ws, stream);
#endif
    if (!mlp_layers_ok(f) || !mlp_layers_ok(g) || !mlp_layers_ok(q1_inv)) return PSVO_ERR_INVALID;
    if (!desc || !Fm || !logW || !lse || !f || !g || !q1_inv || !sig_f || !sig_g || !sig_q1inv || !sig_bq2 || !bmu2 ||
        !minit || !sig_init || !imean || !isig || !obs || !eps_b || !bwX || !bwXanc || !bwW || !lseW || !sel_out ||
        !anc_out || !ws)
        return PSVO_ERR_INVALID;
    if ((!u_b && !sel_in) || (!u_r && !anc_in)) return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T < 2 || desc->N <= 0 || desc->M <= 0) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    WrArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.logW = logW; a.lse = lse;
    a.sig_f = sig_f; a.sig_g = sig_g; a.sig_q1inv = sig_q1inv; a.sig_bq2 = sig_bq2;
    a.bmu2 = bmu2; a.minit = minit; a.sig_init = sig_init; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.u_b = u_b; a.u_r = u_r; a.sel_in = sel_in; a.anc_in = anc_in;
    a.bwX = bwX; a.bwXanc = bwXanc; a.bwW = bwW; a.lseW = lseW; a.sel_out = sel_out; a.anc_out = anc_out;
    a.lam2_all = lam2_all; a.om_all = om_all; a.mu1_all = mu1_all;
    if (reinterpret_cast<uintptr_t>(ws) & 7u) return PSVO_ERR_INVALID;      // 64-bit words
    a.ring = reinterpret_cast<unsigned long long*>(ws);
    a.err = reinterpret_cast<unsigned*>(ws + wr_ws_floats(desc->B, desc->N) - 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return wr_dispatch_dy<2>(a, desc->Dy, desc->H, desc->M, s);
        case 3: return wr_dispatch_dy<3>(a, desc->Dy, desc->H, desc->M, s);
        case 4: return wr_dispatch_dy<4>(a, desc->Dy, desc->H, desc->M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
