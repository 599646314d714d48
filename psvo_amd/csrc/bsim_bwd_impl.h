// Reverse-mode pass of the backward simulation (gradient of sum_t(f+g-Omega) per chain w.r.t. every
// input of psvo_bsim_forward).  The reference obtains it from TensorFlow autodiff, which stores the
// (M, N, N, B[, Dx]) transition tile of every step (reference src/SMC/PSVO.py:128-133; SURVEY.md
// section 5 "Long-context"); here the tile is RECOMPUTED from the LDS-staged forward particles and
// only three small per-(t, chain[, m]) arrays saved by the forward kernel are read back
// (lam2 = log2-domain filter term, omega = normalised sub-particle log-weights, mu1 = MLP_q1inv(x+)).
//
// Reverse order is t = 0 .. T-1 (the forward ran T-1 .. 0); the only state carried between steps is
// d loss / d bwX_{t+1} per chain.  Same lane mapping as the forward kernel: lane = (chain, m), quad
// register blocking over four m in the pair loop.  With a = d loss / d score[chain] and
// pi_m = exp(omega_m):
//     d phi_m = d g_m = a pi_m,   d Lambda_m = -a (delta_{m,sel} - pi_m),   d q_m = -a pi_m
// and, for p_mj = softmax_j(log f(x~_m | F_j) + W^_j) recomputed from lam2:
//     d x~_m  -= dLambda_m sum_j p_mj (x~_m - F_j) / sigma_f^2
//     d F_j   += sum_{chains, m} dLambda_m p_mj (x~_m - F_j) / sigma_f^2      (cross-chain reduction)
//     d W^_j  += sum_{chains, m} dLambda_m p_mj                                (sums to 0 over j)
// The cross-chain sums are formed without atomics: each lane keeps the per-j partials of a chunk of
// forward particles in registers, a 4-stage butterfly reduce-scatters them over the 16 quads of the
// wave, the four waves are folded in a fixed order through LDS and written as per-workgroup partials
// (T, B, nblk, ...) that psvo_filter_backward sums -- no global atomics, no cross-workgroup order.
//
// MLP weight gradients are left to psvo_mlp_wgrad: this kernel writes the rows x~ (xt) and the
// output gradients dFt (MLP_f), dGt (MLP_g), dmu1 (MLP_q1inv).
#pragma once
#include "common.h"

namespace psvo {
#if defined(PSVO_SECTION_TIMERS) && defined(PSVO_BSIM_BWD_V2_UNIT)
extern __device__ unsigned long long g_sec_bsim_bwd[32];   // (defined by the v1 translation unit of the diagnostic build)
#else
PSVO_TIMERS_DEFINE(bsim_bwd)
#endif


struct BsimBwdArgs {
    int B, T, N, emission;
    psvo_mlp f, g, q1inv;
    const float *Fm, *logW, *lse;
    const float *sig_f, *sig_g, *sig_q1inv, *sig_bq2;
    const float *bmu2, *minit, *sig_init, *imean, *isig;
    const float *obs, *eps_b;
    const float* bwX;
    const int32_t* sel;
    const float *lam2_all, *om_all, *mu1_all;
    const float* dscore;  // (B,N)
    float *xt, *dFt, *dGt, *dmu1;
    float *dFm_part, *dlogW_part, *dbmu2_rows, *dminit_rows, *dimean_rows, *sacc_part;
    int skew;        // phase offset (shader-clock cycles) of every other resident workgroup: common.h, phase_skew()
};

template <int DX, int DY>
struct BAcc {
    static constexpr int kSc = 0;           // PoG: direct d c
    static constexpr int kSmm1 = DX;        // sum dmu * mu1
    static constexpr int kSmb = 2 * DX;     // sum dmu * bmu2
    static constexpr int kSmm = 3 * DX;     // sum dmu * mu
    static constexpr int kSf = 4 * DX;      // d sigma_f
    static constexpr int kSinit = 5 * DX;   // d sigma_init
    static constexpr int kSiota = 6 * DX;   // d isig
    static constexpr int kSg = 7 * DX;      // d sigma_g (DY)
    static constexpr int kN = 7 * DX + DY;
};

template <int DX>
struct BTileSlot {
    static constexpr int kFloats = (DX <= 3) ? 4 : 8;
};

__device__ __forceinline__ float quad_bcast_b(float v, int lane, int i) { return __shfl(v, (lane & ~3) | i); }

template <int DX>
__device__ __forceinline__ void read_slot(const float* p, float (&F)[DX], float& W) {
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

// One butterfly stage of the reduce-scatter over quads: the NN live entries are halved; the lane
// keeps the half selected by its `bit` and adds the partner's (lane ^ mask) copy of that half.
// (v_permlane{32,16}_swap of (lo, hi) would do a stage without the selects, but hipcc 7.2 maps both results of
//  __builtin_amdgcn_permlane32_swap to the first register when the operands differ -- it emitted lo' + lo' -- so the
//  builtin is only used with identical operands, in xor_lane.)
template <int CH, int NA, int NN, int MASK>
__device__ __forceinline__ void rs_stage(float (&A)[CH][NA], int bit) {
#pragma unroll
    for (int i = 0; i < NN / 2; ++i) {
#pragma unroll
        for (int d = 0; d < NA; ++d) {
            const float lo = A[i][d], hi = A[i + NN / 2][d];
            const float send = bit ? lo : hi;
            const float keep = bit ? hi : lo;
            A[i][d] = keep + xor_lane<MASK>(send);
        }
    }
}

// lane-id bit of the s-th butterfly stage: the quad-index bits are lane bits 2..5; with HS = 2 the bit
// that selects the chain half (bit log2(M)) is skipped -- the halves walk different forward particles.
template <int M, int HS>
__device__ __host__ constexpr int stage_bit(int s) {
    int hb = 0;
    while ((1 << hb) < M) ++hb;
    int k = -1;
    for (int bit = 2; bit <= 5; ++bit) {
        if (HS == 2 && bit == hb) continue;
        if (++k == s) return bit;
    }
    return 5;
}

inline namespace PSVO_LNS {   // l1 / l2: hidden layers of the per-particle MLPs (common.h)
// HS as in bsim_fwd.hip: HS = 2 spreads a chain over 2*M lanes (half the forward particles and half the
// hidden units of every MLP per lane) so that small problems still run two waves per SIMD.
template <int DX, int DY, int H, int M, int CH, int HS>
__global__ void __launch_bounds__(256, HS) bsim_bwd_kernel(const BsimBwdArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    using AC = BAcc<DX, DY>;
    constexpr int PS = BTileSlot<DX>::kFloats;
    constexpr bool kRolled = true;
    constexpr int NA = DX + 1;  // per-j accumulators: dF (DX) and dW
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTB = blockDim.x, nwv = NTB >> 6;
    const int B = a.B, T = a.T, N = a.N;
    // forward tile padded (W' = -inf: zero weight) to whole chunks of CH entries per quad lane and half, so the
    // walk needs no per-entry predication and the LDS reads of a chunk are issued back to back
    constexpr int kPad = 4 * CH * HS;
    const int NP = ((N + kPad - 1) / kPad) * kPad;
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    constexpr int G = M * HS;
    constexpr int NS = (HS == 2) ? 3 : 4;  // butterfly stages (quads that share a forward-particle range)
    const int cpb = NTB / G;
    const int cl = tid / G, hpart = (tid % G) / M, m = tid % M, q = m & 3;
    const bool h0 = (hpart == 0);
    const int n_raw = blk * cpb + cl;
    const bool valid = n_raw < N;
    const int n = valid ? n_raw : N - 1;
    const int gbase = lane - m;

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;               // [2][NP][PS]
    float* jacc = tile + 2 * NP * PS;            // [nwv][NA][NP] wave-private d F' / d W accumulators
    float* red = jacc + nwv * NA * NP;           // [cpb][DX] + 16

    MQ::load(wf, a.f, tid, NTB);
    MG::load(wg, a.g, tid, NTB);
    MQ::load(wqi, a.q1inv, tid, NTB);
    for (int i = tid; i < nwv * NA * NP; i += NTB) jacc[i] = 0.f;

    // ---- constants ----------------------------------------------------------------------------
    const float kappa = sqrtf(0.5f * kLog2e);  // rho_d * sigma_d
    float sf[DX], isf[DX], rp[DX], isfk[DX], isg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        sf[d] = a.sig_f[d];
        isf[d] = 1.f / sf[d];
        rp[d] = isf[d] * kappa;
        isfk[d] = isf[d] / kappa;      // (divisions by loop constants are hoisted: an IEEE division is ~10 VALU instructions)
    }
    float ikap2 = 1.f / (kappa * kappa);
    keep_in_vgpr(ikap2);
#pragma unroll
    for (int e = 0; e < DY; ++e) isg[e] = 1.f / a.sig_g[e];
    float pc[DX], pic[DX], pi1[DX], pi2[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        pi1[d] = 1.f / a.sig_q1inv[d];
        pi2[d] = 1.f / a.sig_bq2[d];
        pic[d] = pi1[d] + pi2[d];
        pc[d] = 1.f / pic[d];
    }
    float s_init[DX], is_init[DX], i_isig[DX], im[DX], mi[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        s_init[d] = a.sig_init[d];
        is_init[d] = 1.f / s_init[d];
        i_isig[d] = 1.f / a.isig[d];
        im[d] = a.imean[b * DX + d];
        mi[d] = a.minit[b * DX + d];
    }
    const float ninf = -__builtin_huge_valf();
    const float aw = valid ? a.dscore[(size_t)b * N + n] : 0.f;  // d loss / d score of this chain
    keep_in_vgpr(sf); keep_in_vgpr(isf); keep_in_vgpr(rp); keep_in_vgpr(isfk); keep_in_vgpr(isg);
    keep_in_vgpr(pc); keep_in_vgpr(pic); keep_in_vgpr(pi1); keep_in_vgpr(pi2);
    keep_in_vgpr(s_init); keep_in_vgpr(is_init); keep_in_vgpr(i_isig); keep_in_vgpr(im); keep_in_vgpr(mi);

    // ---- forward-tile staging (identical image to the forward kernel) -----------------------------------
    // raw values are loaded at the top of a step and only scaled / written to LDS at its end, so that nothing
    // in between waits for the global loads.  Only the first NTB tile slots are prefetched through registers
    // (N <= 256: the latency-bound regime); the remaining slots of a larger tile are copied at store time.
    float st[DX + 1], st_l = 0.f;
    int st_t = 0;
    auto put_slot = [&](float* buf, int j, const float (&raw)[DX + 1], float l) {
        float F[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) F[d] = raw[d] * rp[d];
        const float W = j < N ? (raw[DX] - l) * kLog2e : ninf;
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
    };
    auto get_slot = [&](int tt, int j, float (&raw)[DX + 1]) {
        const size_t tb = (size_t)tt * B + b;
        const int jc = j < N ? j : N - 1;
#pragma unroll
        for (int d = 0; d < DX; ++d) raw[d] = a.Fm[(tb * DX + d) * N + jc];
        raw[DX] = a.logW[tb * N + jc];
    };
    auto stage_load = [&](int tt) {
        st_t = tt;
        st_l = a.lse[(size_t)tt * B + b];
        if (tid < NP) get_slot(tt, tid, st);
    };
    auto stage_store = [&](float* buf) {
        if (tid < NP) put_slot(buf, tid, st, st_l);
        for (int j = tid + NTB; j < NP; j += NTB) {
            float raw[DX + 1];
            get_slot(st_t, j, raw);
            put_slot(buf, j, raw, st_l);
        }
    };
    // step t reads forward tile t-1; first tile needed is tile(0) at t = 1
    if (T >= 2) {
        stage_load(0);
        stage_store(tile);
    }
    __syncthreads();

    float acc[AC::kN];
#pragma unroll
    for (int i = 0; i < AC::kN; ++i) acc[i] = 0.f;
    float dX[DX];  // d loss / d bwX_t of this chain (all M lanes hold the same value)
#pragma unroll
    for (int d = 0; d < DX; ++d) dX[d] = 0.f;

    // per-step inputs, requested one step ahead of their use
    struct StepIn {
        float eps[DX], bm[DX], xp[DX], mu1[DX], y[DY], om, lam2;
        int sel;
    };
    auto load_step = [&](int t, StepIn& s) {
        const size_t tb = (size_t)t * B + b;
        const bool lst = (t == T - 1);
        s.sel = a.sel[tb * N + n];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            s.eps[d] = a.eps_b[((tb * DX + d) * N + n) * M + m];
            s.bm[d] = a.bmu2[tb * DX + d];
            s.xp[d] = lst ? 0.f : a.bwX[((tb + B) * DX + d) * N + n];
            s.mu1[d] = lst ? 0.f : a.mu1_all[(tb * DX + d) * N + n];
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) s.y[k] = a.obs[tb * DY + k];
        s.om = a.om_all[(tb * N + n) * M + m];
        s.lam2 = t >= 1 ? a.lam2_all[(tb * N + n) * M + m] : 0.f;
    };
    StepIn cur_in, nxt_in;
    load_step(0, cur_in);

    const int nqh = (NP >> 2) / HS;       // forward-tile entries per quad lane and chain half (multiple of CH)
    const int e0 = hpart * nqh, e1 = e0 + nqh;

    SEC_INIT(bsim_bwd)
    // With one wave per SIMD (HS = 1) the first (t = 0: prior term instead of the tile pass) and last (t = T-1: no
    // successor) steps are peeled out of the loop as separate instantiations of the step body, so that the T-2 steps in
    // between carry no first / last branches (-15 % VALU instructions per step; C4 / C5 +3 %).  The half-split kernel
    // (HS = 2, two waves per SIMD) measured SLOWER peeled (1.59 -> 1.87 ms at C*) and keeps the plain loop.
    auto step = [&](auto first_tag, auto last_tag, const int t) {
        SEC(0);
        const size_t tb = (size_t)t * B + b;
        const bool first = first_tag, last = last_tag;     // compile-time constants when the step is peeled
        const float* cur = tile + ((t + 1) & 1) * NP * PS;  // tile(t-1), valid for t >= 1
        float* nxt = tile + (t & 1) * NP * PS;              // tile(t) for step t+1
        if (t + 1 < T && t >= 1) stage_load(t);             // (tile(0) was staged in the prologue)

        // ---- recompute the proposal ---------------------------------------------------------------------
        if (t + 1 < T) load_step(t + 1, nxt_in);
        SEC(1);   // issue of the prefetch loads
        float xp[DX], eps[DX], bm[DX], mu1[DX], mu[DX], x[DX], y[DY];
        const int sel = cur_in.sel;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            eps[d] = cur_in.eps[d];
            bm[d] = cur_in.bm[d];
            xp[d] = cur_in.xp[d];
            mu1[d] = cur_in.mu1[d];
            if (!last) {
                mu[d] = pc[d] * fmaf(pi1[d], mu1[d], pi2[d] * bm[d]);
                x[d] = fmaf(pc[d], eps[d], mu[d]);
            } else {
                mu[d] = mi[d];
                x[d] = fmaf(s_init[d], eps[d], mu[d]);
            }
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = cur_in.y[k];
        const float pi_m = valid ? exp2_fast(cur_in.om * kLog2e) : 0.f;
        const float issel = (m == sel) ? 1.f : 0.f;
        const float dphi = aw * pi_m;                 // = d g_m = d iota_m
        const float dlam = -aw * (issel - pi_m);

        float dxt[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dxt[d] = h0 ? issel * dX[d] : 0.f;  // dxt: this lane's PARTIAL of d x~_m

        SEC(2);   // proposal recompute, coefficients
        // ---- filter term: second pass over the forward tile -------------------------------------------------
#ifdef PSVO_EXP_NOPAIR
        if (false) {
#else
        if (!first) {
#endif
            const float lam2 = cur_in.lam2;
            // packed f32: sub-particles (0,1) and (2,3) of the quad share every v_pk_* instruction
            float lq[4], dl[4];
            quad_bcast4(lam2, lq);
            quad_bcast4(dlam, dl);
            const f2 lqa = f2{lq[0], lq[1]}, lqb = f2{lq[2], lq[3]};
            const f2 dla = f2{dl[0], dl[1]}, dlb = f2{dl[2], dl[3]};
            f2 xa[DX], xb[DX], Ua[DX], Ub[DX], Va[DX], Vb[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                float t4[4];
                quad_bcast4(x[d] * rp[d], t4);
                xa[d] = f2{t4[0], t4[1]};
                xb[d] = f2{t4[2], t4[3]};
                Ua[d] = Ub[d] = Va[d] = Vb[d] = f2{0.f, 0.f};
            }
            float* ja = jacc + wave * NA * NP;
            // The walk over j is done in chunks of CH entries kept in registers; after each chunk the
            // per-j partial sums (over this quad's four m) are reduce-scattered across the 16 quads
            // of the wave (lane bits 2..5) with a 4-stage butterfly, so every lane ends up owning
            // CH/16 fully reduced (j, d) sums which it stores -- no atomics, fixed summation order.
            for (int c0 = e0; c0 < e1; c0 += CH) {
                float A[CH][NA];
#pragma unroll
                for (int i2 = 0; i2 < CH; ++i2) {
                    {
                        const int j = (c0 + i2) * 4 + q;
                        float F[DX], W;
                        read_slot<DX>(cur + j * PS, F, W);
                        f2 ua[DX], ub[DX], la = f2{W, W}, lb = la;
#pragma unroll
                        for (int d = 0; d < DX; ++d) {
                            const f2 Fd = f2{F[d], F[d]};
                            ua[d] = xa[d] - Fd;
                            ub[d] = xb[d] - Fd;
                            la = pk_fma(-ua[d], ua[d], la);
                            lb = pk_fma(-ub[d], ub[d], lb);
                        }
                        la -= lqa;
                        lb -= lqb;
                        const f2 pa = f2{exp2_fast(la.x), exp2_fast(la.y)}, pb = f2{exp2_fast(lb.x), exp2_fast(lb.y)};
                        const f2 ca = dla * pa, cb = dlb * pb;
                        const f2 cs = ca + cb;
                        A[i2][DX] = cs.x + cs.y;
#pragma unroll
                        for (int d = 0; d < DX; ++d) {
                            const f2 pua = pa * ua[d], pub = pb * ub[d];
                            Ua[d] += pua;
                            Ub[d] += pub;
                            Va[d] = pk_fma(pua, ua[d], Va[d]);
                            Vb[d] = pk_fma(pub, ub[d], Vb[d]);
                            const f2 ad = pk_fma(cb, ub[d], ca * ua[d]);
                            A[i2][d] = ad.x + ad.y;
                        }
                    }
                }
                constexpr int b0 = stage_bit<M, HS>(0), b1 = stage_bit<M, HS>(1), b2 = stage_bit<M, HS>(2);
                rs_stage<CH, NA, CH, (1 << b0)>(A, (lane >> b0) & 1);
                rs_stage<CH, NA, CH / 2, (1 << b1)>(A, (lane >> b1) & 1);
                rs_stage<CH, NA, CH / 4, (1 << b2)>(A, (lane >> b2) & 1);
                int ebase = ((lane >> b0) & 1) * (CH / 2) + ((lane >> b1) & 1) * (CH / 4) + ((lane >> b2) & 1) * (CH / 8);
                if constexpr (NS == 4) {
                    constexpr int b3 = stage_bit<M, HS>(3);
                    rs_stage<CH, NA, CH / 8, (1 << b3)>(A, (lane >> b3) & 1);
                    ebase += ((lane >> b3) & 1) * (CH / 16);
                }
                constexpr int R = CH >> NS;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = c0 + ebase + r;
#pragma unroll
                    for (int d = 0; d < NA; ++d) ja[d * NP + e * 4 + q] = A[r][d];
                }
            }
            SEC(3);   // pair loop + butterflies
            // merge the quad's four j-slices; lane q keeps sub-particle i == q
            float Uo[DX], Vo[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                Uo[d] = 0.f;
                Vo[d] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    float u = (i == 0) ? Ua[d].x : (i == 1) ? Ua[d].y : (i == 2) ? Ub[d].x : Ub[d].y;
                    float v = (i == 0) ? Va[d].x : (i == 1) ? Va[d].y : (i == 2) ? Vb[d].x : Vb[d].y;
                    u = group_sum<4>(u);
                    v = group_sum<4>(v);
                    if (i == q) {
                        Uo[d] = u;
                        Vo[d] = v;
                    }
                }
            }
            // (x~-F)/sigma^2 = u / (sigma kappa);  z^2 = u^2 / kappa^2
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                // U, V cover this half's forward particles only: partial sums; the "-1" is counted once
                dxt[d] -= dlam * Uo[d] * isfk[d];
                acc[AC::kSf + d] += dlam * (Vo[d] * ikap2 - (h0 ? 1.f : 0.f)) * isf[d];
            }
        } else {
            // t = 0: iota_m = LN(x~; imean, isig)   (reference PSVO.py:169-175)
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float z = (x[d] - im[d]) * i_isig[d];
                const float tz = dphi * z * i_isig[d];
                if (h0) {
                    dxt[d] -= tz;
                    acc[AC::kSiota + d] += dphi * (z * z - 1.f) * i_isig[d];
                }
            }
        }

        SEC(4);   // quad merges
        // ---- f(x_{t+1} | x~), g(y_t | x~) ------------------------------------------------------------------------
        float dxp_part[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dxp_part[d] = 0.f;
        {
            float dFo[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) dFo[d] = 0.f;
            if (!last) {
                float fmx[DX];
                if constexpr (HS == 1) {
                    MQ::template eval<kRolled>(wf, x, fmx);
                } else {
                    MQ::template eval_part<HS, ilog2(M)>(wf, hpart, x, fmx);
#pragma unroll
                    for (int d = 0; d < DX; ++d) fmx[d] += xor_lane<M>(fmx[d]);
                }
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    const float z = (xp[d] - fmx[d]) * isf[d];
                    dFo[d] = dphi * z * isf[d];
                    if (h0) {
                        dxp_part[d] = -dFo[d];
                        acc[AC::kSf + d] += dphi * (z * z - 1.f) * isf[d];
                    }
                }
                if constexpr (HS == 1) MQ::template bwd_input<kRolled>(wf, x, dFo, dxt);
                else MQ::template bwd_input_part<HS, ilog2(M)>(wf, hpart, x, dFo, dxt);
            }
            if (valid && h0) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    a.dFt[((tb * DX + d) * N + n) * M + m] = dFo[d];
                    a.xt[((tb * DX + d) * N + n) * M + m] = x[d];
                }
            }
            float gm[DY], dGo[DY];
            if constexpr (HS == 1) {
                MG::template eval<kRolled>(wg, x, gm);
            } else {
                MG::template eval_part<HS, ilog2(M)>(wg, hpart, x, gm);
#pragma unroll
                for (int k = 0; k < DY; ++k) gm[k] += xor_lane<M>(gm[k]);
            }
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                float dmean = 1.f;
                if (a.emission) { dmean = emis_dmean(gm[k]); gm[k] = emis_mean(gm[k]); }
                const float z = (y[k] - gm[k]) * isg[k];
                dGo[k] = dphi * z * isg[k] * dmean;
                if (h0) acc[AC::kSg + k] += dphi * (z * z - 1.f) * isg[k];
                if (valid && h0) a.dGt[((tb * DY + k) * N + n) * M + m] = dGo[k];
            }
            if constexpr (HS == 1) MG::template bwd_input<kRolled>(wg, x, dGo, dxt);
            else MG::template bwd_input_part<HS, ilog2(M)>(wg, hpart, x, dGo, dxt);
        }

        SEC(5);   // MLP_f / MLP_g forward + input gradients, row stores
        // ---- reduce over the chain's M sub-particles ------------------------------------------------------------
        float dmu[DX], sce[DX], dxp[DX], dim[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            float v0 = dxt[d], v1 = dxt[d] * eps[d], v2 = dxp_part[d];
            float v3 = (first && h0) ? dphi * (x[d] - im[d]) * i_isig[d] * i_isig[d] : 0.f;
            // over the M sub-particles and the HS halves
            dmu[d] = group_sum<G>(v0);
            sce[d] = group_sum<G>(v1);
            dxp[d] = group_sum<G>(v2);
            dim[d] = first ? group_sum<G>(v3) : 0.f;
        }
        const bool lead = (m == 0) && h0 && valid;
        float outv[DX];  // per-chain value that is summed over the workgroup: d bmu2 / d minit
        // The lead lane's sums are selects, not an `if`, and its d mu1 row is stored with the other per-chain rows BEHIND the
        // MLP pass below: with a divergent region right in front of that pass hipcc 7.2 put the VGPR -> AGPR copies of a
        // live-range split of acc[] at the region's end, ahead of the EXEC restore, i.e. under the lead-only mask -- every
        // other lane then lost what it had added to acc[] since the last copy (tools/exec_restore_check.py).
        float dmu1[DX];
        if (!last) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                dmu1[d] = dmu[d] * pc[d] * pi1[d];
                outv[d] = dmu[d] * pc[d] * pi2[d];
                acc[AC::kSc + d] += lead ? sce[d] + aw * pic[d] : 0.f;   // -sum_m dq_m / c = a / c
                acc[AC::kSmm1 + d] += lead ? dmu[d] * mu1[d] : 0.f;
                acc[AC::kSmb + d] += lead ? dmu[d] * bm[d] : 0.f;
                acc[AC::kSmm + d] += lead ? dmu[d] * mu[d] : 0.f;
            }
            // MLP_q1inv's input is the same in all G lanes of the chain: spread its hidden units over kQS of them
            {
                constexpr int kQS = (H / 4 < G) ? H / 4 : G;
                float dq[DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) dq[d] = 0.f;
                MQ::template bwd_input_part<kQS>(wqi, (tid % G) & (kQS - 1), xp, dmu1, dq);
#pragma unroll
                for (int d = 0; d < DX; ++d) dxp[d] += group_sum<kQS>(dq[d]);
            }
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                outv[d] = dmu[d];
                dmu1[d] = 0.f;
                acc[AC::kSinit + d] += lead ? sce[d] + aw * is_init[d] : 0.f;
            }
        }
#pragma unroll
        for (int d = 0; d < DX; ++d) dX[d] = dxp[d];

        SEC(6);   // chain reductions, MLP_q1inv input gradient
        // ---- workgroup reductions: d bmu2[t] / d minit, d imean; flush d Fm / d logW partials ---------------------------
        // per-chain rows of d bmu2[t] / d minit / d imean: summed over the chains by the host afterwards (a
        // workgroup sum here would add an LDS round trip to every step of the serial chain)
        if (lead) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                a.dmu1[(tb * DX + d) * N + n] = dmu1[d];
                a.dbmu2_rows[(tb * DX + d) * N + n] = last ? 0.f : outv[d];
                if (last) a.dminit_rows[((size_t)b * DX + d) * N + n] = outv[d];
                if (first) a.dimean_rows[((size_t)b * DX + d) * N + n] = dim[d];
            }
        }
        __syncthreads();
        if (!first) {
            // forward step t-1 receives d F (un-prescale: d F = d F' * rho, and the 1/(sigma kappa) factor)
            const size_t tbm = tb - B;
            for (int i = tid; i < NA * N; i += NTB) {
                const int d = i / N, j = i - d * N;
                const float* col = jacc + d * NP + j;
                float s;
                if (nwv == 4) {               // the usual 256-lane workgroup: four independent reads
                    const float a0 = col[0], a1 = col[NA * NP], a2 = col[2 * NA * NP], a3 = col[3 * NA * NP];
                    s = (a0 + a1) + (a2 + a3);
                } else {
                    s = 0.f;
                    for (int w = 0; w < nwv; ++w) s += col[w * NA * NP];
                }
                if (d < DX) a.dFm_part[((tbm * nblk + blk) * DX + d) * N + j] = s * isfk[d];
                else a.dlogW_part[(tbm * nblk + blk) * N + j] = s;
            }
        }
        if (last) {
            for (int i = tid; i < NA * N; i += NTB) {
                const int d = i / N, j = i - d * N;
                if (d < DX) a.dFm_part[((tb * nblk + blk) * DX + d) * N + j] = 0.f;
                else a.dlogW_part[(tb * nblk + blk) * N + j] = 0.f;
            }
        }
        SEC(7);   // barrier, workgroup sums, d Fm / d logW flush
        if (t + 1 < T && t >= 1) stage_store(nxt);
        cur_in = nxt_in;
        __syncthreads();
        SEC(8);   // tile store + barrier
    };
    if constexpr (HS == 1) {
        step(std::true_type{}, std::false_type{}, 0);
        for (int t = 1; t < T - 1; ++t) step(std::false_type{}, std::false_type{}, t);
        step(std::false_type{}, std::true_type{}, T - 1);
    } else {
        for (int t = 0; t < T; ++t) step(t == 0, t == T - 1, t);
    }

    // ---- scalar accumulators: reduce over the workgroup ---------------------------------------------------------------
    for (int i = 0; i < AC::kN; ++i) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < AC::kN; ++k) v = (k == i) ? acc[k] : v;
        v = wave_sum(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < nwv; ++w) s += red[w];
            a.sacc_part[((size_t)b * nblk + blk) * AC::kN + i] = s;
        }
        __syncthreads();
    }
}
}  // inline namespace PSVO_LNS

// After the reverse kernel, ONE launch: blocks 0 .. gridDim.x - 2 fold the per-workgroup partials of d Fm / d logW (what the
// filter's reverse pass waits for) in workgroup order; the last block folds the (B * nblk, NACC) partial sums into the scale
// gradients (see filter_bwd_finalize for the PoG algebra), one accumulator per wave at a time.
template <int DX, int DY>
__global__ void __launch_bounds__(256) bsim_bwd_fold_finalize(
    const float* __restrict__ dFm_part, const float* __restrict__ dlogW_part, long long TB, int nblk, int N,
    float* __restrict__ dFm, float* __restrict__ dlogW, const float* __restrict__ sacc, int rows, const float* sig_q1inv,
    const float* sig_bq2, float* dsig_f, float* dsig_g, float* dsig_q1inv, float* dsig_bq2, float* dsig_init,
    float* disig, float* __restrict__ dlse, unsigned nfold) {
    using AC = BAcc<DX, DY>;
    if (blockIdx.x >= nfold && blockIdx.x + 1 < gridDim.x) {
        // d lse[t, b] = -sum_j d logW^[t, b, j]: the normaliser's share of the filter term's gradient (W^ = logW - lse).  It is
        // zero analytically (sum_j d W^_j = sum_m d Lambda_m = 0 per chain) and ~1e-7 of the terms in fp32; TF's autodiff --
        // like any autodiff of the reference's graph -- carries it, and it is what keeps the filter's reverse pass
        // well-conditioned when the forward particles have left the data range (the softmax-weighted subtraction in
        // psvo_filter_backward then cancels the common part of d logW_j exactly instead of to 1e-7: measured on
        // tests/golden/fhn_illconditioned_state.npz, the emission network's gradient error drops 100 x to the fp32 oracle's).
        // One wave per (t, b) row, straight from the per-workgroup partials, concurrently with the fold blocks.
        const long long tb = (long long)(blockIdx.x - nfold) * 4 + (threadIdx.x >> 6);
        if (tb < TB) {
            const float* p = dlogW_part + tb * nblk * N;
            float v = 0.f;
            for (int i = threadIdx.x & 63; i < nblk * N; i += 64) v += p[i];
            v = wave_sum(v);
            if ((threadIdx.x & 63) == 0) dlse[tb] = -v;
        }
        return;
    }
    if (blockIdx.x + 1 < gridDim.x) {
        const long long rowF = (long long)DX * N, nF = TB * rowF, nW = TB * N;
        long long e = (long long)blockIdx.x * 256 + threadIdx.x;
        if ((N & 3) == 0) {      // four outputs per thread (the launcher sized the grid for it): 16-byte loads
            e *= 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < nF) {
                const long long tb = e / rowF;
                const float* p = dFm_part + tb * nblk * rowF + (e - tb * rowF);
                for (int k = 0; k < nblk; ++k) {
                    const float4 q = *reinterpret_cast<const float4*>(p + k * rowF);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                *reinterpret_cast<float4*>(dFm + e) = v;
            } else if (e < nF + nW) {
                e -= nF;
                const long long tb = e / N;
                const float* p = dlogW_part + tb * nblk * N + (e - tb * N);
                for (int k = 0; k < nblk; ++k) {
                    const float4 q = *reinterpret_cast<const float4*>(p + (long long)k * N);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                *reinterpret_cast<float4*>(dlogW + e) = v;
            }
            return;
        }
        if (e < nF) {            // d Fm (T, B, Dx, N) from (T, B, nblk, Dx, N)
            const long long tb = e / rowF;
            const float* p = dFm_part + tb * nblk * rowF + (e - tb * rowF);
            float v = 0.f;
            for (int k = 0; k < nblk; ++k) v += p[k * rowF];
            dFm[e] = v;
        } else if (e < nF + nW) {   // d logW (T, B, N) from (T, B, nblk, N)
            e -= nF;
            const long long tb = e / N;
            const float* p = dlogW_part + tb * nblk * N + (e - tb * N);
            float v = 0.f;
            for (int k = 0; k < nblk; ++k) v += p[(long long)k * N];
            dlogW[e] = v;
        }
        return;
    }
    __shared__ float tot[AC::kN];
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
    for (int k = wave; k < AC::kN; k += 4) {  // lanes stride over the (sequence, workgroup) rows
        float v = 0.f;
        for (int r = lane; r < rows; r += 64) v += sacc[(size_t)r * AC::kN + k];
        v = wave_sum(v);
        if (lane == 0) tot[k] = v;
    }
    __syncthreads();
    auto total = [&](int k) { return tot[k]; };
    if (d < DX) {
        const float i1 = 1.f / sig_q1inv[d], i2 = 1.f / sig_bq2[d];
        const float c = 1.f / (i1 + i2);
        const float dc = total(AC::kSc + d) + total(AC::kSmm + d) / c;
        const float di1 = c * total(AC::kSmm1 + d) - c * c * dc;
        const float di2 = c * total(AC::kSmb + d) - c * c * dc;
        dsig_q1inv[d] = -i1 * i1 * di1;
        dsig_bq2[d] = -i2 * i2 * di2;
        dsig_f[d] = total(AC::kSf + d);
        dsig_init[d] = total(AC::kSinit + d);
        disig[d] = total(AC::kSiota + d);
    }
    if (d < DY) dsig_g[d] = total(AC::kSg + d);
}

struct BsimBwdOut {
    float *dsig_f, *dsig_g, *dsig_q1inv, *dsig_bq2, *dsig_init, *disig;
    float *dFm, *dlogW;     // folded over the workgroups (the reverse kernels write a.dFm_part / a.dlogW_part)
    float* dlse;            // (T,B) or NULL: -sum_j d logW (see bsim_bwd_fold_finalize)
};

template <int DX, int DY>
static inline void launch_bsim_fold_finalize(const BsimBwdArgs& a, const BsimBwdOut& o, int nblk, hipStream_t stream) {
    const long long TB = (long long)a.T * a.B;
    const long long n = TB * (DX + 1) * a.N / ((a.N & 3) == 0 ? 4 : 1);   // (threads: four outputs each when N % 4 == 0)
    const unsigned nfold = (unsigned)((n + 255) / 256), nrow = o.dlse ? (unsigned)((TB + 3) / 4) : 0u;
    hipLaunchKernelGGL((bsim_bwd_fold_finalize<DX, DY>), dim3(nfold + nrow + 1), dim3(256), 0, stream,
                       a.dFm_part, a.dlogW_part, TB, nblk, a.N, o.dFm, o.dlogW, a.sacc_part, a.B * nblk, a.sig_q1inv,
                       a.sig_bq2, o.dsig_f, o.dsig_g, o.dsig_q1inv, o.dsig_bq2, o.dsig_init, o.disig, o.dlse, nfold);
}

static inline void bsim_geometry(int B, int N, int M, int H, int Dx, int& HS, int& NTB, int& cpb, int& nblk, int layers = 1) {
    // fewer than two waves per SIMD (1024 SIMDs) with one lane per (chain, m): spread a chain over 2M lanes.
    // Only for Dx <= 2: two waves per SIMD need <= 256 VGPRs, which the Dx >= 3 reverse kernel exceeds
    // (measured: 264-1056 B/lane of scratch when forced), and a spilling kernel is slower than a lone wave.
    const long long waves1 = ((long long)B * N * M + 63) / 64;
    HS = (waves1 < 2048 && 2 * M <= 64 && (H / 2) % 4 == 0 && Dx <= 2 && (layers != 2 || g_tune_l2_split)) ? 2 : 1;
    NTB = ((N * M * HS + 63) / 64) * 64;
    if (NTB > 256) NTB = 256;
    cpb = NTB / (M * HS);
    nblk = (N + cpb - 1) / cpb;
}

inline namespace PSVO_LNS {
template <int DX, int DY, int H, int M>
static int launch_bsim_bwd(const BsimBwdArgs& a, const BsimBwdOut& o, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    using AC = BAcc<DX, DY>;
    constexpr int PS = BTileSlot<DX>::kFloats;
    int HS, NTB, cpb, nblk;
    bsim_geometry(a.B, a.N, M, H, DX, HS, NTB, cpb, nblk, PSVO_L);
    const int nwv = NTB / 64;
    // chunk of forward-tile entries reduced in registers per butterfly: 32 when a lane walks >= 32 entries
    const int walk = ((a.N + 3) / 4 + HS - 1) / HS;
    const int CHs = (HS == 2) ? 8 : (walk >= 32 ? 32 : 16);
    const int pad = 4 * CHs * HS;
    const int NP = ((a.N + pad - 1) / pad) * pad;     // as in the kernel
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * PS + (size_t)nwv * (DX + 1) * NP +
                                        2 * cpb * DX + 16);
    if (lds > 160 * 1024) return PSVO_ERR_UNSUPPORTED;
    clear_hip_error();
    if (HS == 2) {
        // chunks of 8 entries keep the half-split kernel at 251 VGPRs (two waves per SIMD, no scratch)
        if constexpr (2 * M <= 64 && (H / 2) % 4 == 0 && DX <= 2)
            hipLaunchKernelGGL((bsim_bwd_kernel<DX, DY, H, M, 8, 2>), dim3(nblk, a.B), dim3(NTB), lds, stream, a);
    } else if (walk >= 32) {
        hipLaunchKernelGGL((bsim_bwd_kernel<DX, DY, H, M, 32, 1>), dim3(nblk, a.B), dim3(NTB), lds, stream, a);
    } else {
        hipLaunchKernelGGL((bsim_bwd_kernel<DX, DY, H, M, 16, 1>), dim3(nblk, a.B), dim3(NTB), lds, stream, a);
    }
    (void)o;
    (void)AC::kN;
    return launch_status();
}

template <int DX, int DY, int H>
static int bb_dispatch_m(const BsimBwdArgs& a, const BsimBwdOut& o, int M, hipStream_t s) {
    switch (M) {
        case 4: return launch_bsim_bwd<DX, DY, H, 4>(a, o, s);
        case 8: return launch_bsim_bwd<DX, DY, H, 8>(a, o, s);
        case 16: return launch_bsim_bwd<DX, DY, H, 16>(a, o, s);
        case 32: return launch_bsim_bwd<DX, DY, H, 32>(a, o, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX, int DY>
static int bb_dispatch_h(const BsimBwdArgs& a, const BsimBwdOut& o, int H, int M, hipStream_t s) {
    switch (H) {
#if PSVO_L == 1   // (two hidden layers: widths 32 and 64; narrower ones are zero-padded upstream)
        case 16: return bb_dispatch_m<DX, DY, 16>(a, o, M, s);
#endif
        case 32: return bb_dispatch_m<DX, DY, 32>(a, o, M, s);
        case 64: return bb_dispatch_m<DX, DY, 64>(a, o, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

// external linkage: explicitly instantiated once per DX in bsim_bwd_dx{2,3,4}.hip (compiled in parallel)
template <int DX>
int bb_dispatch_dy(const BsimBwdArgs& a, const BsimBwdOut& o, int Dy, int H, int M, hipStream_t s) {
    switch (Dy) {
        case 1: return bb_dispatch_h<DX, 1>(a, o, H, M, s);
        case 2: return bb_dispatch_h<DX, 2>(a, o, H, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
}  // inline namespace PSVO_LNS

}  // namespace psvo

