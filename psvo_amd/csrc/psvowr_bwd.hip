// Reverse mode of psvo_bsimwr_forward (PSVOwR, reference src/SMC/PSVOwR.py:65-198 under TensorFlow autodiff).
// As in the forward pass a sequence is owned by a cluster of K persistent workgroups (cooperative launch):
// workgroup k walks t = 0 .. T-1 over the (chain, sub-particle) items of its own chains, publishes
// d loss / d bwXanc_{t+1} of those chains to HBM and meets the others at one barrier per step; the next step
// scatter-adds the published gradients of ALL chains into the parents it owns.  With a = d loss / d bw_log_W[t, chain], pi_m = exp(omega_m), sel the drawn sub-particle and
//     bw_log_W = logsumexp_m(omega_raw) - phi_sel - log M,    omega_raw = Lambda + phi + g - q
// the coefficients are
//     d Lambda_m (d iota_m at t = 0) = d g_m = a pi_m,   d phi_m = a (pi_m - delta_{m,sel}),   d q_m = -a pi_m.
// Unlike PSVO, sum_m d Lambda_m = a != 0, so the normalisation of the forward weights does receive gradient:
// d lse[t-1] = - sum_j d W^_j is written for psvo_filter_backward.
// The cross-chain resampling (bwXanc_t[k] = bwX_t[anc_t[k]]) back-propagates as a scatter-add of
// d bwXanc_t into the selected sub-particle of the parent chain (fixed summation order, N*Dx per step).
// The forward tile is recomputed (second pass) exactly as in bsim_bwd_impl.h: per-j partial sums are
// reduce-scattered over the 16 quads of a wave with a butterfly, folded over waves through LDS and written as
// per-workgroup partials; sums over chains (d bmu2, d minit, d imean) are left to the host as per-chain rows.
#include "common.h"

namespace psvo {
inline namespace PSVO_LNS {   // l1 / l2: hidden layers of the per-particle MLPs (common.h)

PSVO_TIMERS_DEFINE(psvowr_bwd)

struct WrBwdArgs {
    int B, T, N, emission;
    psvo_mlp f, g, q1inv;
    const float *Fm, *logW, *lse;
    const float *sig_f, *sig_g, *sig_q1inv, *sig_bq2;
    const float *bmu2, *minit, *sig_init, *imean, *isig;
    const float *obs, *eps_b;
    const float *bwXanc, *bwW, *lseW;
    const int32_t *sel, *anc;
    const float *lam2_all, *om_all, *mu1_all;
    const float* dlseW;  // (T,B)
    float *xt, *dFt, *dGt, *dmu1;
    float *dFm_part, *dlogW_part, *dlse_part;   // (T,B,K,Dx,N), (T,B,K,N), (T,B,K): per-workgroup partials
    float *dbmu2_rows, *dminit_rows, *dimean_rows;   // (T,B,Dx,N), (B,Dx,N), (B,Dx,N): per-chain rows
    float* sacc;                                 // (B,K,NACC)
    unsigned long long* ring;                    // exchange ring: d loss / d bwXanc_t of every chain, tagged
    unsigned* prog;                              // [B][kWbProg] steps consumed by each workgroup of the cluster
    unsigned* err;                               // error flag: a poll timed out
};

// words a chain publishes per step (d loss / d bwXanc, Dx <= 4) and the exchange workspace: a ring of D = min(T, 16)
// slots [D][B][N][kWbWords] of tagged 64-bit words {bits(value), tag = step + 1}, then a progress word per workgroup
// [B][8] and two 32-bit words whose last one is the error flag.
// The forward kernel gets away with a two-slot ring and no flow control because there every workgroup polls all N chains
// every step, which bounds the run-ahead of any workgroup to one step.  Here a workgroup polls only the chains whose parent
// it owns; under weight degeneracy most workgroups own no parent for many steps, nothing holds them back, and their
// publication for step t+1 could overwrite words of an earlier step that a slower workgroup has not polled yet.  Hence
// BACK-PRESSURE: after it has consumed the words of step t a workgroup publishes progress = t + 1; before a workgroup
// writes into slot (t+1) mod D it makes sure every member of its cluster has consumed that slot's previous occupant (step
// t+1-D).  The progress words are re-read only when the cached minimum no longer covers the step -- about every D - 2
// steps in a balanced run -- so the common case costs nothing, and the ring (2 MB at C*) stays in L2 where a slot per
// step (26 MB, the first fix of round 2) was cold at every step (measured 2.70 ms against 2.39 for the unsafe ring).
constexpr int kWbWords = 4;
constexpr int kWbDepth = 16;
constexpr int kWbProg = 8;       // progress words per sequence (cluster size <= 8)
static inline int wb_depth(int T) { return T < kWbDepth ? T : kWbDepth; }
static inline long long wb_ring_floats(int B, int T, int N) { return 2ll * ((long long)wb_depth(T) * B * N * kWbWords); }
static inline long long wb_ws_floats(int B, int T, int N) { return wb_ring_floats(B, T, N) + (long long)B * kWbProg + 2; }

template <int DX, int DY>
struct WAcc {   // same slots as BAcc in bsim_bwd_impl.h (shares bsim_bwd_finalize's algebra)
    static constexpr int kSc = 0, kSmm1 = DX, kSmb = 2 * DX, kSmm = 3 * DX, kSf = 4 * DX, kSinit = 5 * DX,
                         kSiota = 6 * DX, kSg = 7 * DX, kN = 7 * DX + DY;
};

template <int DX>
struct WbSlot {
    static constexpr int kFloats = (DX <= 3) ? 4 : 8;
};

template <int DX>
__device__ __forceinline__ void wb_read_slot(const float* p, float (&F)[DX], float& W) {
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

template <int DX>
struct WbIn {   // what an item reads from the forward pass's saves
    float bww, om, lam2;
    int sel;
    float eps[DX], xp[DX], mu1[DX];
};

template <int CH, int NA, int NN, int MASK>
__device__ __forceinline__ void wb_rs_stage(float (&A)[CH][NA], int bit) {
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

// MAXT = 256: one wave per SIMD, so up to 512 VGPRs -- no scratch for Dx >= 3 and room to unroll the small MLPs.
template <int DX, int DY, int H, int M, int MAXT>
__global__ void __launch_bounds__(MAXT) psvowr_bwd_kernel(const WrBwdArgs a) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    using AC = WAcc<DX, DY>;
    constexpr int PS = WbSlot<DX>::kFloats;
    constexpr bool kRolled = (MAXT > 256) || (H > 32) || (DX > 2);   // small MLPs: unrolled, their LDS reads overlap
    constexpr int NA = DX + 1;
    // forward-tile entries per butterfly: 16 (one owned entry per lane after four reduce-scatter stages over the 16
    // quads of a wave), or 8 for Dx >= 3 (three stages, then the two half-waves are summed) to stay inside 256 VGPRs
    constexpr int CH = (DX <= 2) ? 16 : 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTB = blockDim.x, nw = NTB >> 6;
    const int B = a.B, T = a.T, N = a.N;
    const int NP = ((N + 4 * CH - 1) / (4 * CH)) * (4 * CH);   // tile padded (W' = -inf) to whole chunks: no predication
    const int b = blockIdx.y, kb = blockIdx.x, K = gridDim.x;
    const int Nc = (N + K - 1) / K;                // chains per workgroup
    const int c0 = kb * Nc, c1 = min(N, c0 + Nc);  // this workgroup's chains
    const int cpr = NTB / M;
    const int rounds = (Nc + cpr - 1) / cpr;
    const int cl = tid / M, m = tid % M, q = m & 3;
    unsigned* const err = a.err;

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;              // [2][NP][PS]
    float* jacc = tile + 2 * NP * PS;           // [nw][NA][NP] wave-private d F' / d W^ sums of the step
    constexpr int kXq = (DX <= 3) ? 4 : 8;      // floats per child in xq: d loss / d bwXanc_t[k] (DX), ..., parent (last)
    float* red = jacc + nw * NA * NP;           // 64
    float* xq = red + 64;                       // [N][kXq] polled gradients of the children gathered from our chains

    MQ::load(wf, a.f, tid, NTB);
    MG::load(wg, a.g, tid, NTB);
    MQ::load(wqi, a.q1inv, tid, NTB);

    const float kappa = sqrtf(0.5f * kLog2e);
    float isf[DX], rp[DX], isfk[DX], isg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        isf[d] = 1.f / a.sig_f[d];
        rp[d] = isf[d] * kappa;
        isfk[d] = isf[d] / kappa;
    }
    const float ikap2 = 1.f / (kappa * kappa);
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
    float s_init[DX], i_isig[DX], im[DX], mi[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        s_init[d] = a.sig_init[d];
        i_isig[d] = 1.f / a.isig[d];
        im[d] = a.imean[b * DX + d];
        mi[d] = a.minit[b * DX + d];
    }
    const float ninf = -__builtin_huge_valf();

    auto stage = [&](int tt, float* buf) {
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
    if (T >= 2) stage(0, tile);   // step t reads forward tile t-1

    // per-item inputs of round r of step tb (issue only)
    auto load_in = [&](size_t tb, int r, bool last, bool first, WbIn<DX>& s) {
        const int n_raw = c0 + r * cpr + cl;
        const bool valid = n_raw < c1 && (r * cpr + cl) < Nc;
        const int n = valid ? n_raw : max(c1 - 1, 0);
        s.bww = a.bwW[tb * N + n];
        s.sel = a.sel[tb * N + n];
        s.om = a.om_all[(tb * N + n) * M + m];
        s.lam2 = first ? 0.f : a.lam2_all[(tb * N + n) * M + m];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            s.eps[d] = a.eps_b[((tb * DX + d) * N + n) * M + m];
            s.xp[d] = last ? 0.f : a.bwXanc[((tb + B) * DX + d) * N + n];
            s.mu1[d] = last ? 0.f : a.mu1_all[(tb * DX + d) * N + n];
        }
    };

    float acc[AC::kN];
#pragma unroll
    for (int i = 0; i < AC::kN; ++i) acc[i] = 0.f;
    const int nq = NP >> 2;
    constexpr int b0 = 2, b1 = 3, b2 = 4, b3 = 5;   // quad-index bits of the lane id
    const int ebase = ((lane >> b0) & 1) * (CH / 2) + ((lane >> b1) & 1) * (CH / 4) + ((lane >> b2) & 1) * (CH / 8) +
                      (CH >= 16 ? ((lane >> b3) & 1) * (CH / 16) : 0);
    const bool owner = CH >= 16 || ((lane >> b3) & 1) == 0;
    __syncthreads();

    const int D = T < kWbDepth ? T : kWbDepth;      // ring depth
    unsigned seen = 0;                              // (lanes < K: cached progress of cluster member `tid`)
    SEC_INIT(psvowr_bwd)
    for (int t = 0; t < T; ++t) {
        const size_t tb = (size_t)t * B + b;
        const bool last = (t == T - 1), first = (t == 0);
        const float* cur = tile + ((t + 1) & 1) * NP * PS;   // tile(t-1)
        float* nxt = tile + (t & 1) * NP * PS;               // tile(t), read at step t+1
        const bool staging = (t + 1 < T && t >= 1);
        if (staging) {
            if (one_entry) stage_load(t);
            else stage(t, nxt);
        }

        // per-step inputs and the first round's per-item inputs (all saved by the forward pass) are requested BEFORE the
        // exchange below, so that their HBM latency hides behind the poll; issue only -- no arithmetic on them here
        float bm[DX], y[DY];
#pragma unroll
        for (int d = 0; d < DX; ++d) bm[d] = a.bmu2[tb * DX + d];
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = a.obs[tb * DY + k];
        const float lw = a.lseW[tb], dlw = a.dlseW[tb];
        WbIn<DX> in0;
        load_in(tb, 0, last, first, in0);

        for (int i = tid; i < nw * NA * NP; i += NTB) jacc[i] = 0.f;
        // back-pressure: this step publishes into slot (t+1) mod D, whose previous occupant (step t+1-D) every member of the
        // cluster must have consumed (progress >= t+2-D); `seen` caches the member's progress as of the last look
        if (t >= D && !last && tid < K) {     // (an occupant exists once t + 1 - D >= 1)
            const unsigned need = (unsigned)(t + 2 - D);
            unsigned spins = 0;
            while (seen < need) {
                seen = __hip_atomic_load(a.prog + (size_t)b * kWbProg + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen >= need) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21) ||
                    ((spins & 63u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();

        float* ja = jacc + wave * NA * NP;
        SEC(0);   // step inputs requested, accumulators cleared, barrier

        // ---- phase 1: the (chain, sub-particle) items ----------------------------------------------------------
        for (int r = 0; r < rounds; ++r) {
            const int n_raw = c0 + r * cpr + cl;
            const bool valid = n_raw < c1 && (r * cpr + cl) < Nc;
            const int n = valid ? n_raw : max(c1 - 1, 0);
            const int nl = n - c0;
            WbIn<DX> in;
            if (r == 0) in = in0;
            else load_in(tb, r, last, first, in);
            const float aw = valid ? dlw * expf(in.bww - lw) : 0.f;   // d loss / d bw_log_W[t, n]
            const int sel = in.sel;
            float xp[DX], eps[DX], mu1[DX], mu[DX], x[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                eps[d] = in.eps[d];
                if (!last) {
                    xp[d] = in.xp[d];
                    mu1[d] = in.mu1[d];
                    mu[d] = pc[d] * fmaf(pi1[d], mu1[d], pi2[d] * bm[d]);
                    x[d] = fmaf(pc[d], eps[d], mu[d]);
                } else {
                    xp[d] = 0.f;
                    mu1[d] = 0.f;
                    mu[d] = mi[d];
                    x[d] = fmaf(s_init[d], eps[d], mu[d]);
                }
            }
            const float pi_m = valid ? expf(in.om) : 0.f;
            const float issel = (m == sel) ? 1.f : 0.f;
            const float cg = aw * pi_m;                 // d g_m = d Lambda_m = d iota_m
            const float cphi = aw * (pi_m - issel);     // d phi_m

            float dxt[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) dxt[d] = 0.f;

            if (!first) {
                const float lam2 = in.lam2;
                // packed f32: sub-particles (0,1) and (2,3) of the quad share every v_pk_* instruction (as bsim_bwd_impl.h)
                float lq[4], dl[4];
                quad_bcast4(lam2, lq);
                quad_bcast4(cg, dl);
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
                SEC(1);   // item inputs, proposal recompute, coefficients
                for (int c0 = 0; c0 < nq; c0 += CH) {   // nq is a multiple of CH
                    float A[CH][NA];
#pragma unroll
                    for (int i2 = 0; i2 < CH; ++i2) {
                        float F[DX], W;
                        wb_read_slot<DX>(cur + ((c0 + i2) * 4 + q) * PS, F, W);
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
                    wb_rs_stage<CH, NA, CH, (1 << b0)>(A, (lane >> b0) & 1);
                    wb_rs_stage<CH, NA, CH / 2, (1 << b1)>(A, (lane >> b1) & 1);
                    wb_rs_stage<CH, NA, CH / 4, (1 << b2)>(A, (lane >> b2) & 1);
                    if constexpr (CH >= 16) {
                        wb_rs_stage<CH, NA, CH / 8, (1 << b3)>(A, (lane >> b3) & 1);
                    } else {
#pragma unroll
                        for (int d = 0; d < NA; ++d) A[0][d] += xor_lane<(1 << b3)>(A[0][d]);
                    }
                    const int e = c0 + ebase;       // the one entry this lane owns; the wave's rounds accumulate
                    if (owner) {
#pragma unroll
                        for (int d = 0; d < NA; ++d) ja[d * NP + e * 4 + q] += A[0][d];
                    }
                }
                SEC(2);   // pair loop + butterflies
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
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    dxt[d] -= cg * Uo[d] * isfk[d];
                    acc[AC::kSf + d] += cg * (Vo[d] * ikap2 - 1.f) * isf[d];
                }
            } else {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    const float z = (x[d] - im[d]) * i_isig[d];
                    dxt[d] -= cg * z * i_isig[d];
                    acc[AC::kSiota + d] += cg * (z * z - 1.f) * i_isig[d];
                }
            }

            SEC(3);   // quad merges
            float dxp_part[DX], dFo[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                dxp_part[d] = 0.f;
                dFo[d] = 0.f;
            }
            if (!last) {
                float fmx[DX];
                MQ::template eval<kRolled>(wf, x, fmx);
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    const float z = (xp[d] - fmx[d]) * isf[d];
                    dFo[d] = cphi * z * isf[d];
                    dxp_part[d] = -dFo[d];
                    acc[AC::kSf + d] += cphi * (z * z - 1.f) * isf[d];
                }
                MQ::template bwd_input<kRolled>(wf, x, dFo, dxt);
            }
            if (valid) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    a.dFt[((tb * DX + d) * N + n) * M + m] = dFo[d];
                    a.xt[((tb * DX + d) * N + n) * M + m] = x[d];
                }
            }
            {
                float gm[DY], dGo[DY];
                MG::template eval<kRolled>(wg, x, gm);
#pragma unroll
                for (int k = 0; k < DY; ++k) {
                    float dmean = 1.f;
                    if (a.emission) { dmean = emis_dmean(gm[k]); gm[k] = emis_mean(gm[k]); }
                    const float z = (y[k] - gm[k]) * isg[k];
                    dGo[k] = cg * z * isg[k] * dmean;
                    acc[AC::kSg + k] += cg * (z * z - 1.f) * isg[k];
                    if (valid) a.dGt[((tb * DY + k) * N + n) * M + m] = dGo[k];
                }
                MG::template bwd_input<kRolled>(wg, x, dGo, dxt);
            }

            // ---- exchange: scatter d bwXanc_t of ALL chains to the parents this workgroup owns (bwXanc_t[k] =
            //      bwX_t[anc_t[k]]; the owners published d bwXanc_t at the end of their step t-1).  Nothing above depends
            //      on it -- the pair loop and the f / g chains only add into dxt -- so the poll comes here, after them,
            //      and its latency is hidden instead of opening every step. ----------------------------------------------
            SEC(4);   // MLP_f / MLP_g forward + input gradients, row stores
            if (r == 0) {
                if (t >= 1) {
                    const unsigned long long* const slot = a.ring + ((size_t)(t % D) * B + b) * N * kWbWords;
                    const unsigned tag = (unsigned)(t + 1);
                    // Poll, one lane per child chain k whose parent this workgroup owns (all polls in flight together:
                    // one round trip), and stage {d loss / d bwXanc_t[k] (DX), parent} in LDS.  The scatter-add into the
                    // parents is done by the CONSUMERS below in a fixed order -- no LDS float atomics, whose order would
                    // make the gradients differ from launch to launch.
                    for (int k = tid; k < N; k += NTB) {
                        const int p = a.anc[tb * N + k] - c0;
                        float val[DX];
#pragma unroll
                        for (int d = 0; d < DX; ++d) val[d] = 0.f;
                        if (p >= 0 && p < c1 - c0) {   // (its owner wrote the words during step t-1; bounded spin)
                            const unsigned long long* const w = slot + (size_t)k * kWbWords;
                            unsigned spins = 0;
                            for (;;) {
                                unsigned long long v[DX];
                                bool ok = true;
#pragma unroll
                                for (int d = 0; d < DX; ++d) {
                                    v[d] = __hip_atomic_load(w + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = ok && (unsigned)(v[d] >> 32) == tag;
                                }
                                if (ok) {
#pragma unroll
                                    for (int d = 0; d < DX; ++d) val[d] = __uint_as_float((unsigned)v[d]);
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > (1u << 21) ||
                                    ((spins & 63u) == 0u &&
                                     __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    break;
                                }
                            }
                        }
                        float* q8 = xq + k * kXq;
#pragma unroll
                        for (int d = 0; d < DX; ++d) q8[d] = val[d];
                        q8[kXq - 1] = __int_as_float((p >= 0 && p < c1 - c0) ? p : -1);
                    }
                }
                __syncthreads();
                // every lane of this workgroup has read the words of step t: tell the cluster (flow control above)
                if (tid == 0 && t >= 1)
                    __hip_atomic_store(a.prog + (size_t)b * kWbProg + kb, (unsigned)(t + 1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            SEC(5);   // exchange poll + barrier
            if (t >= 1) {
                // d loss / d (selected sub-particle of chain nl) = sum over the children k gathered from it, in a fixed
                // order: lane m of the chain takes k = m, m + M, ..., the M lanes are folded by a fixed xor tree
                float gs[DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) gs[d] = 0.f;
                for (int k = m; k < N; k += M) {
                    const float* q8 = xq + k * kXq;
                    const bool mine = __float_as_int(q8[kXq - 1]) == nl;
#pragma unroll
                    for (int d = 0; d < DX; ++d) gs[d] += mine ? q8[d] : 0.f;
                }
#pragma unroll
                for (int d = 0; d < DX; ++d) dxt[d] += issel * group_sum<M>(gs[d]);
            }

            // ---- reduce over the chain's M sub-particles --------------------------------------------------------
            float dmu[DX], sce[DX], dxp[DX], dim[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                dmu[d] = group_sum<M>(dxt[d]);
                sce[d] = group_sum<M>(dxt[d] * eps[d]);
                dxp[d] = group_sum<M>(dxp_part[d]);
                dim[d] = first ? group_sum<M>(cg * (x[d] - im[d]) * i_isig[d] * i_isig[d]) : 0.f;
            }
            const bool lead = (m == 0) && valid;
            float outv[DX];
            // (selects instead of an `if (lead)`, and the d mu1 row stored behind the MLP pass: no divergent region in front
            //  of that pass -- see the same place in bsim_bwd_impl.h and tools/exec_restore_check.py)
            float dmu1[DX];
            if (!last) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    dmu1[d] = dmu[d] * pc[d] * pi1[d];
                    outv[d] = dmu[d] * pc[d] * pi2[d];
                    acc[AC::kSc + d] += lead ? sce[d] + aw * pic[d] : 0.f;
                    acc[AC::kSmm1 + d] += lead ? dmu[d] * mu1[d] : 0.f;
                    acc[AC::kSmb + d] += lead ? dmu[d] * bm[d] : 0.f;
                    acc[AC::kSmm + d] += lead ? dmu[d] * mu[d] : 0.f;
                }
                MQ::template bwd_input<kRolled>(wqi, xp, dmu1, dxp);
            } else {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    outv[d] = dmu[d];
                    dmu1[d] = 0.f;
                    acc[AC::kSinit + d] += lead ? sce[d] + aw / s_init[d] : 0.f;
                }
            }
            if (lead) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    a.dmu1[(tb * DX + d) * N + n] = dmu1[d];
                    if (!last)     // d loss / d bwXanc_{t+1}[n], polled by the parents' owners at step t+1 (tag t+2)
                        __hip_atomic_store(a.ring + (((size_t)((t + 1) % D) * B + b) * N + n) * kWbWords + d,
                                           ((unsigned long long)(unsigned)(t + 2) << 32) |
                                               (unsigned long long)__float_as_uint(dxp[d]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.dbmu2_rows[(tb * DX + d) * N + n] = last ? 0.f : outv[d];
                    if (last) a.dminit_rows[((size_t)b * DX + d) * N + n] = outv[d];
                    if (first) a.dimean_rows[((size_t)b * DX + d) * N + n] = dim[d];
                }
            }
        }
        __syncthreads();

        SEC(6);   // chain reductions, MLP_q1inv input gradient, publication, barrier
        // ---- phase 2: this workgroup's partial of d Fm[t-1] / d logW[t-1] / d lse[t-1] -----------------------------------
        if (!first) {
            const size_t tbm = tb - B;
            float wsum = 0.f;
            for (int i = tid; i < NA * N; i += NTB) {
                const int d = i / N, j = i - d * N;
                float s = 0.f;
                for (int w = 0; w < nw; ++w) s += jacc[(w * NA + d) * NP + j];
                if (d < DX) a.dFm_part[((tbm * K + kb) * DX + d) * N + j] = s * isfk[d];
                else {
                    a.dlogW_part[(tbm * K + kb) * N + j] = s;
                    wsum += s;
                }
            }
            // d lse[t-1] = - sum_j d W^_j
            wsum = wave_sum(wsum);
            if (lane == 0) red[wave] = wsum;
            __syncthreads();
            if (tid == 0) {
                float s = 0.f;
                for (int w = 0; w < nw; ++w) s += red[w];
                a.dlse_part[tbm * K + kb] = -s;
            }
        }
        if (last) {
            for (int i = tid; i < NA * N; i += NTB) {
                const int d = i / N, j = i - d * N;
                if (d < DX) a.dFm_part[((tb * K + kb) * DX + d) * N + j] = 0.f;
                else a.dlogW_part[(tb * K + kb) * N + j] = 0.f;
            }
            if (tid == 0) a.dlse_part[tb * K + kb] = 0.f;
        }
        if (staging && one_entry) stage_store(nxt);
        __syncthreads();   // (jacc is cleared at the top of the next step)
        SEC(7);   // per-workgroup partials of d Fm / d logW / d lse, barrier
    }

    for (int i = 0; i < AC::kN; ++i) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < AC::kN; ++k) v = (k == i) ? acc[k] : v;
        v = wave_sum(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < nw; ++w) s += red[w];
            a.sacc[((size_t)b * K + kb) * AC::kN + i] = s;
        }
        __syncthreads();
    }
}

// same algebra as bsim_bwd_finalize (one row of sums per sequence)
template <int DX, int DY>
__global__ void psvowr_bwd_finalize(const float* __restrict__ sacc, int rows, const float* sig_q1inv, const float* sig_bq2,
                                    float* dsig_f, float* dsig_g, float* dsig_q1inv, float* dsig_bq2, float* dsig_init,
                                    float* disig) {
    using AC = WAcc<DX, DY>;
    __shared__ float tot[AC::kN];
    const int d = threadIdx.x;
    for (int k = 0; k < AC::kN; ++k) {
        float v = 0.f;
        for (int r = d; r < rows; r += 64) v += sacc[(size_t)r * AC::kN + k];
        v = wave_sum(v);
        if (d == 0) tot[k] = v;
    }
    __syncthreads();
    if (d < DX) {
        const float i1 = 1.f / sig_q1inv[d], i2 = 1.f / sig_bq2[d];
        const float c = 1.f / (i1 + i2);
        const float dc = tot[AC::kSc + d] + tot[AC::kSmm + d] / c;
        const float di1 = c * tot[AC::kSmm1 + d] - c * c * dc;
        const float di2 = c * tot[AC::kSmb + d] - c * c * dc;
        dsig_q1inv[d] = -i1 * i1 * di1;
        dsig_bq2[d] = -i2 * i2 * di2;
        dsig_f[d] = tot[AC::kSf + d];
        dsig_init[d] = tot[AC::kSinit + d];
        disig[d] = tot[AC::kSiota + d];
    }
    if (d < DY) dsig_g[d] = tot[AC::kSg + d];
}

struct WrBwdOut {
    float *dsig_f, *dsig_g, *dsig_q1inv, *dsig_bq2, *dsig_init, *disig;
};

template <int DX, int DY, int H, int M>
static int launch_wr_bwd(const WrBwdArgs& a, const WrBwdOut& o, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, PSVO_L>;
    using MG = MlpLds<DX, H, DY, PSVO_L>;
    constexpr int PS = WbSlot<DX>::kFloats;
    constexpr int CH = (DX <= 2) ? 16 : 8;         // (as in the kernel)
    const int NP = ((a.N + 4 * CH - 1) / (4 * CH)) * (4 * CH);
    const int K = wr_cluster(a.B, a.N, M);
    const int Nc = (a.N + K - 1) / K;
    long long items = (long long)Nc * M;
    int NTB = (int)(((items + 63) / 64) * 64);
    if (NTB > 512) NTB = 512;
    const int nw = NTB / 64;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * PS + (size_t)nw * (DX + 1) * NP +
                                        64 + (size_t)((DX <= 3) ? 4 : 8) * a.N);
    if (lds > 160 * 1024) return PSVO_ERR_UNSUPPORTED;
    clear_hip_error();
    // tags of an earlier launch must not be mistaken for this one's: clear the ring and the error flag
    if (hipMemsetAsync(a.ring, 0, sizeof(float) * (size_t)wb_ws_floats(a.B, a.T, a.N), stream) != hipSuccess)
        return launch_status();
    WrBwdArgs args = a;
    void* kargs[] = {(void*)&args};
    const void* fn = NTB <= 256 ? (const void*)psvowr_bwd_kernel<DX, DY, H, M, 256>
                                : (const void*)psvowr_bwd_kernel<DX, DY, H, M, 512>;
    // (K == 1: no cross-workgroup exchange, so no residency requirement -- see psvowr_fwd.hip)
    const hipError_t e = K > 1 ? hipLaunchCooperativeKernel(fn, dim3(K, a.B), dim3(NTB), kargs, lds, stream)
                               : hipLaunchKernel(fn, dim3(K, a.B), dim3(NTB), kargs, lds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = e;
        (void)hipGetLastError();
        return PSVO_ERR_HIP;
    }
    hipLaunchKernelGGL((psvowr_bwd_finalize<DX, DY>), dim3(1), dim3(64), 0, stream, a.sacc, a.B * K, a.sig_q1inv,
                       a.sig_bq2, o.dsig_f, o.dsig_g, o.dsig_q1inv, o.dsig_bq2, o.dsig_init, o.disig);
    return launch_status();
}

template <int DX, int DY, int H>
static int wb_dispatch_m(const WrBwdArgs& a, const WrBwdOut& o, int M, hipStream_t s) {
    switch (M) {
        case 4: return launch_wr_bwd<DX, DY, H, 4>(a, o, s);
        case 8: return launch_wr_bwd<DX, DY, H, 8>(a, o, s);
        case 16: return launch_wr_bwd<DX, DY, H, 16>(a, o, s);
        case 32: return launch_wr_bwd<DX, DY, H, 32>(a, o, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX, int DY>
static int wb_dispatch_h(const WrBwdArgs& a, const WrBwdOut& o, int H, int M, hipStream_t s) {
    switch (H) {
#if PSVO_L == 1   // (two hidden layers: widths 32 and 64; narrower ones are zero-padded upstream)
        case 16: return wb_dispatch_m<DX, DY, 16>(a, o, M, s);
#endif
        case 32: return wb_dispatch_m<DX, DY, 32>(a, o, M, s);
        case 64: return wb_dispatch_m<DX, DY, 64>(a, o, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX>
static int wb_dispatch_dy(const WrBwdArgs& a, const WrBwdOut& o, int Dy, int H, int M, hipStream_t s) {
    switch (Dy) {
        case 1: return wb_dispatch_h<DX, 1>(a, o, H, M, s);
        case 2: return wb_dispatch_h<DX, 2>(a, o, H, M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // inline namespace PSVO_LNS
}  // namespace psvo

#if PSVO_L == 1   // (sizing helpers: independent of the number of hidden layers)
extern "C" long long psvo_bsimwr_bwd_ws_floats(int B, int T, int N, int Dx) {
    (void)Dx;
    return psvo::wb_ws_floats(B, T, N);
}
#endif

PSVO_L2_DECL(psvo_bsimwr_backward)
PSVO_ENTRY(psvo_bsimwr_backward)(
    const psvo_desc* desc, const float* Fm, const float* logW, const float* lse, const psvo_mlp* f, const psvo_mlp* g,
    const psvo_mlp* q1_inv, const float* sig_f, const float* sig_g, const float* sig_q1inv, const float* sig_bq2,
    const float* bmu2, const float* minit, const float* sig_init, const float* imean, const float* isig,
    const float* obs, const float* eps_b, const float* bwXanc, const float* bwW, const float* lseW, const int32_t* sel,
    const int32_t* anc, const float* lam2_all, const float* om_all, const float* mu1_all, const float* dlseW, float* xt,
    float* dFt, float* dGt, float* dmu1, float* dFm_part, float* dlogW_part, float* dlse_part, float* dbmu2_rows,
    float* dminit_rows, float* dimean_rows, float* dsig_f, float* dsig_g, float* dsig_q1inv, float* dsig_bq2,
    float* dsig_init, float* disig, float* sacc, float* ws, void* stream) {
    using namespace psvo;
    if (!desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;
#if PSVO_L == 1
    if (desc && desc->layers == 2)
        return psvo_bsimwr_backward_l2(desc, Fm, logW, lse, f, g, q1_inv, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2,
            minit, sig_init, imean, isig, obs, eps_b, bwXanc, bwW, lseW, sel, anc, lam2_all, om_all, mu1_all,
            dlseW, xt, dFt, dGt, dmu1, dFm_part, dlogW_part, dlse_part, dbmu2_rows, dminit_rows, dimean_rows,
            dsig_f, dsig_g, dsig_q1inv, dsig_bq2, dsig_init, disig, sacc, ws, stream);
#endif
    if (!mlp_layers_ok(f) || !mlp_layers_ok(g) || !mlp_layers_ok(q1_inv)) return PSVO_ERR_INVALID;
    if (!desc || !Fm || !logW || !lse || !f || !g || !q1_inv || !sig_f || !sig_g || !sig_q1inv || !sig_bq2 || !bmu2 ||
        !minit || !sig_init || !imean || !isig || !obs || !eps_b || !bwXanc || !bwW || !lseW || !sel || !anc ||
        !lam2_all || !om_all || !mu1_all || !dlseW || !xt || !dFt || !dGt || !dmu1 || !dFm_part || !dlogW_part ||
        !dlse_part || !dbmu2_rows || !dminit_rows || !dimean_rows || !dsig_f || !dsig_g || !dsig_q1inv || !dsig_bq2 ||
        !dsig_init || !disig || !sacc || !ws)
        return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T < 2 || desc->N <= 0 || desc->M <= 0) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    WrBwdArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.logW = logW; a.lse = lse;
    a.sig_f = sig_f; a.sig_g = sig_g; a.sig_q1inv = sig_q1inv; a.sig_bq2 = sig_bq2;
    a.bmu2 = bmu2; a.minit = minit; a.sig_init = sig_init; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.bwXanc = bwXanc; a.bwW = bwW; a.lseW = lseW; a.sel = sel; a.anc = anc;
    a.lam2_all = lam2_all; a.om_all = om_all; a.mu1_all = mu1_all; a.dlseW = dlseW;
    a.xt = xt; a.dFt = dFt; a.dGt = dGt; a.dmu1 = dmu1;
    a.dFm_part = dFm_part; a.dlogW_part = dlogW_part; a.dlse_part = dlse_part;
    a.dbmu2_rows = dbmu2_rows; a.dminit_rows = dminit_rows; a.dimean_rows = dimean_rows; a.sacc = sacc;
    if (reinterpret_cast<uintptr_t>(ws) & 7u) return PSVO_ERR_INVALID;      // 64-bit words
    a.ring = reinterpret_cast<unsigned long long*>(ws);
    a.prog = reinterpret_cast<unsigned*>(ws + wb_ring_floats(desc->B, desc->T, desc->N));
    a.err = reinterpret_cast<unsigned*>(ws + wb_ws_floats(desc->B, desc->T, desc->N) - 1);
    WrBwdOut o{dsig_f, dsig_g, dsig_q1inv, dsig_bq2, dsig_init, disig};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return wb_dispatch_dy<2>(a, o, desc->Dy, desc->H, desc->M, s);
        case 3: return wb_dispatch_dy<3>(a, o, desc->Dy, desc->H, desc->M, s);
        case 4: return wb_dispatch_dy<4>(a, o, desc->Dy, desc->H, desc->M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
