// PSVOwR backward simulation: PSVO's backward simulation with an additional multinomial resampling
// ACROSS the chains of a sequence after every step and a per-step ELBO (reference
// src/SMC/PSVOwR.py:65-198).  The cross-chain draw couples all N chains of a sequence once per step, so
// this variant runs ONE persistent workgroup per sequence (up to 1024 lanes) that walks the N*M
// (chain, sub-particle) items in rounds; chain state lives in LDS between steps.  The per-item work
// (proposal, MLP_f / MLP_g, quad-blocked pass over the LDS-staged forward tile, normalisation and draw
// over the M sub-particles) is the same arithmetic as psvo_bsim_forward.
//
// With omega_raw = Lambda + phi + g - q the reference's per-step weight
//     bw_log_W = (Lambda + g)_sel - q_sel - omega_sel - log M          (PSVOwR.py:135-142)
// equals  logsumexp_m(omega_raw) - phi_sel - log M, which is what the kernel writes.
#include "common.h"

namespace psvo {

struct WrArgs {
    int B, T, N;
    psvo_mlp f, g, q1inv;
    const float *Fm, *logW, *lse;
    const float *sig_f, *sig_g, *sig_q1inv, *sig_bq2;
    const float *bmu2, *minit, *sig_init, *imean, *isig;
    const float *obs, *eps_b, *u_b, *u_r;
    const int32_t *sel_in, *anc_in;
    float *bwX, *bwXanc, *bwW, *lseW;
    int32_t *sel_out, *anc_out;
    float *lam2_all, *om_all, *mu1_all;
};

template <int DX>
struct WrSlot {
    static constexpr int kFloats = (DX <= 3) ? 4 : 8;
};

__device__ __forceinline__ void wr_lse2_merge(float& m, float& s, float m2, float s2) {
    const float nm = fmaxf(m, m2);
    const float a = (m == nm) ? 1.f : exp2_fast(m - nm);
    const float b = (m2 == nm) ? 1.f : exp2_fast(m2 - nm);
    s = s * a + s2 * b;
    m = nm;
}

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

template <int DX, int DY, int H, int M>
__global__ void __launch_bounds__(1024) psvowr_fwd_kernel(const WrArgs a) {
    using MQ = MlpLds<DX, H, DX>;
    using MG = MlpLds<DX, H, DY>;
    constexpr int PS = WrSlot<DX>::kFloats;
    constexpr bool kRolled = true;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTB = blockDim.x, nw = NTB >> 6;
    const int B = a.B, T = a.T, N = a.N;
    const int NP = (N + 3) & ~3;
    const int b = blockIdx.x;
    const int cpr = NTB / M;                       // chains per round
    const int rounds = (N + cpr - 1) / cpr;
    const int cl = tid / M, m = tid % M, q = m & 3;
    const int gbase = lane - m;

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;                 // [2][NP][PS]
    float* xanc = tile + 2 * NP * PS;              // [DX][N]  resampled chain states (x_{t+1} of every chain)
    float* xsel = xanc + DX * N;                   // [DX][N]  selected sub-particle of every chain
    float* omsel = xsel + DX * N;                  // [N]      its normalised log-weight (logits of the cross-chain draw)
    float* wsel = omsel + N;                       // [N]      per-step weight bw_log_W
    float* cdf = wsel + N;                         // [N]
    float* red = cdf + N;                          // 64 floats scratch

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
    const float logN = logf((float)N);
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
    if (T >= 2) stage(T - 2, tile);
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * B + b;
        const float* cur = tile + ((T - 1 - t) & 1) * NP * PS;
        float* nxt = tile + ((T - t) & 1) * NP * PS;
        const bool last = (t == T - 1);
        if (t >= 2) stage(t - 2, nxt);   // consumed two barriers from now

        float bm[DX], y[DY];
#pragma unroll
        for (int d = 0; d < DX; ++d) bm[d] = a.bmu2[tb * DX + d];
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = a.obs[tb * DY + k];

        for (int r = 0; r < rounds; ++r) {
            const int n_raw = r * cpr + cl;
            const bool valid = n_raw < N;
            const int n = valid ? n_raw : N - 1;
            float xp[DX], eps[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                xp[d] = last ? 0.f : xanc[d * N + n];
                eps[d] = a.eps_b[((tb * DX + d) * N + n) * M + m];
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
            const float g_lp = diag_lp<DY>(y, gm, isg, kg);

            // ---- filter term over the LDS tile (quad register blocking, online lse in log2) ----------------
            float lam;
            if (t >= 1) {
                float xq[4][DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    float t4[4];
                    quad_bcast4(x[d] * rp[d], t4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) xq[i][d] = t4[i];
                }
                float mx[4], sm[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    mx[i] = ninf;
                    sm[i] = 0.f;
                }
                const int nq = NP >> 2;
                for (int jj = 0; jj < nq; ++jj) {
                    float F[DX], W;
                    wr_read_slot<DX>(cur + (jj * 4 + q) * PS, F, W);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float acc = W;
#pragma unroll
                        for (int d = 0; d < DX; ++d) {
                            const float df = xq[i][d] - F[d];
                            acc = fmaf(-df, df, acc);
                        }
                        const float nm = fmaxf(mx[i], acc);
                        const float base = (nm == ninf) ? 0.f : nm;
                        sm[i] = sm[i] * exp2_fast(mx[i] - base) + exp2_fast(acc - base);
                        mx[i] = nm;
                    }
                }
                float lm = ninf, ls = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float mm = mx[i], ss = sm[i];
                    wr_lse2_merge(mm, ss, xor_lane<1>(mx[i]), xor_lane<1>(sm[i]));
                    const float m2 = xor_lane<2>(mm), s2 = xor_lane<2>(ss);
                    wr_lse2_merge(mm, ss, m2, s2);
                    if (i == q) {
                        lm = mm;
                        ls = ss;
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
            if (valid && m == 0) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    xsel[d * N + n] = xs[d];
                    a.bwX[(tb * DX + d) * N + n] = xs[d];
                }
                omsel[n] = om_s;
                wsel[n] = bw;
                a.bwW[tb * N + n] = bw;
                a.sel_out[tb * N + n] = sel;
            }
        }
        __syncthreads();

        // ---- resample the chains: a[k] ~ Categorical(softmax_n omega_sel[n]), k = 0..N-1 (PSVOwR.py:103,145,185) ----
        {
            const bool act = tid < N;
            const float lo = act ? omsel[tid] : ninf;
            const float lw = act ? wsel[tid] : ninf;
            // block max of both vectors
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
            const float ew = act ? expf(lw - mw) : 0.f;
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
            if (tid == 0) a.lseW[tb] = mw + logf(totw);   // logsumexp_n bw_log_W[t, :, b]
            __syncthreads();
            if (act) {
                int anc;
                if (a.anc_in) {
                    anc = a.anc_in[tb * N + tid];
                } else {
                    const float target = a.u_r[tb * N + tid] * tot;
                    int pos = 0;
                    for (int s = 1 << (31 - __clz(N)); s > 0; s >>= 1) {
                        const int p = pos + s;
                        if (p <= N && cdf[p - 1] <= target) pos = p;
                    }
                    anc = min(pos, N - 1);
                }
                a.anc_out[tb * N + tid] = anc;
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    const float v = xsel[d * N + anc];
                    xanc[d * N + tid] = v;          // (xanc is only read in the item rounds, after the barrier)
                    a.bwXanc[(tb * DX + d) * N + tid] = v;
                }
            }
            (void)logN;
        }
        __syncthreads();
    }
}

template <int DX, int DY, int H, int M>
static int launch_wr_fwd(const WrArgs& a, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX>;
    using MG = MlpLds<DX, H, DY>;
    constexpr int PS = WrSlot<DX>::kFloats;
    const int NP = (a.N + 3) & ~3;
    long long items = (long long)a.N * M;
    int NTB = (int)(((items + 63) / 64) * 64);
    if (NTB > 1024) NTB = 1024;
    if (a.N > NTB) return PSVO_ERR_UNSUPPORTED;   // the cross-chain draw uses one lane per chain
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * PS + (2 * DX + 3) * (size_t)a.N + 64);
    if (lds > 160 * 1024) return PSVO_ERR_UNSUPPORTED;
    clear_hip_error();
    hipLaunchKernelGGL((psvowr_fwd_kernel<DX, DY, H, M>), dim3(a.B), dim3(NTB), lds, stream, a);
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
        case 16: return wr_dispatch_m<DX, DY, 16>(a, M, s);
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

}  // namespace psvo

extern "C" int psvo_bsimwr_forward(const psvo_desc* desc, const float* Fm, const float* logW, const float* lse,
                                   const psvo_mlp* f, const psvo_mlp* g, const psvo_mlp* q1_inv, const float* sig_f,
                                   const float* sig_g, const float* sig_q1inv, const float* sig_bq2, const float* bmu2,
                                   const float* minit, const float* sig_init, const float* imean, const float* isig,
                                   const float* obs, const float* eps_b, const float* u_b, const float* u_r,
                                   const int32_t* sel_in, const int32_t* anc_in, float* bwX, float* bwXanc, float* bwW,
                                   float* lseW, int32_t* sel_out, int32_t* anc_out, float* lam2_all, float* om_all,
                                   float* mu1_all, void* stream) {
    using namespace psvo;
    if (!desc || !Fm || !logW || !lse || !f || !g || !q1_inv || !sig_f || !sig_g || !sig_q1inv || !sig_bq2 || !bmu2 ||
        !minit || !sig_init || !imean || !isig || !obs || !eps_b || !bwX || !bwXanc || !bwW || !lseW || !sel_out ||
        !anc_out)
        return PSVO_ERR_INVALID;
    if ((!u_b && !sel_in) || (!u_r && !anc_in)) return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T < 2 || desc->N <= 0 || desc->M <= 0) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;
    WrArgs a;
    a.B = desc->B; a.T = desc->T; a.N = desc->N;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.logW = logW; a.lse = lse;
    a.sig_f = sig_f; a.sig_g = sig_g; a.sig_q1inv = sig_q1inv; a.sig_bq2 = sig_bq2;
    a.bmu2 = bmu2; a.minit = minit; a.sig_init = sig_init; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.u_b = u_b; a.u_r = u_r; a.sel_in = sel_in; a.anc_in = anc_in;
    a.bwX = bwX; a.bwXanc = bwXanc; a.bwW = bwW; a.lseW = lseW; a.sel_out = sel_out; a.anc_out = anc_out;
    a.lam2_all = lam2_all; a.om_all = om_all; a.mu1_all = mu1_all;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return wr_dispatch_dy<2>(a, desc->Dy, desc->H, desc->M, s);
        case 3: return wr_dispatch_dy<3>(a, desc->Dy, desc->H, desc->M, s);
        case 4: return wr_dispatch_dy<4>(a, desc->Dy, desc->H, desc->M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
