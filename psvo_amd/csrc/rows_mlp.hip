// One-hidden-layer MLP over plain rows: the "hoisted" per-(sequence, step) networks of the path -- q0, q2,
// BSim_q2, BSim_q_init (tf_mvn.mean of MLP_transformation, reference src/transformation/MLP.py:48-68 as called from
// src/SMC/SVO.py:80-84,134-138 and src/SMC/PSVO.py:86-87,120-122).  They are < 1 % of the arithmetic but, as
// PyTorch ops, were ~90 of the ~180 launches of a training step, most of them on its critical path.
//   forward : out = relu(X W1 + b1) W2 + b2                              one launch
//   backward: dX, and [dW1 | db1 | dW2 | db2] as per-workgroup partials  one launch + a deterministic reduction
// Rows are (R, Din) row-major; Din <= 128, H in {16, 32, 64}, Dout <= 4.
#include "common.h"

namespace psvo {

struct RowsArgs {
    long long R;
    int Din, Dout;
    const float *X, *dOut;
    psvo_mlp w;
    float *out, *dX, *partial;
};

// ---- forward: one lane per row, weights in LDS (wave-uniform reads) --------------------------------------------------
template <int H>
__global__ void __launch_bounds__(128) rows_mlp_fwd_kernel(const RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Din = a.Din, Dout = a.Dout, tid = threadIdx.x;
    float* W1 = smem;                  // [Din][H]
    float* b1 = W1 + Din * H;          // [H]
    float* W2 = b1 + H;                // [H][Dout]
    float* b2 = W2 + H * Dout;         // [Dout]
    for (int i = tid; i < Din * H; i += blockDim.x) W1[i] = a.w.W1[i];
    for (int i = tid; i < H; i += blockDim.x) b1[i] = a.w.b1[i];
    for (int i = tid; i < H * Dout; i += blockDim.x) W2[i] = a.w.W2[i];
    for (int i = tid; i < Dout; i += blockDim.x) b2[i] = a.w.b2[i];
    __syncthreads();
    const long long r = (long long)blockIdx.x * blockDim.x + tid;
    if (r >= a.R) return;
    float h[H];
#pragma unroll
    for (int k = 0; k < H; ++k) h[k] = b1[k];
    const float* x = a.X + r * Din;
    for (int i = 0; i < Din; ++i) {
        const float xi = x[i];
        const float4* wr = reinterpret_cast<const float4*>(W1 + i * H);
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 wv = wr[k4];
            h[4 * k4 + 0] = fmaf(xi, wv.x, h[4 * k4 + 0]);
            h[4 * k4 + 1] = fmaf(xi, wv.y, h[4 * k4 + 1]);
            h[4 * k4 + 2] = fmaf(xi, wv.z, h[4 * k4 + 2]);
            h[4 * k4 + 3] = fmaf(xi, wv.w, h[4 * k4 + 3]);
        }
    }
    for (int o = 0; o < Dout; ++o) {
        float s = b2[o];
#pragma unroll
        for (int k = 0; k < H; ++k) s = fmaf(fmaxf(h[k], 0.f), W2[k * Dout + o], s);
        a.out[r * Dout + o] = s;
    }
}

// ---- backward: RB rows per workgroup, 256 / RB lanes per row -----------------------------------------------------------------
// RB = 64 for many rows (fewer partials to fold, weights staged once per 64 rows); RB = 16 below kRowsBwdSmall rows, where the
// launch is a handful of workgroups on the dependent chain between the backward simulation and the encoder BPTT and the
// per-thread loop over the rows of the workgroup (NP * RB / 256 dependent LDS steps) is what it costs.
constexpr int kRowsBwdSmall = 1024;
static int g_rows_bwd_rb = 0;      // psvo_set_tuning(PSVO_TUNE_ROWS_BWD, 0 | 16 | 64): 0 = by the number of rows
int set_rows_bwd_rb(int v) {
    if (v != 0 && v != 16 && v != 64) return PSVO_ERR_INVALID;
    g_rows_bwd_rb = v;
    return PSVO_OK;
}
int get_rows_bwd_rb() { return g_rows_bwd_rb; }
static inline int rows_bwd_rb(long long R) { return g_rows_bwd_rb ? g_rows_bwd_rb : (R <= kRowsBwdSmall ? 16 : 64); }
template <int H, int RB>
__global__ void __launch_bounds__(256) rows_mlp_bwd_kernel(const RowsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LPR = 256 / RB, HP = H / LPR;
    static_assert(H % LPR == 0, "hidden units split evenly over the lanes of a row");
    const int Din = a.Din, Dout = a.Dout, tid = threadIdx.x;
    const int XS = Din + 1, HS = H + 1;    // padded row strides (bank conflicts)
    float* W1 = smem;                       // [Din][H]
    float* b1 = W1 + Din * H;               // [H]
    float* W2 = b1 + H;                     // [H][4]
    float* Xs = W2 + H * 4;                 // [RB][XS]
    float* dhs = Xs + RB * XS;              // [RB][HS]
    float* hs = dhs + RB * HS;              // [RB][HS]
    float* dos = hs + RB * HS;              // [RB][4]
    const long long r0 = (long long)blockIdx.x * RB;
    for (int i = tid; i < Din * H; i += 256) W1[i] = a.w.W1[i];
    for (int i = tid; i < H; i += 256) b1[i] = a.w.b1[i];
    for (int i = tid; i < H * 4; i += 256) W2[i] = (i & 3) < Dout ? a.w.W2[(i >> 2) * Dout + (i & 3)] : 0.f;
    for (int i = tid; i < RB * Din; i += 256) {
        const int rl = i / Din, c = i - rl * Din;
        Xs[rl * XS + c] = (r0 + rl < a.R) ? a.X[r0 * Din + i] : 0.f;
    }
    for (int i = tid; i < RB * 4; i += 256) {
        const int rl = i >> 2, o = i & 3;
        dos[i] = (o < Dout && r0 + rl < a.R) ? a.dOut[(r0 + rl) * Dout + o] : 0.f;
    }
    __syncthreads();
    const int rl = tid / LPR, p = tid % LPR;
    {   // hidden slice of this lane: pre-activation, h, d h
        float pre[HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) pre[k] = b1[p * HP + k];
        for (int i = 0; i < Din; ++i) {
            const float xi = Xs[rl * XS + i];
#pragma unroll
            for (int k = 0; k < HP; ++k) pre[k] = fmaf(xi, W1[i * H + p * HP + k], pre[k]);
        }
        const float4 g = *reinterpret_cast<const float4*>(dos + rl * 4);
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            const float4 w2 = *reinterpret_cast<const float4*>(W2 + (p * HP + k) * 4);
            const float dh = fmaf(g.x, w2.x, fmaf(g.y, w2.y, fmaf(g.z, w2.z, g.w * w2.w)));
            hs[rl * HS + p * HP + k] = fmaxf(pre[k], 0.f);
            dhs[rl * HS + p * HP + k] = pre[k] > 0.f ? dh : 0.f;
        }
    }
    __syncthreads();
    if (a.dX && r0 + rl < a.R) {   // d X[r][i] = sum_k d h[r][k] W1[i][k]
        for (int i = p; i < Din; i += LPR) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < H; ++k) s = fmaf(dhs[rl * HS + k], W1[i * H + k], s);
            a.dX[(r0 + rl) * Din + i] = s;
        }
    }
    // this workgroup's partial of [dW1 (Din,H) | db1 (H) | dW2 (H,Dout) | db2 (Dout)]
    const int NP = Din * H + H + H * Dout + Dout;
    float* part = a.partial + (size_t)blockIdx.x * NP;
    for (int e = tid; e < NP; e += 256) {
        float s = 0.f;
        if (e < Din * H) {
            const int i = e / H, k = e - i * H;
            for (int r = 0; r < RB; ++r) s = fmaf(Xs[r * XS + i], dhs[r * HS + k], s);
        } else if (e < Din * H + H) {
            const int k = e - Din * H;
            for (int r = 0; r < RB; ++r) s += dhs[r * HS + k];
        } else if (e < Din * H + H + H * Dout) {
            const int q = e - Din * H - H, k = q / Dout, o = q - k * Dout;
            for (int r = 0; r < RB; ++r) s = fmaf(hs[r * HS + k], dos[r * 4 + o], s);
        } else {
            const int o = e - Din * H - H - H * Dout;
            for (int r = 0; r < RB; ++r) s += dos[r * 4 + o];
        }
        part[e] = s;
    }
}

// out[p] (+)= sum_blk partial[blk][p]: one wave per output element (fixed order, deterministic)
__global__ void rows_reduce_partials_kernel(const float* __restrict__ partial, int nblk, int NP, float* __restrict__ out,
                                            int accumulate) {
    const int p = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * NP + p];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[p] = accumulate ? out[p] + s : s;
}

template <int H>
static int launch_rows_fwd(const RowsArgs& a, hipStream_t s) {
    const size_t lds = sizeof(float) * ((size_t)a.Din * H + H + (size_t)H * a.Dout + a.Dout + 4);
    clear_hip_error();
    hipLaunchKernelGGL((rows_mlp_fwd_kernel<H>), dim3((unsigned)((a.R + 127) / 128)), dim3(128), lds, s, a);
    return launch_status();
}

template <int H>
static int launch_rows_bwd(const RowsArgs& a, float* grad, int accumulate, hipStream_t s) {
    const int rb = rows_bwd_rb(a.R);
    const int nblk = (int)((a.R + rb - 1) / rb);
    const int NP = a.Din * H + H + H * a.Dout + a.Dout;
    const size_t lds = sizeof(float) * ((size_t)a.Din * H + H + 4 * H + rb * (a.Din + 1) + 2 * rb * (H + 1) + rb * 4);
    if (lds > 160 * 1024) return PSVO_ERR_UNSUPPORTED;
    clear_hip_error();
    if (rb == 16) hipLaunchKernelGGL((rows_mlp_bwd_kernel<H, 16>), dim3(nblk), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((rows_mlp_bwd_kernel<H, 64>), dim3(nblk), dim3(256), lds, s, a);
    hipLaunchKernelGGL(rows_reduce_partials_kernel, dim3(NP), dim3(64), 0, s, a.partial, nblk, NP, grad, accumulate);
    return launch_status();
}

}  // namespace psvo

extern "C" int psvo_rows_mlp_blocks(long long R) {
    const int rb = psvo::rows_bwd_rb(R);
    return (int)((R + rb - 1) / rb);
}

extern "C" int psvo_rows_mlp_forward(long long R, int Din, int H, int Dout, const float* X, const psvo_mlp* w,
                                     float* out, void* stream) {
    using namespace psvo;
    if (!X || !w || !out || R <= 0) return PSVO_ERR_INVALID;
    if (Din <= 0 || Din > 128 || Dout <= 0 || Dout > 4) return PSVO_ERR_UNSUPPORTED;
    RowsArgs a{R, Din, Dout, X, nullptr, *w, out, nullptr, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (H) {
        case 16: return launch_rows_fwd<16>(a, s);
        case 32: return launch_rows_fwd<32>(a, s);
        case 64: return launch_rows_fwd<64>(a, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

extern "C" int psvo_rows_mlp_backward(long long R, int Din, int H, int Dout, const float* X, const float* dOut,
                                      const psvo_mlp* w, float* dX, float* partial, float* grad, int accumulate,
                                      void* stream) {
    using namespace psvo;
    if (!X || !dOut || !w || !partial || !grad || R <= 0) return PSVO_ERR_INVALID;
    if (Din <= 0 || Din > 128 || Dout <= 0 || Dout > 4) return PSVO_ERR_UNSUPPORTED;
    RowsArgs a{R, Din, Dout, X, dOut, *w, nullptr, dX, partial};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (H) {
        case 16: return launch_rows_bwd<16>(a, grad, accumulate, s);
        case 32: return launch_rows_bwd<32>(a, grad, accumulate, s);
        case 64: return launch_rows_bwd<64>(a, grad, accumulate, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
