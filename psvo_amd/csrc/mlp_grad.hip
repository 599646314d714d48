// Weight gradients of a one-hidden-layer MLP over a large batch of independent rows.
//
// The time-sequential backward kernels (filter_bwd / bsim_bwd) only propagate gradients to MLP
// *inputs*; for every MLP evaluation they leave behind the row's input x and the gradient of the
// loss w.r.t. the row's output.  This kernel turns those rows into parameter gradients:
//     dW2 += h^T dout,  db2 += dout,  dh = relu'(pre) * (dout W2^T),  dW1 += x^T dh,  db1 += dh.
// It is the only GEMM-shaped contraction of the backward pass (K = #rows, up to 1.3e7 at C*), but
// its other two dimensions are (Din + 1) x H and H x Dout with Din, Dout <= 4: an f32 MFMA tile
// (32x32x2 / 16x16x4) would be <= 12 % occupied at the same peak rate as the vector ALU, so each
// lane keeps the full set of (Din+1)*H + (H+1)*Dout accumulators in VGPRs over a grid-stride loop
// and the workgroup reduces them once at the end (deterministic two-stage reduction, no atomics).
//
// Row layout ("segments"): X[s][i][l], dOut[s][o][l] with s < S segments of L contiguous rows --
// (T,B,Dx,N) particle tensors have L = N, the bsim sub-particle tensors (T,B,Dx,N,M) have L = N*M.
#include "common.h"

namespace psvo {

struct WgradArgs {
    long long S;     // segments
    int L;           // rows per segment
    const float* X;  // [S][DIN][L]
    const float* dOut;  // [S][DOUT][L]
    psvo_mlp w;
    float* partial;  // [gridDim.x][NP]
};

// One block column (blockIdx.y) owns KC = 16 hidden units, so the accumulators are
// (DIN + 1 + DOUT) * 16 VGPRs whatever H is; the chunk's weights sit in registers too.
constexpr int kKC = 16;

template <int DIN, int DOUT>
__global__ void __launch_bounds__(256) mlp_wgrad_kernel(const WgradArgs a, const int H) {
    constexpr int KC = kKC;
    constexpr int NPC = (DIN + 1 + DOUT) * KC + DOUT;  // per-chunk partial sums (+ db2 in chunk 0)
    __shared__ float red[4][NPC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.y * KC;

    float w1[DIN][KC], b1[KC], w2[DOUT][KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        b1[k] = a.w.b1[k0 + k];
#pragma unroll
        for (int i = 0; i < DIN; ++i) w1[i][k] = a.w.W1[i * H + k0 + k];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) w2[o][k] = a.w.W2[(k0 + k) * DOUT + o];
    }

    float gW1[DIN][KC], gb1[KC], gW2[DOUT][KC], gb2[DOUT];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        gb1[k] = 0.f;
#pragma unroll
        for (int i = 0; i < DIN; ++i) gW1[i][k] = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) gW2[o][k] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < DOUT; ++o) gb2[o] = 0.f;

    const long long R = a.S * a.L;
    const long long stride = (long long)gridDim.x * 256;
    for (long long r = (long long)blockIdx.x * 256 + tid; r < R; r += stride) {
        const long long s = r / a.L;
        const int l = (int)(r - s * a.L);
        float x[DIN], dout[DOUT];
#pragma unroll
        for (int i = 0; i < DIN; ++i) x[i] = a.X[(s * DIN + i) * a.L + l];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            dout[o] = a.dOut[(s * DOUT + o) * a.L + l];
            gb2[o] += dout[o];
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            float pre = b1[k];
#pragma unroll
            for (int i = 0; i < DIN; ++i) pre = fmaf(x[i], w1[i][k], pre);
            const float h = fmaxf(pre, 0.f);
            float dh = 0.f;
#pragma unroll
            for (int o = 0; o < DOUT; ++o) {
                dh = fmaf(dout[o], w2[o][k], dh);
                gW2[o][k] = fmaf(h, dout[o], gW2[o][k]);
            }
            dh = pre > 0.f ? dh : 0.f;
            gb1[k] += dh;
#pragma unroll
            for (int i = 0; i < DIN; ++i) gW1[i][k] = fmaf(x[i], dh, gW1[i][k]);
        }
    }

    // block reduction: wave shuffle, then the four waves through LDS, fixed order
    auto put = [&](int p, float v) {
        v = wave_sum(v);
        if (lane == 0) red[wave][p] = v;
    };
#pragma unroll
    for (int k = 0; k < KC; ++k) {
#pragma unroll
        for (int i = 0; i < DIN; ++i) put(i * KC + k, gW1[i][k]);
        put(DIN * KC + k, gb1[k]);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) put((DIN + 1 + o) * KC + k, gW2[o][k]);
    }
#pragma unroll
    for (int o = 0; o < DOUT; ++o) put((DIN + 1 + DOUT) * KC + o, gb2[o]);
    __syncthreads();
    // scatter the chunk into the flat keras-layout vector dW1 (DIN,H) | db1 (H) | dW2 (H,DOUT) | db2 (DOUT)
    const int NP = DIN * H + H + H * DOUT + DOUT;
    float* dst = a.partial + (size_t)blockIdx.x * NP;
    for (int p = tid; p < NPC; p += 256) {
        const float v = (red[0][p] + red[1][p]) + (red[2][p] + red[3][p]);
        if (p < (DIN + 1 + DOUT) * KC) {
            const int row = p / KC, k = p - row * KC;
            if (row < DIN) dst[row * H + k0 + k] = v;
            else if (row == DIN) dst[DIN * H + k0 + k] = v;
            else dst[DIN * H + H + (k0 + k) * DOUT + (row - DIN - 1)] = v;
        } else if (blockIdx.y == 0) {
            dst[DIN * H + H + H * DOUT + (p - (DIN + 1 + DOUT) * KC)] = v;
        }
    }
}

// out[p] (+)= sum_blk partial[blk][p]: one wave per output element, lanes stride over the blocks
// (fixed order, deterministic)
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int nblk, int NP, float* __restrict__ out,
                                       int accumulate) {
    const int p = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) s += partial[(size_t)b * NP + p];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[p] = accumulate ? out[p] + s : s;
}

template <int DIN, int DOUT>
static int launch_wgrad(const WgradArgs& a, int H, int nblk, float* out, int accumulate, hipStream_t s) {
    const int NP = DIN * H + H + H * DOUT + DOUT;
    clear_hip_error();
    hipLaunchKernelGGL((mlp_wgrad_kernel<DIN, DOUT>), dim3(nblk, H / kKC), dim3(256), 0, s, a, H);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(NP), dim3(64), 0, s, a.partial, nblk, NP, out, accumulate);
    return launch_status();
}

template <int DIN>
static int wgrad_dispatch_out(const WgradArgs& a, int H, int Dout, int nblk, float* out, int acc, hipStream_t s) {
    switch (Dout) {
        case 1: return launch_wgrad<DIN, 1>(a, H, nblk, out, acc, s);
        case 2: return launch_wgrad<DIN, 2>(a, H, nblk, out, acc, s);
        case 3: return launch_wgrad<DIN, 3>(a, H, nblk, out, acc, s);
        case 4: return launch_wgrad<DIN, 4>(a, H, nblk, out, acc, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // namespace psvo

extern "C" int psvo_mlp_wgrad_blocks(long long rows) {
    long long nb = (rows + 256 * 8 - 1) / (256 * 8);  // >= 8 rows per lane amortise the final reduction
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    return (int)nb;
}

extern "C" int psvo_mlp_wgrad(long long S, int L, int Din, int H, int Dout, const float* X, const float* dOut,
                              const psvo_mlp* w, float* partial, float* grad, int accumulate, void* stream) {
    using namespace psvo;
    if (!X || !dOut || !w || !partial || !grad || S <= 0 || L <= 0) return PSVO_ERR_INVALID;
    WgradArgs a{S, L, X, dOut, *w, partial};
    const int nblk = psvo_mlp_wgrad_blocks(S * L);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (H <= 0 || H % kKC != 0) return PSVO_ERR_UNSUPPORTED;
    switch (Din) {
        case 1: return wgrad_dispatch_out<1>(a, H, Dout, nblk, grad, accumulate, s);
        case 2: return wgrad_dispatch_out<2>(a, H, Dout, nblk, grad, accumulate, s);
        case 3: return wgrad_dispatch_out<3>(a, H, Dout, nblk, grad, accumulate, s);
        case 4: return wgrad_dispatch_out<4>(a, H, Dout, nblk, grad, accumulate, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
