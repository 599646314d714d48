// Bidirectional LSTMBlockCell layer as one persistent launch (encoder upstream of the particle path).
// Restates tf.contrib.rnn.LSTMBlockCell (forget_bias = 1, gate order i, j, f, o; reference
// src/model.py:164-176) as driven by stack_bidirectional_dynamic_rnn (reference src/SMC/SVO.py:337-341).
//
// A per-step launch sequence (GEMM + pointwise per time step and direction) costs ~800 launches
// for T = 200; here one workgroup owns one (sequence, direction) and loops over t in-kernel:
// lane g owns gate column g of the (Din + Dh, 4 Dh) kernel in registers, h_{t-1} and x_t are LDS
// broadcasts, gates are exchanged through LDS once per step.
#include "common.h"

namespace psvo {

struct LstmArgs {
    int B, T, Din;
    const float *x, *Wf, *bf, *Wb, *bb;
    float *out, *cs, *gates;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

template <int DINP, int DH>
__global__ void __launch_bounds__(4 * DH) bilstm_fwd_kernel(const LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[DINP];
    __shared__ __attribute__((aligned(16))) float hs[DH];
    __shared__ float zs[4 * DH];

    const int g = threadIdx.x;
    const int b = blockIdx.x, dir = blockIdx.y;
    const int T = a.T, Din = a.Din, B = a.B;
    const float* W = dir ? a.Wb : a.Wf;
    const float* bias = dir ? a.bb : a.bf;

    float wx[DINP], wh[DH];
#pragma unroll
    for (int i = 0; i < DINP; ++i) wx[i] = i < Din ? W[(size_t)i * 4 * DH + g] : 0.f;
#pragma unroll
    for (int k = 0; k < DH; ++k) wh[k] = W[(size_t)(Din + k) * 4 * DH + g];
    // forget_bias = 1 is added to the f gate (third block of columns)
    const float bg = bias[g] + ((g >= 2 * DH && g < 3 * DH) ? 1.f : 0.f);

    const float* xb = a.x + (size_t)b * T * Din;
    int t = dir ? T - 1 : 0;
    const int dt = dir ? -1 : 1;
    constexpr int NTH = 4 * DH;
    constexpr int XR = (DINP + NTH - 1) / NTH;  // x entries staged per thread
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int i = g + r * NTH;
        if (i < DINP) xs[i] = i < Din ? xb[(size_t)t * Din + i] : 0.f;
    }
    if (g < DH) hs[g] = 0.f;
    float c = 0.f;
    __syncthreads();

    for (int s = 0; s < T; ++s, t += dt) {
        float xn[XR];
        const int tn = t + dt;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int i = g + r * NTH;
            xn[r] = (s + 1 < T && i < Din) ? xb[(size_t)tn * Din + i] : 0.f;
        }

        float z = bg;
#pragma unroll
        for (int i = 0; i < DINP; i += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xs + i);
            z = fmaf(v.x, wx[i], z);
            z = fmaf(v.y, wx[i + 1], z);
            z = fmaf(v.z, wx[i + 2], z);
            z = fmaf(v.w, wx[i + 3], z);
        }
#pragma unroll
        for (int k = 0; k < DH; k += 4) {
            const float4 v = *reinterpret_cast<const float4*>(hs + k);
            z = fmaf(v.x, wh[k], z);
            z = fmaf(v.y, wh[k + 1], z);
            z = fmaf(v.z, wh[k + 2], z);
            z = fmaf(v.w, wh[k + 3], z);
        }
        // activation per gate block: i, f, o sigmoid; j tanh
        const float act = (g >= DH && g < 2 * DH) ? tanhf(z) : sigmoidf_(z);
        zs[g] = act;
        __syncthreads();
        if (g < DH) {
            const float gi = zs[g], gj = zs[DH + g], gf = zs[2 * DH + g], go = zs[3 * DH + g];
            c = c * gf + gi * gj;
            const float h = tanhf(c) * go;
            hs[g] = h;
            a.out[((size_t)b * T + t) * 2 * DH + dir * DH + g] = h;
            if (a.cs) a.cs[(((size_t)dir * B + b) * T + t) * DH + g] = c;
        }
        if (a.gates) a.gates[(((size_t)dir * B + b) * T + t) * 4 * DH + g] = act;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int i = g + r * NTH;
            if (i < DINP) xs[i] = xn[r];
        }
        __syncthreads();
    }
}

template <int DINP>
static int lstm_dispatch_dh(const LstmArgs& a, int Dh, hipStream_t s) {
    dim3 grid(a.B, 2);
    clear_hip_error();
    switch (Dh) {
        case 8: hipLaunchKernelGGL((bilstm_fwd_kernel<DINP, 8>), grid, dim3(32), 0, s, a); break;
        case 16: hipLaunchKernelGGL((bilstm_fwd_kernel<DINP, 16>), grid, dim3(64), 0, s, a); break;
        case 32: hipLaunchKernelGGL((bilstm_fwd_kernel<DINP, 32>), grid, dim3(128), 0, s, a); break;
        case 64: hipLaunchKernelGGL((bilstm_fwd_kernel<DINP, 64>), grid, dim3(256), 0, s, a); break;
        default: return PSVO_ERR_UNSUPPORTED;
    }
    return launch_status();
}

}  // namespace psvo

extern "C" int psvo_bilstm_forward(int B, int T, int Din, int Dh, const float* x, const float* W_fw,
                                   const float* b_fw, const float* W_bw, const float* b_bw, float* out,
                                   float* cs, float* gates, void* stream) {
    using namespace psvo;
    if (!x || !W_fw || !b_fw || !W_bw || !b_bw || !out) return PSVO_ERR_INVALID;
    if (B <= 0 || T <= 0 || Din <= 0 || Dh <= 0 || B > 65535) return PSVO_ERR_INVALID;
    LstmArgs a{B, T, Din, x, W_fw, b_fw, W_bw, b_bw, out, cs, gates};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Din <= 4) return lstm_dispatch_dh<4>(a, Dh, s);
    if (Din <= 16) return lstm_dispatch_dh<16>(a, Dh, s);
    if (Din <= 32) return lstm_dispatch_dh<32>(a, Dh, s);
    if (Din <= 64) return lstm_dispatch_dh<64>(a, Dh, s);
    if (Din <= 128) return lstm_dispatch_dh<128>(a, Dh, s);
    return PSVO_ERR_UNSUPPORTED;
}
