// Back-propagation through time of one bidirectional LSTMBlockCell layer, one persistent workgroup
// per (sequence, direction) -- the reverse of psvo_bilstm_forward (lstm.hip), which saved the cell
// states and the activated gates.  The reference gets this from TensorFlow autodiff of
// stack_bidirectional_dynamic_rnn (reference src/SMC/SVO.py:337-341, src/trainer.py:115-118).
//
// Thread g of the 4*Dh-thread workgroup plays two roles per step:
//   column role: accumulates dW[:, g] += [x_t, h_{t-1}] * dz[g] in registers (Din + Dh accumulators);
//   row role   : row p of the kernel is held in registers and produces d[x_t, h_{t-1}][p] =
//                sum_g dz[g] W[p][g]  (d x_t goes to HBM, d h_{t-1} stays in LDS for the next step).
// Weight-gradient partials are written per (sequence, direction) and summed over sequences by the host.
#include "common.h"

namespace psvo {

struct LstmBwdArgs {
    int B, T, Din;
    const float *x, *Wf, *Wb, *out, *cs, *gates, *dout;
    float *dx_part, *dW_part, *db_part;
};

template <int DINP, int DH>
__global__ void __launch_bounds__(4 * DH) bilstm_bwd_kernel(const LstmBwdArgs a) {
    constexpr int NTH = 4 * DH;
    constexpr int K = DINP + DH;                 // padded rows of the kernel
    constexpr int RPT = (K + NTH - 1) / NTH;     // rows per thread in the row role
    __shared__ __attribute__((aligned(16))) float xh[K];
    __shared__ __attribute__((aligned(16))) float dz[NTH];
    __shared__ float dhrec[DH];

    const int g = threadIdx.x;
    const int b = blockIdx.x, dir = blockIdx.y;
    const int T = a.T, Din = a.Din, B = a.B;
    const float* W = dir ? a.Wb : a.Wf;
    const int dtf = dir ? -1 : 1;  // forward-order time increment of this direction

    // real kernel row of padded row p (-1: padding)
    auto real_row = [&](int p) { return p < DINP ? (p < Din ? p : -1) : Din + (p - DINP); };

    float wr[RPT][NTH];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int p = g + r * NTH;
        const int row = p < K ? real_row(p) : -1;
#pragma unroll
        for (int c = 0; c < NTH; ++c) wr[r][c] = row >= 0 ? W[(size_t)row * NTH + c] : 0.f;
    }
    float dWc[K], dbc = 0.f;
#pragma unroll
    for (int p = 0; p < K; ++p) dWc[p] = 0.f;
    if (g < DH) dhrec[g] = 0.f;
    float dcrec = 0.f;

    const float* xb = a.x + (size_t)b * T * Din;
    const float* ob = a.out + (size_t)b * T * 2 * DH + dir * DH;
    const float* dob = a.dout + (size_t)b * T * 2 * DH + dir * DH;
    const float* csb = a.cs + ((size_t)dir * B + b) * T * DH;
    const float* gb = a.gates + ((size_t)dir * B + b) * T * NTH;
    float* dxb = a.dx_part + ((size_t)dir * B + b) * T * Din;
    __syncthreads();

    // Everything a step reads from HBM was saved by the forward pass, so it is requested one step ahead (issue only:
    // no arithmetic on the loaded values here) -- otherwise every one of the T dependent steps starts with an exposed
    // HBM round trip, which was most of this kernel's time.
    struct StepIn {
        float xv[RPT];                       // rows of [x_t, h_{t-1}] this thread stages
        float gi, gj, gf, go, c, cprev, dout;
    };
    auto load_step = [&](int s, StepIn& in) {
        const int t = dir ? s : T - 1 - s;
        const int tprev = t - dtf;
        const bool has_prev = dir ? (t < T - 1) : (t > 0);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int p = g + r * NTH;
            float v = 0.f;
            if (p < DINP) {
                if (p < Din) v = xb[(size_t)t * Din + p];
            } else if (p < K && has_prev) {
                v = ob[(size_t)tprev * 2 * DH + (p - DINP)];
            }
            in.xv[r] = v;
        }
        if (g < DH) {
            in.gi = gb[(size_t)t * NTH + g];
            in.gj = gb[(size_t)t * NTH + DH + g];
            in.gf = gb[(size_t)t * NTH + 2 * DH + g];
            in.go = gb[(size_t)t * NTH + 3 * DH + g];
            in.c = csb[(size_t)t * DH + g];
            in.cprev = has_prev ? csb[(size_t)tprev * DH + g] : 0.f;
            in.dout = dob[(size_t)t * 2 * DH + g];
        }
    };
    StepIn cur;
    load_step(0, cur);

    // reverse of the direction's forward order
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s;
        StepIn nxt = cur;
        if (s + 1 < T) load_step(s + 1, nxt);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int p = g + r * NTH;
            if (p < K) xh[p] = cur.xv[r];
        }
        if (g < DH) {
            const float gi = cur.gi, gj = cur.gj, gf = cur.gf, go = cur.go;
            const float c = cur.c, cprev = cur.cprev;
            const float dh = cur.dout + dhrec[g];
            const float tc = tanhf(c);
            const float dc = dcrec + dh * go * (1.f - tc * tc);
            dz[g] = dc * gj * gi * (1.f - gi);
            dz[DH + g] = dc * gi * (1.f - gj * gj);
            dz[2 * DH + g] = dc * cprev * gf * (1.f - gf);
            dz[3 * DH + g] = dh * tc * go * (1.f - go);
            dcrec = dc * gf;
        }
        __syncthreads();
        // column role
        const float my = dz[g];
        dbc += my;
#pragma unroll
        for (int p = 0; p < K; p += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xh + p);
            dWc[p] = fmaf(v.x, my, dWc[p]);
            dWc[p + 1] = fmaf(v.y, my, dWc[p + 1]);
            dWc[p + 2] = fmaf(v.z, my, dWc[p + 2]);
            dWc[p + 3] = fmaf(v.w, my, dWc[p + 3]);
        }
        // row role
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int p = g + r * NTH;
            if (p < K) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < NTH; c += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(dz + c);
                    acc = fmaf(v.x, wr[r][c], acc);
                    acc = fmaf(v.y, wr[r][c + 1], acc);
                    acc = fmaf(v.z, wr[r][c + 2], acc);
                    acc = fmaf(v.w, wr[r][c + 3], acc);
                }
                if (p < DINP) {
                    if (p < Din) dxb[(size_t)t * Din + p] = acc;
                } else {
                    dhrec[p - DINP] = acc;
                }
            }
        }
        __syncthreads();
        cur = nxt;
    }
    // partial weight gradients of this (sequence, direction)
    float* dWp = a.dW_part + ((size_t)b * 2 + dir) * (size_t)(Din + DH) * NTH;
#pragma unroll
    for (int p = 0; p < K; ++p) {
        const int row = real_row(p);
        if (row >= 0) dWp[(size_t)row * NTH + g] = dWc[p];
    }
    a.db_part[((size_t)b * 2 + dir) * NTH + g] = dbc;
}

template <int DINP>
static int lstm_bwd_dispatch_dh(const LstmBwdArgs& a, int Dh, hipStream_t s) {
    dim3 grid(a.B, 2);
    clear_hip_error();
    switch (Dh) {
        case 8: hipLaunchKernelGGL((bilstm_bwd_kernel<DINP, 8>), grid, dim3(32), 0, s, a); break;
        case 16: hipLaunchKernelGGL((bilstm_bwd_kernel<DINP, 16>), grid, dim3(64), 0, s, a); break;
        case 32: hipLaunchKernelGGL((bilstm_bwd_kernel<DINP, 32>), grid, dim3(128), 0, s, a); break;
        case 64: hipLaunchKernelGGL((bilstm_bwd_kernel<DINP, 64>), grid, dim3(256), 0, s, a); break;
        default: return PSVO_ERR_UNSUPPORTED;
    }
    return launch_status();
}

// [kernel (K, 4Dh) | bias (4Dh)] of each direction (+)= sum over the B per-sequence partials, fixed order; one launch
__global__ void bilstm_wgrad_fold_kernel(const float* __restrict__ dW_part, const float* __restrict__ db_part, int B, int KG,
                                         int G4, float* __restrict__ g_fw, float* __restrict__ g_bw, int accumulate) {
    const int per = KG + G4;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * per) return;
    const int dir = e / per, p = e - dir * per;
    float s = 0.f;
    if (p < KG) {
        for (int b = 0; b < B; ++b) s += dW_part[((size_t)b * 2 + dir) * KG + p];
    } else {
        for (int b = 0; b < B; ++b) s += db_part[((size_t)b * 2 + dir) * G4 + (p - KG)];
    }
    float* g = dir ? g_bw : g_fw;
    g[p] = accumulate ? g[p] + s : s;
}

}  // namespace psvo

extern "C" int psvo_bilstm_wgrad_fold(int B, int Din, int Dh, const float* dW_part, const float* db_part, float* g_fw,
                                      float* g_bw, int accumulate, void* stream) {
    using namespace psvo;
    if (!dW_part || !db_part || !g_fw || !g_bw || B <= 0 || Din <= 0 || Dh <= 0) return PSVO_ERR_INVALID;
    const int G4 = 4 * Dh, KG = (Din + Dh) * G4, n = 2 * (KG + G4);
    clear_hip_error();
    hipLaunchKernelGGL(bilstm_wgrad_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dW_part, db_part, B, KG, G4, g_fw, g_bw, accumulate);
    return launch_status();
}

extern "C" int psvo_bilstm_backward(int B, int T, int Din, int Dh, const float* x, const float* W_fw,
                                    const float* W_bw, const float* out, const float* cs, const float* gates,
                                    const float* dout, float* dx_part, float* dW_part, float* db_part, void* stream) {
    using namespace psvo;
    if (!x || !W_fw || !W_bw || !out || !cs || !gates || !dout || !dx_part || !dW_part || !db_part)
        return PSVO_ERR_INVALID;
    if (B <= 0 || T <= 0 || Din <= 0 || Dh <= 0 || B > 65535) return PSVO_ERR_INVALID;
    LstmBwdArgs a{B, T, Din, x, W_fw, W_bw, out, cs, gates, dout, dx_part, dW_part, db_part};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Din <= 4) return lstm_bwd_dispatch_dh<4>(a, Dh, s);
    if (Din <= 16) return lstm_bwd_dispatch_dh<16>(a, Dh, s);
    if (Din <= 32) return lstm_bwd_dispatch_dh<32>(a, Dh, s);
    if (Din <= 64) return lstm_bwd_dispatch_dh<64>(a, Dh, s);
    if (Din <= 128) return lstm_bwd_dispatch_dh<128>(a, Dh, s);
    return PSVO_ERR_UNSUPPORTED;
}
