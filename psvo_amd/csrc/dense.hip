// Dense layers over plain rows on the f32 matrix instruction: the hoisted per-(sequence, step) networks of the path
// (q0, q2, BSim_q2, BSim_q_init: tf_mvn.mean of MLP_transformation, reference src/transformation/MLP.py:24-68) when they
// have MORE THAN ONE hidden layer or are wider than psvo_rows_mlp_* covers (`q2_layers="64,64"`, `y_smoother_Dhs=128` ->
// 256 inputs).  These are the only GEMM-shaped contractions of the path -- R = B*T rows, K = Din up to a few hundred,
// N = H -- so they go to v_mfma_f32_16x16x4_f32 (exact f32, the reference's arithmetic type) through LDS tiles:
//   forward : Y = act(X W + b)                                   psvo_dense_forward
//   backward: dZ = dY * [Y > 0] (relu) ;  dX = dZ W^T ;  [dW ; db] = [X | 1]^T dZ        psvo_dense_backward
// One 256-thread workgroup computes a 64 x 64 tile of the output; wave w owns rows 16 w .. 16 w + 15 of it as four
// 16 x 16 accumulators; K is walked in LDS-staged slabs of 16.  Operand layout of the instruction (lane l): A[i = l & 15]
// [k = l >> 4], B[k = l >> 4][j = l & 15], D[i = 4 (l >> 4) + reg][j = l & 15].  The weight-gradient product has K = R
// (thousands): it is split over row blocks into per-workgroup partials that a fixed-order reduction folds (no atomics).
#include "common.h"

namespace psvo {

typedef float dense_f4 __attribute__((ext_vector_type(4)));

struct DenseArgs {
    // C (M x N) = A' (M x K) B' (K x N), K restricted to [k0 + blockIdx.z * kchunk, +kchunk)
    const float* A; long long lda;   // TA = 0: A'[m][k] = A[m * lda + k];  TA = 1: A'[m][k] = A[k * lda + m] (and a row of ones at m == M - 1 when ONES)
    const float* B; long long ldb;   // TB = 0: B'[k][n] = B[k * ldb + n];  TB = 1: B'[k][n] = B[n * ldb + k]
    const float* mask; long long ldm;   // relu mask source (same shape as the masked operand) or null
    const float* bias;               // added per column n before the activation (forward) or null
    float* C; long long ldc;         // C[m * ldc + n]; with split K: C + blockIdx.z * M * N (dense partials)
    int M, N, K, kchunk, relu;
};

// MASKED: 0 none, 1 the A operand is multiplied by [mask > 0] (dX = (dY * mask) W^T), 2 the B operand (dW = X^T (dY * mask))
template <int TA, int TB, int MASKED, int ONES>
__global__ void __launch_bounds__(256) dense_tile_kernel(const DenseArgs a) {
    __shared__ __attribute__((aligned(16))) float As[64][17];   // [m][k]  (padded: the operand read walks m across lanes)
    __shared__ __attribute__((aligned(16))) float Bs[16][68];   // [k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int kbeg = blockIdx.z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    dense_f4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = dense_f4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        // ---- stage the slabs: 64 x 16 of A', 16 x 64 of B' (1024 elements each, 4 per thread), zero outside the matrices ----
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            {   // A' slab: consecutive threads along the contiguous axis of A
                const int mm = TA ? (idx & 63) : (idx >> 4), kk = TA ? (idx >> 6) : (idx & 15);
                const int m = m0 + mm, k = k0 + kk;
                float v = 0.f;
                if (m < a.M && k < kend) {
                    if (ONES && m == a.M - 1) v = 1.f;
                    else {
                        const long long off = TA ? (long long)k * a.lda + m : (long long)m * a.lda + k;
                        v = a.A[off];
                        if (MASKED == 1) v = a.mask[(long long)m * a.ldm + k] > 0.f ? v : 0.f;
                    }
                }
                As[mm][kk] = v;
            }
            {   // B' slab
                const int nn = TB ? (idx >> 4) : (idx & 63), kk = TB ? (idx & 15) : (idx >> 6);
                const int n = n0 + nn, k = k0 + kk;
                float v = 0.f;
                if (n < a.N && k < kend) {
                    const long long off = TB ? (long long)n * a.ldb + k : (long long)k * a.ldb + n;
                    v = a.B[off];
                    if (MASKED == 2) v = a.mask[(long long)k * a.ldm + n] > 0.f ? v : 0.f;
                }
                Bs[kk][nn] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const float av = As[wave * 16 + (lane & 15)][kq * 4 + (lane >> 4)];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float bv = Bs[kq * 4 + (lane >> 4)][c * 16 + (lane & 15)];
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* C = a.C + (size_t)blockIdx.z * a.M * a.N;     // (split K: one dense M x N partial per slice)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = n0 + c * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * 16 + 4 * (lane >> 4) + r;
            if (m < a.M && n < a.N) {
                float v = acc[c][r];
                if (a.bias) v += a.bias[n];
                if (a.relu) v = fmaxf(v, 0.f);
                C[(long long)m * a.ldc + n] = v;
            }
        }
    }
}

// out[p] (+)= sum_z partial[z][p]  (fixed order)
__global__ void dense_fold_kernel(const float* __restrict__ partial, int nz, int n, float* __restrict__ out, int accumulate) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += partial[(size_t)z * n + p];
    out[p] = accumulate ? out[p] + s : s;
}

static inline int dense_wgrad_chunks(long long R) {
    // split-K slices of >= 128 rows, at most 256 of them: the product has few output tiles ((Din + 1) x Dout), so the rows
    // are what fills the chip (R = 6400: 50 slices; 2048-row slices left 8 workgroups walking 100 slabs each: 150 us)
    long long nz = (R + 127) / 128;
    if (nz > 256) nz = 256;
    if (nz < 1) nz = 1;
    return (int)nz;
}

}  // namespace psvo

extern "C" int psvo_dense_wgrad_slices(long long R) { return psvo::dense_wgrad_chunks(R); }

extern "C" int psvo_dense_forward(long long R, int Din, int Dout, const float* X, const float* W, const float* b, int relu,
                                  float* Y, void* stream) {
    using namespace psvo;
    if (!X || !W || !Y || R <= 0 || Din <= 0 || Dout <= 0) return PSVO_ERR_INVALID;
    if (R > (1ll << 30) || Din > 4096 || Dout > 4096) return PSVO_ERR_UNSUPPORTED;
    DenseArgs a{X, Din, W, Dout, nullptr, 0, b, Y, Dout, (int)R, Dout, Din, Din, relu ? 1 : 0};
    a.kchunk = ((Din + 15) / 16) * 16;
    clear_hip_error();
    hipLaunchKernelGGL((dense_tile_kernel<0, 0, 0, 0>), dim3((unsigned)((R + 63) / 64), (Dout + 63) / 64, 1), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return launch_status();
}

// grad = [dW (Din x Dout) | db (Dout)] ; partial: psvo_dense_wgrad_slices(R) * (Din + 1) * Dout floats
extern "C" int psvo_dense_backward(long long R, int Din, int Dout, const float* X, const float* Y, const float* dY,
                                   const float* W, int relu, float* dX, float* partial, float* grad, int accumulate,
                                   void* stream) {
    using namespace psvo;
    if (!X || !dY || !W || !partial || !grad || (relu && !Y) || R <= 0 || Din <= 0 || Dout <= 0) return PSVO_ERR_INVALID;
    if (R > (1ll << 30) || Din > 4096 || Dout > 4096) return PSVO_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    clear_hip_error();
    if (dX) {   // dX (R x Din) = (dY * mask) (R x Dout) . W^T : B'[k = o][n = i] = W[i * Dout + o]
        DenseArgs a{dY, Dout, W, Dout, relu ? Y : nullptr, Dout, nullptr, dX, Din, (int)R, Din, Dout, 0, 0};
        a.kchunk = ((Dout + 15) / 16) * 16;
        if (relu)
            hipLaunchKernelGGL((dense_tile_kernel<0, 1, 1, 0>), dim3((unsigned)((R + 63) / 64), (Din + 63) / 64, 1),
                               dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((dense_tile_kernel<0, 1, 0, 0>), dim3((unsigned)((R + 63) / 64), (Din + 63) / 64, 1),
                               dim3(256), 0, s, a);
    }
    {   // [dW ; db] ((Din + 1) x Dout) = [X | 1]^T (dY * mask): split over row slices
        const int nz = dense_wgrad_chunks(R);
        long long chunk = (R + nz - 1) / nz;
        chunk = ((chunk + 15) / 16) * 16;
        DenseArgs a{X, Din, dY, Dout, relu ? Y : nullptr, Dout, nullptr, partial, Dout, Din + 1, Dout, (int)R, (int)chunk, 0};
        const dim3 grid((Din + 1 + 63) / 64, (Dout + 63) / 64, nz);
        if (relu) hipLaunchKernelGGL((dense_tile_kernel<1, 0, 2, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((dense_tile_kernel<1, 0, 0, 1>), grid, dim3(256), 0, s, a);
        const int n = (Din + 1) * Dout;
        hipLaunchKernelGGL(dense_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, nz, n, grad, accumulate);
    }
    return launch_status();
}
