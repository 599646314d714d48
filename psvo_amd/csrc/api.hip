// Small entry points of the C ABI: version/status strings and the per-sequence ELBO reductions.
#include "common.h"

namespace psvo {

// out[b] = sum_t lse[t, b]   (SVO.compute_log_ZSMC before the batch mean, reference SVO.py:302-311)
__global__ void elbo_filter_kernel(const float* __restrict__ lse, float* __restrict__ out, int T, int B) {
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int t = threadIdx.x; t < T; t += 64) acc += lse[(size_t)t * B + b];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[b] = acc;
}

// out[b] = logsumexp_n score[b, n] - log N   (PSVO.compute_log_ZSMC, reference PSVO.py:52-67)
__global__ void elbo_bsim_kernel(const float* __restrict__ score, float* __restrict__ out, int N) {
    const int b = blockIdx.x;
    const float ninf = -__builtin_huge_valf();
    float mx = ninf;
    for (int n = threadIdx.x; n < N; n += 64) mx = fmaxf(mx, score[(size_t)b * N + n]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 64) s += expf(score[(size_t)b * N + n] - mx);
    s = wave_sum(s);
    if (threadIdx.x == 0) out[b] = mx + logf(s) - logf((float)N);
}

// out[0] = mean_b [ logsumexp_n score[b, n] - log N ]: the training objective in one launch
// (one workgroup of 4 waves, sequences strided over the waves, fixed summation order)
__global__ void __launch_bounds__(256) elbo_bsim_mean_kernel(const float* __restrict__ score, float* __restrict__ out,
                                                             int B, int N) {
    __shared__ float part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float ninf = -__builtin_huge_valf();
    float acc = 0.f;
    for (int b = wave; b < B; b += 4) {
        float mx = ninf;
        for (int n = lane; n < N; n += 64) mx = fmaxf(mx, score[(size_t)b * N + n]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int n = lane; n < N; n += 64) s += expf(score[(size_t)b * N + n] - mx);
        s = wave_sum(s);
        acc += mx + logf(s);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (part[0] + part[1] + part[2] + part[3]) / (float)B - logf((float)N);
}

// dscore[b, n] = dz[0] / B * softmax_n(score[b, :])   (reverse of elbo_bsim_mean_kernel)
__global__ void elbo_bsim_mean_bwd_kernel(const float* __restrict__ score, const float* __restrict__ dz,
                                          float* __restrict__ dscore, int B, int N) {
    const int b = blockIdx.x;
    const float ninf = -__builtin_huge_valf();
    float mx = ninf;
    for (int n = threadIdx.x; n < N; n += 64) mx = fmaxf(mx, score[(size_t)b * N + n]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 64) s += expf(score[(size_t)b * N + n] - mx);
    s = wave_sum(s);
    const float a = dz[0] / ((float)B * s);
    for (int n = threadIdx.x; n < N; n += 64) dscore[(size_t)b * N + n] = a * expf(score[(size_t)b * N + n] - mx);
}

}  // namespace psvo

namespace psvo {
thread_local hipError_t g_last_hip_error = hipSuccess;
}

extern "C" int psvo_abi_version(void) { return PSVO_ABI_VERSION; }

extern "C" const char* psvo_last_hip_error(void) { return hipGetErrorString(psvo::g_last_hip_error); }

extern "C" const char* psvo_status_string(int status) {
    switch (status) {
        case PSVO_OK: return "ok";
        case PSVO_ERR_INVALID: return "invalid argument (null pointer, non-positive size or inconsistent descriptor)";
        case PSVO_ERR_UNSUPPORTED: return "unsupported configuration (Dx/Dy/H/M/N outside the instantiated kernel set)";
        case PSVO_ERR_HIP: return "HIP runtime error at kernel launch";
        default: return "unknown status";
    }
}

extern "C" int psvo_elbo_filter(const psvo_desc* desc, const float* lse, float* out, void* stream) {
    if (!desc || !lse || !out || desc->B <= 0 || desc->T <= 0) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::elbo_filter_kernel, dim3(desc->B), dim3(64), 0, static_cast<hipStream_t>(stream), lse,
                       out, desc->T, desc->B);
    return psvo::launch_status();
}

extern "C" int psvo_elbo_bsim(const psvo_desc* desc, const float* score, float* out, void* stream) {
    if (!desc || !score || !out || desc->B <= 0 || desc->N <= 0) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::elbo_bsim_kernel, dim3(desc->B), dim3(64), 0, static_cast<hipStream_t>(stream), score,
                       out, desc->N);
    return psvo::launch_status();
}

extern "C" int psvo_elbo_bsim_mean(const psvo_desc* desc, const float* score, float* out, void* stream) {
    if (!desc || !score || !out || desc->B <= 0 || desc->N <= 0) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::elbo_bsim_mean_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), score,
                       out, desc->B, desc->N);
    return psvo::launch_status();
}

extern "C" int psvo_elbo_bsim_mean_backward(const psvo_desc* desc, const float* score, const float* dz,
                                            float* dscore, void* stream) {
    if (!desc || !score || !dz || !dscore || desc->B <= 0 || desc->N <= 0) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::elbo_bsim_mean_bwd_kernel, dim3(desc->B), dim3(64), 0, static_cast<hipStream_t>(stream),
                       score, dz, dscore, desc->B, desc->N);
    return psvo::launch_status();
}

// ---------------------------------------------------------------------------------------------
// small host-overhead killers: at C* the persistent kernels leave ~1 ms of a training step to
// ~300 tiny launches, so groups of them are fused here.
// ---------------------------------------------------------------------------------------------
namespace psvo {

// out[p] (+)= sum_r part[r * stride + p]   (one wave per output element, fixed order)
__global__ void reduce_rows_kernel(const float* __restrict__ part, int nrows, long long stride, int n,
                                   float* __restrict__ out, int accumulate) {
    const int p = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < nrows; r += 64) s += part[(size_t)r * stride + p];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[p] = accumulate ? out[p] + s : s;
}

// sigma = max(softplus(raw), min) with NaN -> 0 first (reference src/distribution/mvn.py:80-90)
__global__ void sigma_fwd_kernel(const float* __restrict__ raw, const float* __restrict__ mins, float* __restrict__ sig,
                                 int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = raw[i];
    float s = r > 20.f ? r : log1pf(expf(r));   // softplus (same threshold as torch / TF for fp32)
    if (s != s) s = 0.f;
    sig[i] = fmaxf(s, mins[i]);
}

// graw (+)= dsig * sigmoid(raw) * [softplus(raw) >= min]
__global__ void sigma_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ mins,
                                 const float* __restrict__ dsig, float* __restrict__ graw, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = raw[i];
    const float s = r > 20.f ? r : log1pf(expf(r));
    const float g = (s == s && s >= mins[i]) ? dsig[i] / (1.f + expf(-r)) : 0.f;
    graw[i] = accumulate ? graw[i] + g : g;
}

}  // namespace psvo

extern "C" int psvo_reduce_rows(const float* part, int nrows, long long stride, int n, float* out, int accumulate,
                                void* stream) {
    if (!part || !out || nrows <= 0 || n <= 0) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::reduce_rows_kernel, dim3(n), dim3(64), 0, static_cast<hipStream_t>(stream), part, nrows,
                       stride, n, out, accumulate);
    return psvo::launch_status();
}

extern "C" int psvo_sigma_forward(const float* raw, const float* mins, float* sig, int n, void* stream) {
    if (!raw || !mins || !sig || n <= 0) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::sigma_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), raw,
                       mins, sig, n);
    return psvo::launch_status();
}

extern "C" int psvo_sigma_backward(const float* raw, const float* mins, const float* dsig, float* graw, int n,
                                   int accumulate, void* stream) {
    if (!raw || !mins || !dsig || !graw || n <= 0) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::sigma_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), raw,
                       mins, dsig, graw, n, accumulate);
    return psvo::launch_status();
}


// ---------------------------------------------------------------------------------------------
// on-device self-test of the cross-lane primitives in common.h (tests/test_gpu_parity.py)
// out rows: xor 1,2,4,8,16,32 | inclusive scan | sum | max   -> 9 x 64 floats
// ---------------------------------------------------------------------------------------------
namespace psvo {
__global__ void lane_selftest_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int l = threadIdx.x;
    const float v = in[l];
    out[0 * 64 + l] = xor_lane<1>(v);
    out[1 * 64 + l] = xor_lane<2>(v);
    out[2 * 64 + l] = xor_lane<4>(v);
    out[3 * 64 + l] = xor_lane<8>(v);
    out[4 * 64 + l] = xor_lane<16>(v);
    out[5 * 64 + l] = xor_lane<32>(v);
    out[6 * 64 + l] = wave_incl_scan(v, l);
    out[7 * 64 + l] = wave_sum(v);
    out[8 * 64 + l] = wave_max(v);
}
}  // namespace psvo

#include "bsim_bwd2_impl.h"
namespace psvo {
__global__ void lane_selftest2_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int l = threadIdx.x;
    const float lo = in[l], hi = in[64 + l], ex = in[128 + l];
    out[0 * 64 + l] = swap_add32(lo, hi);
    out[1 * 64 + l] = swap_add16(lo, hi);
    out[2 * 64 + l] = row_sum16(lo);
    // D[i][j] = sum_k A[i][k] B[k][j], A operand: lane l holds A[l & 15][l >> 4]; B: B[l >> 4][l & 15];
    // D: lane l holds rows 4 (l >> 4) + reg of column l & 15
    f4v acc = f4v{0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lo, hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ex, hi, acc, 0, 0, 0);
    out[3 * 64 + l] = acc[0];
    out[4 * 64 + l] = acc[1];
}
}  // namespace psvo

namespace psvo {
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace psvo

extern "C" int psvo_debug_stamp(unsigned long long* slot, void* stream) {
    if (!slot) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::stamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), slot);
    return psvo::launch_status();
}

extern "C" int psvo_selftest_lanes2(const float* in192, float* out320, void* stream) {
    if (!in192 || !out320) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::lane_selftest2_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), in192, out320);
    return psvo::launch_status();
}

extern "C" int psvo_selftest_lanes(const float* in64, float* out576, void* stream) {
    if (!in64 || !out576) return PSVO_ERR_INVALID;
    psvo::clear_hip_error();
    hipLaunchKernelGGL(psvo::lane_selftest_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), in64, out576);
    return psvo::launch_status();
}
