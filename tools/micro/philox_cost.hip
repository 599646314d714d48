// Cost of generating the backward simulation's noise in the kernels instead of reading it (review item 6):
// Philox4x32-10 keyed by a seed, counter = item index, then two Box-Muller pairs -> four standard normals per call (one call
// per (chain, sub-particle) lane and time step covers Dx <= 4).  Measures, on the GPU, VALU-bound throughput of exactly that
// device function with all SIMDs busy, and prints cycles per call per wave -- to be set against the 5 846 VALU-active cycles
// per wave and step of bsim_bwd2 and the ~3 000 of bsim_fwd (profiles/r02_sq_counters_all_Cstar.json).
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/philox_cost tools/micro/philox_cost.hip && /tmp/philox_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ void normals4(uint32_t idx, uint32_t t, uint32_t seed, float (&z)[4]) {
    uint32_t u[4];
    philox4x32_10(idx, t, 0u, 0u, seed, 0x5eedu, u);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = (float)(u[2 * i] >> 8) * (1.0f / 16777216.0f) + (0.5f / 16777216.0f);   // (0, 1)
        const float b = (float)(u[2 * i + 1] >> 8) * (1.0f / 16777216.0f);
        const float r = __builtin_sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(a));   // v_log_f32 is log2
        const float ph = b;                                                                         // revolutions: v_sin / v_cos take them
        z[2 * i] = r * __builtin_amdgcn_sinf(ph);
        z[2 * i + 1] = r * __builtin_amdgcn_cosf(ph);
    }
}

__global__ void __launch_bounds__(256) gen(float* out, int steps, uint32_t seed) {
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (int t = 0; t < steps; ++t) {
        float z[4];
        normals4(idx, (uint32_t)t, seed, z);
        s += (z[0] + z[1]) + (z[2] + z[3]);
    }
    out[idx] = s;
}

__global__ void __launch_bounds__(256) rd(const float4* __restrict__ in, float* out, int steps, int n) {
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (int t = 0; t < steps; ++t) {
        const float4 v = in[(size_t)t * n + idx];
        s += (v.x + v.y) + (v.z + v.w);
    }
    out[idx] = s;
}

int main() {
    const int blocks = 256 * 8, n = blocks * 256, steps = 200;      // 8 192 waves on 1 024 SIMDs: every SIMD's issue slots are full
    const double waves_per_simd = blocks * 4 / 1024.0;
    float *out, *in;
    hipMalloc(&out, n * sizeof(float));
    hipMalloc(&in, (size_t)steps * n * 4 * sizeof(float));
    hipMemset(in, 0, (size_t)steps * n * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(gen, dim3(blocks), dim3(256), 0, 0, out, steps, 1234u);
            else hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, (const float4*)in, out, steps, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // a SIMD issues for one wave at a time: issue cycles per call and wave = time * clock / (steps * waves per SIMD)
        const double cyc = best * 1e-3 * 2.4e9 / (steps * waves_per_simd);
        printf("%s: %.3f ms for %d lanes x %d steps (four normals each) = %.1f cycles per call and wave at 2.4 GHz%s\n",
               which == 0 ? "philox4x32-10 + Box-Muller" : "read of four floats from HBM   ", best, n, steps, cyc,
               which == 1 ? " -- i.e. HBM time, not issue slots" : "");
    }
    float h; hipMemcpy(&h, out, sizeof(float), hipMemcpyDeviceToHost);
    printf("(checksum %g)\n", h);
    return 0;
}
