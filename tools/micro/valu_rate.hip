// What does one SIMD of gfx950 sustain in f32 FMAs from the vector ALU -- scalar v_fma_f32 against packed v_pk_fma_f32 -- at
// one, two and four waves per SIMD?   hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate && ./valu_rate
// Each wave runs ITER iterations of 32 independent accumulator updates (32 v_fma_f32, or 16 v_pk_fma_f32 on register pairs);
// the grid is 256 CUs x 4 SIMDs x W waves, timed with events, and s_memtime brackets the loop of every wave (shader cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PACKED>
__global__ void __launch_bounds__(256) rate(float* out, unsigned long long* cyc, int iters, float a, float b) {
    float acc[32];
    f2 accp[16];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
    for (int i = 0; i < 16; ++i) accp[i] = f2{acc[2 * i], acc[2 * i + 1]};
    f2 ap = f2{a, a}, bp = f2{b, b};
    asm volatile("" : "+v"(ap), "+v"(bp), "+v"(a), "+v"(b));
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
        if constexpr (PACKED) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(accp[i]) : "v"(accp[i]), "v"(ap), "v"(bp));
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(acc[i]) : "v"(acc[i]), "v"(a), "v"(b));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += PACKED ? (i & 1 ? accp[i >> 1].y : accp[i >> 1].x) : acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    const int iters = 20000;
    for (int W : {1, 2, 4, 8}) {
        const int nb = 256 * W;      // 256-thread workgroups = four waves, one per SIMD of a CU when evenly placed
        float* out;
        unsigned long long* cyc;
        hipMalloc(&out, nb * 256 * 4);
        hipMalloc(&cyc, nb * 4 * 8);
        for (int packed = 0; packed < 2; ++packed) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (packed) hipLaunchKernelGGL(rate<1>, dim3(nb), dim3(256), 0, 0, out, cyc, iters, 0.999f, 0.001f);
                else hipLaunchKernelGGL(rate<0>, dim3(nb), dim3(256), 0, 0, out, cyc, iters, 0.999f, 0.001f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(nb * 4);
            hipMemcpy(h.data(), cyc, nb * 4 * 8, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto v : h) mean += v;
            mean /= h.size();
            const double insts = (double)iters * (packed ? 16 : 32);
            const double fma_total = (double)nb * 256 * iters * 32;
            // s_memtime counts at a fixed 100 MHz on this part; the event time gives wall clock
            printf("waves/SIMD %d  %-14s  %.3f ms  %.1f TFLOP/s  %.2f ns per wave instruction per SIMD  (s_memtime ticks per wave: %.0f)\n",
                   W, packed ? "v_pk_fma_f32" : "v_fma_f32", ms, 2.0 * fma_total / (ms * 1e-3) / 1e12,
                   ms * 1e6 / (insts * W), mean);
        }
        hipFree(out);
        hipFree(cyc);
    }
    return 0;
}
