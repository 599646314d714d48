// Which hardware slots do co-resident workgroups get?  512 workgroups of 256 threads with 78 KB of LDS each (two per CU),
// every workgroup records HW_REG_HW_ID and XCC_ID and the shader clock at entry.   hipcc --offload-arch=gfx950 -O2 hwid.hip -o hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) probe(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = clock64();
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) { lds[threadIdx.x] = acc; __syncthreads(); acc += lds[(threadIdx.x + 1) & 255] * 1e-9f; }
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 0] = hw;
        out[4 * blockIdx.x + 1] = xcc;
        out[4 * blockIdx.x + 2] = (unsigned)(t0 & 0xffffffffu);
        out[4 * blockIdx.x + 3] = __float_as_uint(acc);
    }
}
int main() {
    const int nb = 512;
    unsigned* d;
    hipMalloc(&d, nb * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 78 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    int tg_hist[16] = {0};
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[4 * b], xcc = h[4 * b + 1] & 0xf;
        const unsigned wave = hw & 0xf, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 0xf;
        per_cu[(xcc << 8) | (se << 5) | (sh << 4) | cu].push_back((int)tg);
        tg_hist[tg]++;
        if (b < 8) printf("wg %3d: xcc %u se %u sh %u cu %2u simd %u wave %u tg %u\n", b, xcc, se, sh, cu, simd, wave, tg);
    }
    printf("distinct CUs seen: %zu\n", per_cu.size());
    int both_same = 0, two = 0;
    for (auto& kv : per_cu) if (kv.second.size() == 2) { ++two; both_same += (kv.second[0] & 1) == (kv.second[1] & 1); }
    printf("CUs with two workgroups: %d, of which both have the same TG_ID parity: %d\n", two, both_same);
    printf("TG_ID histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", tg_hist[i]); printf("\n");
    return 0;
}
