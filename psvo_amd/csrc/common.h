// Device-side helpers shared by the filter and backward-simulation kernels (gfx950 only).
#pragma once
#include <cstddef>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psvo_hip.h"

// Hidden layers of the per-particle MLPs a translation unit is compiled for (psvo_desc.layers).  The persistent kernels are
// written once; `X_l2.hip` is `#define PSVO_L 2` + `#include "X.hip"`.  Everything that depends on the MLP depth lives in the
// inline namespace psvo::l1 / psvo::l2 (distinct symbols, no template parameter threaded through every kernel), and the C
// entry point of the L = 1 unit forwards to the hidden `<entry>_l2` of the L = 2 unit when desc->layers == 2.
#ifndef PSVO_L
#define PSVO_L 1
#endif
#if PSVO_L == 1
#define PSVO_LNS l1
#define PSVO_ENTRY(name) extern "C" int name
#define PSVO_L2_DECL(name) extern "C" __attribute__((visibility("hidden"))) decltype(name) name##_l2;
#elif PSVO_L == 2
#define PSVO_LNS l2
#define PSVO_ENTRY(name) extern "C" __attribute__((visibility("hidden"))) int name##_l2
#define PSVO_L2_DECL(name)
#else
#error "PSVO_L must be 1 or 2"
#endif

namespace psvo {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kHalfLog2Pi = 0.9189385332046727f;
constexpr int kWave = 64;

// Launch-status helper.  hipGetLastError() is sticky across *any* earlier HIP call of the process
// (torch's own probing calls included), so clear it before the launch and read it after.
extern thread_local hipError_t g_last_hip_error;
inline void clear_hip_error() { (void)hipGetLastError(); }
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) g_last_hip_error = e;
    return e == hipSuccess ? PSVO_OK : PSVO_ERR_HIP;
}

// psvo_desc.layers: 0 or 1 = one hidden layer, 2 = two; anything else (three or more layers, or the garbage an ABI-3 caller's
// shorter struct leaves there) is refused loudly by every entry point instead of silently running the one-layer kernels
inline bool desc_layers_ok(const psvo_desc* d) { return !d || (d->layers >= 0 && d->layers <= 2); }

// PSVO_TUNE_L2_SPLIT (psvo_set_tuning): two-layer builds of the backward-simulation kernels -- 0 (default): one lane per
// (chain, m) whatever the problem size, 1: spread a chain over 2 M lanes when that gives two waves per SIMD, as the one-layer
// builds do.  (With the H x H layer on the matrix pipe a second wave per SIMD shares that pipe and the per-row overheads
// double: measured)
extern int g_tune_l2_split;
// PSVO_TUNE_FILTER_BWD (psvo_set_tuning): 1 (default) = the reverse filter as an affine scan where it applies (bootstrap wiring
// with resampling, one hidden layer: filter_bwd.hip) AND pays, 2 = wherever it applies, 0 = the persistent reverse kernels always
extern int g_tune_filter_bwd_scan;

// PSVO_TUNE_SKEW (psvo_set_tuning): phase offset between the workgroups that share a CU, in per cent of the kernel's own
// estimate of its pair-phase length (0 = off = default).  See phase_skew().
extern int g_tune_skew_pct;

// Two workgroups that share a CU run the same program on the same amount of data, so they stay IN PHASE for the whole
// launch: both are in their VALU-dense pair loop at the same time (each then runs at half rate -- the section timers of
// bsim_bwd2 show the pair phase at 2 x its arithmetic floor) and both are in the latency-bound rest of the step at the same
// time (vector ALU idle).  Delaying every other resident workgroup ONCE, at the start of the kernel, by about one pair-phase
// length puts the two out of phase for good -- their step times are equal, so the offset persists -- and the pair loop of one
// then runs under the latency of the other.  (MEASURED: no effect, C* and C5 alike -- both kernels are VALU-bound, and the
// vector ALU does the same work whatever the phase; profiles/r03_bsim_bwd_C5_ab.md.  Kept as a knob.)  Which workgroup waits is read from the hardware: TG_ID of HW_REG_HW_ID is the
// workgroup's slot on its CU (the same for all its waves), so the two residents of a CU always differ in it.
__device__ __forceinline__ void phase_skew(int cycles) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (cycles > 0 && ((hw >> 16) & 1u)) {
        for (int i = 0; i < cycles; i += 64 * 32) __builtin_amdgcn_s_sleep(32);      // (s_sleep n: 64 n cycles)
    }
}

// an MLP argument carries the second hidden layer when this unit is compiled for two (NULL arguments pass: optional MLPs)
inline bool mlp_layers_ok(const psvo_mlp* m) { return PSVO_L == 1 || !m || (m->Wh && m->bh); }

// PSVOwR: workgroups per sequence (cluster size) -- as many as keep every workgroup busy and the whole
// cooperative grid resident (one workgroup per CU)
static inline int wr_cluster(int B, int N, int M) {
    int K = 8;
    while (K > 1 && ((long long)B * K > 256 || N < 4 * K || (N + K - 1) / K * (K - 1) >= N)) K >>= 1;
    (void)M;
    return K;
}

// ---------------------------------------------------------------------------------------------
// Section timers (diagnostic builds only: tools/section_timers.py compiles one translation unit with
// -DPSVO_SECTION_TIMERS into a separate library).  Lane 0 of workgroup (0,0) accumulates the shader-clock
// cycles between consecutive SEC(i) marks of a persistent kernel; the product library has none of this.
// ---------------------------------------------------------------------------------------------
#ifdef PSVO_SECTION_TIMERS
#define PSVO_TIMERS_DEFINE(name)                                                                        \
    __device__ unsigned long long g_sec_##name[32];                                                       \
    extern "C" int psvo_debug_timers_##name(unsigned long long* out, int reset) {                         \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sec_##name), 32 * sizeof(unsigned long long)) != hipSuccess) \
            return -1;                                                                                    \
        if (reset) {                                                                                      \
            unsigned long long z[32] = {0};                                                               \
            if (hipMemcpyToSymbol(HIP_SYMBOL(g_sec_##name), z, sizeof(z)) != hipSuccess) return -1;       \
        }                                                                                                 \
        return 0;                                                                                         \
    }
#define SEC_INIT(name)                                                                          \
    unsigned long long* const _sec = g_sec_##name;                                              \
    const bool _rec = (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0);                 \
    unsigned long long _tp = clock64();
#define SEC(i)                                   \
    do {                                         \
        const unsigned long long _tn = clock64(); \
        if (_rec) _sec[i] += _tn - _tp;          \
        _tp = _tn;                               \
    } while (0)
#else
#define PSVO_TIMERS_DEFINE(name)
#define SEC_INIT(name)
#define SEC(i)
#endif

// v_exp_f32 / v_log_f32 are base-2 on CDNA; keep hot loops in the log2 domain.
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_fast(float x) { return __builtin_amdgcn_logf(x); }

// Packed f32 pairs: gfx950 executes v_pk_{add,mul,fma,max}_f32 on two floats per lane at the rate of one
// scalar VALU op (this is where the 157.3 TFLOP/s f32 vector peak comes from).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_max(f2 a, f2 b) { return __builtin_elementwise_max(a, b); }
// c + {a.x, a.x} * b (HALF = 0) or c + {a.y, a.y} * b (HALF = 1): the splat is the instruction's op_sel, not a register
// pair.  (Written as f2{a.x, a.x} hipcc materialises the pair with v_mov, and hoists all of them out of a loop that
// re-uses a: two registers per value instead of one -- 128 extra VGPRs in the two-layer MLP at H = 64.)
template <int HALF>
__device__ __forceinline__ f2 pk_fma_bcast(f2 a, f2 b, f2 c) {
    f2 d;
    if constexpr (HALF == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Wave-uniform loop constants (inverse scales, proposal coefficients ...) would be kept in SGPRs by hipcc; the persistent
// kernels hold 20-30 array base pointers (two SGPRs each) on top of them, more than the 102 SGPRs a wave has, and every
// SGPR that does not fit is spilled to a VGPR lane and read back with v_readlane_b32 + s_nop INSIDE the time loop
// (bsim_bwd at C*: 109 spilled SGPRs, 125 v_readlane + 121 s_nop per step).  An empty asm with a "+v" constraint makes
// the value opaque in a VGPR: one v_mov before the loop, no scalar register, and VALU instructions that read two such
// constants need no extra v_mov for the constant-bus limit either.
__device__ __forceinline__ void keep_in_vgpr(float& x) { asm volatile("" : "+v"(x)); }
template <int K>
__device__ __forceinline__ void keep_in_vgpr(float (&x)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) asm volatile("" : "+v"(x[i]));
}

// Re-read a POINTER kernel argument from the kernarg segment at its point of use.  The persistent kernels take 20-30
// array pointers; hipcc loads them all into SGPRs in the prologue and keeps them live across the time loop (two SGPRs
// each), which is what overflows the 102 SGPRs of a wave.  Routed through this helper an argument costs one s_load_dwordx2
// (scalar cache, off the VALU) where it is used and no register in between: the opaque offset keeps hipcc from hoisting
// the load back out of the loop.  `byte_off` = offsetof(ArgsStruct, field); the struct is the kernel's only parameter.
template <class T>
__device__ __forceinline__ T* arg_ptr(unsigned byte_off) {
    asm volatile("" : "+s"(byte_off));
    typedef const char __attribute__((address_space(4))) * kptr;
    const kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    typedef T* const __attribute__((address_space(4))) * pptr;
    return *(pptr)(kp + byte_off);
}
// NPTR (2, 4 or 8) pointer arguments that sit next to each other in the argument struct, fetched with ONE scalar load
// (s_load_dwordx4 / x8 / x16).  A section that issues a dozen global loads through PSVO_ARG pays the scalar-cache latency of
// each pointer in turn (s_load, s_waitcnt, global_load, s_load, ...: ~1500 cycles per step of bsim_bwd2 for loads that only
// ISSUE); with the block form the wave waits once.  `byte_off` = offsetof(ArgsStruct, first field), a multiple of 8 NPTR.
template <int NPTR>
__device__ __forceinline__ void arg_block(unsigned byte_off, unsigned long long (&out)[NPTR]) {
    static_assert(NPTR == 2 || NPTR == 4 || NPTR == 8, "pointer block of 2, 4 or 8");
    asm volatile("" : "+s"(byte_off));
    typedef const char __attribute__((address_space(4))) * kptr;
    const kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    typedef unsigned long long vec_t __attribute__((ext_vector_type(NPTR)));
    typedef const vec_t __attribute__((address_space(4))) * vptr;
    const vec_t v = *(vptr)(kp + byte_off);
#pragma unroll
    for (int i = 0; i < NPTR; ++i) out[i] = v[i];
}
#define PSVO_ARG(args_type, field) \
    ::psvo::arg_ptr<std::remove_pointer_t<decltype(args_type::field)>>((unsigned)offsetof(args_type, field))

// ---------------------------------------------------------------------------------------------
// One-hidden-layer MLP with weights staged in LDS, read as wave-uniform float4 broadcasts.
// LDS image: W1[DIN][H] | b1[H] | W2T[DOUT][H] | b2[DOUT padded to 4]
// ---------------------------------------------------------------------------------------------
// mean of the reference's tf_poisson emission (src/distribution/poisson.py:33-38): softplus(MLP_g(x)) + 1e-6 under a
// unit-scale normal; emis_dmean is d mean / d MLP output (softplus threshold 20 as in psvo_sigma_forward)
__device__ __forceinline__ float emis_mean(float r) { return (r > 20.f ? r : log1pf(expf(r))) + 1e-6f; }
__device__ __forceinline__ float emis_dmean(float r) { return 1.f / (1.f + expf(-r)); }

template <int DIN, int H, int DOUT, int L = 1>
struct MlpLds;

template <int DIN, int H, int DOUT>
struct MlpLds<DIN, H, DOUT, 1> {
    static constexpr int kW1 = 0;
    static constexpr int kB1 = DIN * H;
    static constexpr int kW2 = kB1 + H;
    static constexpr int kB2 = kW2 + DOUT * H;
    static constexpr int kSize = kB2 + ((DOUT + 3) & ~3);

    // cooperative load by the whole block; caller syncs afterwards
    __device__ static void load(float* __restrict__ w, const psvo_mlp& p, int tid, int nthreads) {
        for (int i = tid; i < DIN * H; i += nthreads) w[kW1 + i] = p.W1[i];
        for (int i = tid; i < H; i += nthreads) w[kB1 + i] = p.b1[i];
        for (int i = tid; i < DOUT * H; i += nthreads) {
            const int o = i / H, k = i - o * H;
            w[kW2 + i] = p.W2[k * DOUT + o];
        }
        for (int i = tid; i < ((DOUT + 3) & ~3); i += nthreads) w[kB2 + i] = i < DOUT ? p.b2[i] : 0.f;
    }

    // Four hidden units with packed f32 math (units (k, k+1) and (k+2, k+3) share an instruction):
    // h = relu(x W1[:, k:k+4] + b1[k:k+4]); acc[o] += h (*) W2[k:k+4, o] kept as two partial sums per output.
    __device__ __forceinline__ static void group4(const float* __restrict__ w, int k, const float (&x)[DIN],
                                                  f2 (&acc)[DOUT]) {
        const float4 b = *reinterpret_cast<const float4*>(w + kB1 + k);
        f2 ha = f2{b.x, b.y}, hb = f2{b.z, b.w};
#pragma unroll
        for (int i = 0; i < DIN; ++i) {
            const float4 wi = *reinterpret_cast<const float4*>(w + kW1 + i * H + k);
            const f2 xi = f2{x[i], x[i]};
            ha = pk_fma(xi, f2{wi.x, wi.y}, ha);
            hb = pk_fma(xi, f2{wi.z, wi.w}, hb);
        }
        const f2 zero = f2{0.f, 0.f};
        ha = pk_max(ha, zero);
        hb = pk_max(hb, zero);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            const float4 wo = *reinterpret_cast<const float4*>(w + kW2 + o * H + k);
            acc[o] = pk_fma(ha, f2{wo.x, wo.y}, acc[o]);
            acc[o] = pk_fma(hb, f2{wo.z, wo.w}, acc[o]);
        }
    }

    // out = relu(x W1 + b1) W2 + b2.
    // ROLLED = false: fully unrolled (the compiler may keep loop-invariant weights in VGPRs);
    // ROLLED = true : a real loop over groups of 8 hidden units (bounded register pressure).
    template <bool ROLLED = false>
    __device__ __forceinline__ static void eval(const float* __restrict__ w, const float (&x)[DIN],
                                                float (&out)[DOUT]) {
        f2 acc[DOUT];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) acc[o] = f2{w[kB2 + o], 0.f};
        if constexpr (ROLLED && (H % 8 == 0) && (H > 8)) {
#pragma unroll 1
            for (int k = 0; k < H; k += 8) {
                group4(w, k, x, acc);
                group4(w, k + 4, x, acc);
            }
        } else {
#pragma unroll
            for (int k = 0; k < H; k += 4) group4(w, k, x, acc);
        }
#pragma unroll
        for (int o = 0; o < DOUT; ++o) out[o] = acc[o].x + acc[o].y;
    }

    // Partial evaluation over the hidden units [part*H/S, (part+1)*H/S): S lanes share one MLP
    // evaluation and the caller sums `out` over them (xor-shuffles).  b2 is contributed by part 0.
    // (PB: position of the lane bits that tell the S lanes apart -- used by the two-layer form only)
    template <int S, int PB = 0>
    __device__ __forceinline__ static void eval_part(const float* __restrict__ w, int part, const float (&x)[DIN],
                                                     float (&out)[DOUT]) {
        constexpr int HP = H / S;
        static_assert(HP % 4 == 0, "hidden slice must be a multiple of 4");
        const float* wp = w + part * HP;  // shifts k in all three arrays (W1 rows, b1, W2T rows)
        f2 acc[DOUT];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) acc[o] = f2{part == 0 ? w[kB2 + o] : 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < HP; k += 4) group4(wp, k, x, acc);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) out[o] = acc[o].x + acc[o].y;
    }

    // input gradient of four hidden units (recomputes their pre-activations; relu' = [pre > 0])
    __device__ __forceinline__ static void bwd_group4(const float* __restrict__ w, int k, const float (&x)[DIN],
                                                      const float (&dout)[DOUT], f2 (&dxa)[DIN]) {
        const float4 b = *reinterpret_cast<const float4*>(w + kB1 + k);
        f2 pa = f2{b.x, b.y}, pb = f2{b.z, b.w};
        f2 wa[DIN], wb[DIN];
#pragma unroll
        for (int i = 0; i < DIN; ++i) {
            const float4 wi = *reinterpret_cast<const float4*>(w + kW1 + i * H + k);
            wa[i] = f2{wi.x, wi.y};
            wb[i] = f2{wi.z, wi.w};
            const f2 xi = f2{x[i], x[i]};
            pa = pk_fma(xi, wa[i], pa);
            pb = pk_fma(xi, wb[i], pb);
        }
        f2 da = f2{0.f, 0.f}, db = da;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            const float4 wo = *reinterpret_cast<const float4*>(w + kW2 + o * H + k);
            const f2 go = f2{dout[o], dout[o]};
            da = pk_fma(go, f2{wo.x, wo.y}, da);
            db = pk_fma(go, f2{wo.z, wo.w}, db);
        }
        da = f2{pa.x > 0.f ? da.x : 0.f, pa.y > 0.f ? da.y : 0.f};
        db = f2{pb.x > 0.f ? db.x : 0.f, pb.y > 0.f ? db.y : 0.f};
#pragma unroll
        for (int i = 0; i < DIN; ++i) {
            dxa[i] = pk_fma(da, wa[i], dxa[i]);
            dxa[i] = pk_fma(db, wb[i], dxa[i]);
        }
    }

    // dx += partial input gradient over the same hidden slice (caller sums dx over the S lanes)
    template <int S, int PB = 0>
    __device__ __forceinline__ static void bwd_input_part(const float* __restrict__ w, int part, const float (&x)[DIN],
                                                          const float (&dout)[DOUT], float (&dx)[DIN]) {
        constexpr int HP = H / S;
        const float* wp = w + part * HP;
        f2 dxa[DIN];
#pragma unroll
        for (int i = 0; i < DIN; ++i) dxa[i] = f2{dx[i], 0.f};
#pragma unroll
        for (int k = 0; k < HP; k += 4) bwd_group4(wp, k, x, dout, dxa);
#pragma unroll
        for (int i = 0; i < DIN; ++i) dx[i] = dxa[i].x + dxa[i].y;
    }

    // dx += (d out / d x)^T dout : recomputes the hidden pre-activations (nothing is stored)
    template <bool ROLLED = false>
    __device__ __forceinline__ static void bwd_input(const float* __restrict__ w, const float (&x)[DIN],
                                                     const float (&dout)[DOUT], float (&dx)[DIN]) {
        f2 dxa[DIN];
#pragma unroll
        for (int i = 0; i < DIN; ++i) dxa[i] = f2{dx[i], 0.f};
        if constexpr (ROLLED && (H % 8 == 0) && (H > 8)) {
#pragma unroll 1
            for (int k = 0; k < H; k += 8) {
                bwd_group4(w, k, x, dout, dxa);
                bwd_group4(w, k + 4, x, dout, dxa);
            }
        } else {
#pragma unroll
            for (int k = 0; k < H; k += 4) bwd_group4(w, k, x, dout, dxa);
        }
#pragma unroll
        for (int i = 0; i < DIN; ++i) dx[i] = dxa[i].x + dxa[i].y;
    }
};

// ---------------------------------------------------------------------------------------------
// Two hidden layers of width H (reference src/transformation/MLP.py:24-38,50-54 with *_layers = "H,H"):
//     mu = relu(relu(x W1 + b1) Wh + bh) W2 + b2.
// The H x H layer is the one contraction of the per-particle path with a GEMM shape (K = H = 32 / 64), and it runs on the
// matrix pipe INSIDE the persistent kernels (round 3; round 2 walked it per lane on the VALU with Wh broadcast from LDS --
// 4 FMAs per ds_read_b128, LDS-issue-bound: mlp2_valu.h, -DPSVO_L2_VALU).
//
// Same interface as the one-layer struct (eval / eval_part<S> / bwd_input / bwd_input_part<S>), so the kernels do not
// change; what changes is who computes.  The ROWS of a wave -- one per lane, or one per S lanes that hold the same input --
// are worked on in groups of 16 with the lane mapping of v_mfma_f32_16x16x4_f32 itself: lane = (g, r), g = lane >> 4 the
// K slot, r = lane & 15 the row of the group.  Lane (g, r) fetches row r's input (ds_bpermute from the lane that owns the
// row: DIN values), evaluates the H / 4 first-layer units u = 16 q + 4 g + c (q < H / 16, c < 4) of THAT row, and these are
// the B operands of the product as they stand:
//     pre2^T[j][r] = sum_k Wh[k][j] h1[r][k]      D[i][r] += A[i][g] B[g][r],  A = WhT[16 jt + i][16 q + 4 g + c]  (LDS),
//                                                                               B = h1 of row r, unit 16 q + 4 g + c (own register)
// and the accumulator layout hands lane (g, r) the pre-activations of row r for units 16 jt + 4 g + v -- the same unit set
// again, so relu, the narrow output layer (partial sums over the lane's H / 4 units, summed over g with two swap-adds),
// and in the reverse pass  d h1^T[k][r] = sum_j Wh[k][j] d pre2[r][j]  (B = the lane's own d pre2 registers,
// A = WhT[16 jt + 4 g + v][16 kt + i]) chain without any transposition through LDS.  The result travels back to the
// row's owner lane with one ds_bpermute per value.  Nothing is stored per particle; the reverse pass recomputes.
// LDS image: the one-layer image, then WhT[j][k] = Wh[k][j] with rows padded to H + 4 floats (both products then read it
// with the minimum of two lanes per bank), then bh[H].
// Cost per group of 16 rows at H = 64: 64 MFMAs (2048 cycles of the matrix pipe = its f32 peak) + 16 ds_read_b128 forward;
// 128 MFMAs + 16 ds_read_b128 + 64 ds_read_b32 for the input gradient.
// ---------------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));
template <int MASK>
__device__ __forceinline__ float xor_lane(float v);      // (defined with the cross-lane primitives below)
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

__device__ __forceinline__ float lane_fetch(float v, int src_lane) {      // value of `v` in lane src_lane
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

#ifdef PSVO_L2_VALU
#include "mlp2_valu.h"
#else
template <int DIN, int H, int DOUT>
struct MlpLds<DIN, H, DOUT, 2> {
    static constexpr int kW1 = 0;
    static constexpr int kB1 = DIN * H;
    static constexpr int kW2 = kB1 + H;
    static constexpr int kB2 = kW2 + DOUT * H;
    static constexpr int kWS = H + 4;                          // row stride of the transposed hidden kernel
    static constexpr int kWh = kB2 + ((DOUT + 3) & ~3);        // WhT[j][k], j = second-layer unit, k = first-layer unit
    static constexpr int kBh = kWh + H * kWS;
    static constexpr int kSize = kBh + H;
    static constexpr int NQ = H / 16;                          // 16-unit tiles per layer
    static_assert(H % 16 == 0, "hidden width must be a multiple of 16 (MFMA tile)");

    __device__ static void load(float* __restrict__ w, const psvo_mlp& p, int tid, int nthreads) {
        MlpLds<DIN, H, DOUT, 1>::load(w, p, tid, nthreads);
        for (int i = tid; i < H * H; i += nthreads) {
            const int k = i / H, j = i - k * H;                // Wh is keras (in = k, out = j), row-major
            w[kWh + j * kWS + k] = p.Wh[i];
        }
        for (int i = tid; i < H * (kWS - H); i += nthreads) w[kWh + (i / (kWS - H)) * kWS + H + i % (kWS - H)] = 0.f;
        for (int i = tid; i < H; i += nthreads) w[kBh + i] = p.bh[i];
    }

    // rows of a wave: S lanes per row that differ in the lane bits [PB, PB + log2 S); the lane with those bits zero owns it
    template <int S, int PB>
    struct Rows {
        static constexpr int LS = (S == 1) ? 0 : (S == 2) ? 1 : (S == 4) ? 2 : (S == 8) ? 3 : (S == 16) ? 4 : (S == 32) ? 5 : 6;
        static_assert((1 << LS) == S, "lanes per row: a power of two");
        static constexpr int NR = 64 / S;                      // rows per wave
        static constexpr int NG = (NR + 15) / 16;              // groups of 16
        static constexpr int LOW = (1 << PB) - 1;
        __device__ __forceinline__ static int row(int lane) { return ((lane >> (PB + LS)) << PB) | (lane & LOW); }
        __device__ __forceinline__ static int owner(int rho) { return ((rho >> PB) << (PB + LS)) | (rho & LOW); }
        __device__ __forceinline__ static bool owns(int lane) { return ((lane >> PB) & (S - 1)) == 0; }
    };

    // pre-activations of the lane's first-layer units u = 16 q + 4 g + c for the row whose input is xr
    __device__ __forceinline__ static void layer1(const float* __restrict__ w, int g, const float (&xr)[DIN],
                                                  float (&pre)[NQ][4]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 b = *reinterpret_cast<const float4*>(w + kB1 + 16 * q + 4 * g);
            pre[q][0] = b.x; pre[q][1] = b.y; pre[q][2] = b.z; pre[q][3] = b.w;
#pragma unroll
            for (int d = 0; d < DIN; ++d) {
                const float4 wi = *reinterpret_cast<const float4*>(w + kW1 + d * H + 16 * q + 4 * g);
                pre[q][0] = fmaf(xr[d], wi.x, pre[q][0]);
                pre[q][1] = fmaf(xr[d], wi.y, pre[q][1]);
                pre[q][2] = fmaf(xr[d], wi.z, pre[q][2]);
                pre[q][3] = fmaf(xr[d], wi.w, pre[q][3]);
            }
        }
    }

    // D[jt][v] = pre2 of the lane's row, unit 16 jt + 4 g + v (bias included)
    __device__ __forceinline__ static void hidden2(const float* __restrict__ w, int g, int r, const float (&pre1)[NQ][4],
                                                   f4v (&D)[NQ]) {
#pragma unroll
        for (int jt = 0; jt < NQ; ++jt) {
            const float4 b = *reinterpret_cast<const float4*>(w + kBh + 16 * jt + 4 * g);
            D[jt] = f4v{b.x, b.y, b.z, b.w};
        }
        const float* wa = w + kWh + r * kWS + 4 * g;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float h[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) h[c] = fmaxf(pre1[q][c], 0.f);
#pragma unroll
            for (int jt = 0; jt < NQ; ++jt) {
                const float4 a4 = *reinterpret_cast<const float4*>(wa + 16 * jt * kWS + 16 * q);
                D[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, h[0], D[jt], 0, 0, 0);
                D[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, h[1], D[jt], 0, 0, 0);
                D[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, h[2], D[jt], 0, 0, 0);
                D[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, h[3], D[jt], 0, 0, 0);
            }
        }
    }

    template <int S, int PB>
    __device__ __forceinline__ static void eval_rows(const float* __restrict__ w, bool with_bias, const float (&x)[DIN],
                                                     float (&out)[DOUT]) {
        using R = Rows<S, PB>;
        const int lane = __lane_id(), g = lane >> 4, r = lane & 15;
        const int my_row = R::row(lane);
        const bool mine = R::owns(lane);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) out[o] = 0.f;
#pragma unroll 1
        for (int p = 0; p < R::NG; ++p) {
            const int src = R::owner((16 * p + r) & (R::NR - 1));
            float xr[DIN];
#pragma unroll
            for (int d = 0; d < DIN; ++d) xr[d] = lane_fetch(x[d], src);
            float pre1[NQ][4];
            layer1(w, g, xr, pre1);
            f4v D[NQ];
            hidden2(w, g, r, pre1, D);
            float acc[DOUT];
#pragma unroll
            for (int o = 0; o < DOUT; ++o) acc[o] = 0.f;
#pragma unroll
            for (int jt = 0; jt < NQ; ++jt) {
                const float h0 = fmaxf(D[jt][0], 0.f), h1 = fmaxf(D[jt][1], 0.f), h2 = fmaxf(D[jt][2], 0.f),
                            h3 = fmaxf(D[jt][3], 0.f);
#pragma unroll
                for (int o = 0; o < DOUT; ++o) {
                    const float4 wo = *reinterpret_cast<const float4*>(w + kW2 + o * H + 16 * jt + 4 * g);
                    acc[o] = fmaf(h0, wo.x, fmaf(h1, wo.y, fmaf(h2, wo.z, fmaf(h3, wo.w, acc[o]))));
                }
            }
#pragma unroll
            for (int o = 0; o < DOUT; ++o) {
                float v = acc[o];
                v += xor_lane<16>(v);
                v += xor_lane<32>(v);                          // every lane (., r) now holds the row's output
                const float t = lane_fetch(v, my_row & 15);
                if (mine && (my_row >> 4) == p) out[o] = t + (with_bias ? w[kB2 + o] : 0.f);
            }
        }
    }

    // dx += (d out / d x)^T dout for the rows of the wave (owner lanes; the other lanes of a row add nothing)
    template <int S, int PB>
    __device__ __forceinline__ static void bwd_rows(const float* __restrict__ w, const float (&x)[DIN],
                                                    const float (&dout)[DOUT], float (&dx)[DIN]) {
        using R = Rows<S, PB>;
        const int lane = __lane_id(), g = lane >> 4, r = lane & 15;
        const int my_row = R::row(lane);
        const bool mine = R::owns(lane);
#pragma unroll 1
        for (int p = 0; p < R::NG; ++p) {
            const int src = R::owner((16 * p + r) & (R::NR - 1));
            float xr[DIN], gr[DOUT];
#pragma unroll
            for (int d = 0; d < DIN; ++d) xr[d] = lane_fetch(x[d], src);
#pragma unroll
            for (int o = 0; o < DOUT; ++o) gr[o] = lane_fetch(dout[o], src);
            float pre1[NQ][4];
            layer1(w, g, xr, pre1);
            f4v D[NQ];
            hidden2(w, g, r, pre1, D);
            // d pre2 of the lane's units: relu'(pre2) * (W2 dout)
            float dp[NQ][4];
#pragma unroll
            for (int jt = 0; jt < NQ; ++jt) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < DOUT; ++o) {
                    const float4 wo = *reinterpret_cast<const float4*>(w + kW2 + o * H + 16 * jt + 4 * g);
                    s[0] = fmaf(gr[o], wo.x, s[0]); s[1] = fmaf(gr[o], wo.y, s[1]);
                    s[2] = fmaf(gr[o], wo.z, s[2]); s[3] = fmaf(gr[o], wo.w, s[3]);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) dp[jt][v] = D[jt][v] > 0.f ? s[v] : 0.f;
            }
            // d h1^T[k][r] = sum_j Wh[k][j] d pre2[r][j]: A = WhT[16 jt + 4 g + v][16 kt + i], B = dp[jt][v]
            f4v E[NQ];
#pragma unroll
            for (int kt = 0; kt < NQ; ++kt) E[kt] = f4v{0.f, 0.f, 0.f, 0.f};
            const float* wb = w + kWh + 4 * g * kWS + r;
#pragma unroll
            for (int jt = 0; jt < NQ; ++jt) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
#pragma unroll
                    for (int kt = 0; kt < NQ; ++kt) {
                        const float a = wb[(16 * jt + v) * kWS + 16 * kt];
                        E[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dp[jt][v], E[kt], 0, 0, 0);
                    }
                }
            }
            // d pre1 = relu'(pre1) d h1;  d x = W1 d pre1 (partial over the lane's units, summed over g)
            float acc[DIN];
#pragma unroll
            for (int d = 0; d < DIN; ++d) acc[d] = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) e[c] = pre1[q][c] > 0.f ? E[q][c] : 0.f;
#pragma unroll
                for (int d = 0; d < DIN; ++d) {
                    const float4 wi = *reinterpret_cast<const float4*>(w + kW1 + d * H + 16 * q + 4 * g);
                    acc[d] = fmaf(e[0], wi.x, fmaf(e[1], wi.y, fmaf(e[2], wi.z, fmaf(e[3], wi.w, acc[d]))));
                }
            }
#pragma unroll
            for (int d = 0; d < DIN; ++d) {
                float v = acc[d];
                v += xor_lane<16>(v);
                v += xor_lane<32>(v);
                const float t = lane_fetch(v, my_row & 15);
                if (mine && (my_row >> 4) == p) dx[d] += t;
            }
        }
    }

    template <bool ROLLED = false>
    __device__ __forceinline__ static void eval(const float* __restrict__ w, const float (&x)[DIN], float (&out)[DOUT]) {
        eval_rows<1, 0>(w, true, x, out);
    }
    // S lanes (lane bits [PB, PB + log2 S)) hold the same input: the lane with those bits zero receives the whole result,
    // the others zero -- the caller's sum over the S lanes is unchanged
    template <int S, int PB = 0>
    __device__ __forceinline__ static void eval_part(const float* __restrict__ w, int /*part*/, const float (&x)[DIN],
                                                     float (&out)[DOUT]) {
        eval_rows<S, PB>(w, true, x, out);
    }
    template <bool ROLLED = false>
    __device__ __forceinline__ static void bwd_input(const float* __restrict__ w, const float (&x)[DIN],
                                                     const float (&dout)[DOUT], float (&dx)[DIN]) {
        bwd_rows<1, 0>(w, x, dout, dx);
    }
    template <int S, int PB = 0>
    __device__ __forceinline__ static void bwd_input_part(const float* __restrict__ w, int /*part*/, const float (&x)[DIN],
                                                          const float (&dout)[DOUT], float (&dx)[DIN]) {
        bwd_rows<S, PB>(w, x, dout, dx);
    }
};
#endif   // PSVO_L2_VALU

// diagonal-Gaussian log density given inverse scales and the constant -sum(log s) - D/2 log 2pi
template <int D>
__device__ __forceinline__ float diag_lp(const float (&x)[D], const float (&mu)[D], const float (&inv_s)[D],
                                         float cst) {
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float z = (x[d] - mu[d]) * inv_s[d];
        acc = fmaf(z, z, acc);
    }
    return fmaf(-0.5f, acc, cst);
}

// ---------------------------------------------------------------------------------------------
// cross-lane primitives (wave = 64 lanes) on the VALU: DPP row / quad controls and the gfx950
// v_permlane{16,32}_swap instead of ds_bpermute (LDS crossbar, ~100+ cycles of exposed latency per
// dependent step when a SIMD holds one or two waves, which is the regime of every kernel here).
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = false>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                  CTRL, ROW_MASK, BANK_MASK, BOUND));
}

// value of lane (l ^ MASK), MASK a power of two < 64
template <int MASK>
__device__ __forceinline__ float xor_lane(float v) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8 || MASK == 16 || MASK == 32, "xor_lane mask");
    if constexpr (MASK == 1) {
        return dpp_mov<0xB1, 0xF, 0xF, true>(v, v);      // quad_perm [1,0,3,2]
    } else if constexpr (MASK == 2) {
        return dpp_mov<0x4E, 0xF, 0xF, true>(v, v);      // quad_perm [2,3,0,1]
    } else if constexpr (MASK == 4) {
        float t = dpp_mov<0x104, 0xF, 0x5>(v, v);        // banks 0,2 <- lane + 4   (row_shl:4)
        return dpp_mov<0x114, 0xF, 0xA>(t, v);           // banks 1,3 <- lane - 4   (row_shr:4)
    } else if constexpr (MASK == 8) {
        return dpp_mov<0x128, 0xF, 0xF, true>(v, v);     // row_ror:8
    } else if constexpr (MASK == 16) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        const bool odd_row = (__lane_id() >> 4) & 1;
        return __builtin_bit_cast(float, odd_row ? p[0] : p[1]);
    } else {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __builtin_bit_cast(float, (__lane_id() >= 32) ? p[0] : p[1]);
    }
}

// inclusive prefix sum over the wave (classic 7-step DPP scan; lane 63 ends up with the total)
__device__ __forceinline__ float wave_incl_scan(float v, int /*lane*/) {
    v += dpp_mov<0x111, 0xF, 0xF, true>(0.f, v);        // row_shr:1
    v += dpp_mov<0x112, 0xF, 0xF, true>(0.f, v);        // row_shr:2
    v += dpp_mov<0x114, 0xF, 0xF, true>(0.f, v);        // row_shr:4
    v += dpp_mov<0x118, 0xF, 0xF, true>(0.f, v);        // row_shr:8
    v += dpp_mov<0x142, 0xA, 0xF, false>(0.f, v);       // row_bcast:15 -> rows 1, 3
    v += dpp_mov<0x143, 0xC, 0xF, false>(0.f, v);       // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = wave_incl_scan(v, 0);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -__builtin_huge_valf();
    v = fmaxf(v, dpp_mov<0x111>(ninf, v));
    v = fmaxf(v, dpp_mov<0x112>(ninf, v));
    v = fmaxf(v, dpp_mov<0x114>(ninf, v));
    v = fmaxf(v, dpp_mov<0x118>(ninf, v));
    v = fmaxf(v, dpp_mov<0x142, 0xA>(ninf, v));
    v = fmaxf(v, dpp_mov<0x143, 0xC>(ninf, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// the four lanes of each quad broadcast to all of them: out[i] = value of quad lane i
__device__ __forceinline__ void quad_bcast4(float v, float (&out)[4]) {
    out[0] = dpp_mov<0x00>(v, v);
    out[1] = dpp_mov<0x55>(v, v);
    out[2] = dpp_mov<0xAA>(v, v);
    out[3] = dpp_mov<0xFF>(v, v);
}

// all-reduce over aligned groups of G lanes (G a power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G > 1) v += xor_lane<1>(v);
    if constexpr (G > 2) v += xor_lane<2>(v);
    if constexpr (G > 4) v += xor_lane<4>(v);
    if constexpr (G > 8) v += xor_lane<8>(v);
    if constexpr (G > 16) v += xor_lane<16>(v);
    if constexpr (G > 32) v += xor_lane<32>(v);
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
    if constexpr (G > 1) v = fmaxf(v, xor_lane<1>(v));
    if constexpr (G > 2) v = fmaxf(v, xor_lane<2>(v));
    if constexpr (G > 4) v = fmaxf(v, xor_lane<4>(v));
    if constexpr (G > 8) v = fmaxf(v, xor_lane<8>(v));
    if constexpr (G > 16) v = fmaxf(v, xor_lane<16>(v));
    if constexpr (G > 32) v = fmaxf(v, xor_lane<32>(v));
    return v;
}
// inclusive prefix sum inside aligned groups of G lanes; `m` = lane index inside its group
template <int G>
__device__ __forceinline__ float group_incl_scan(float v, int m) {
    if constexpr (G > 1) { const float t = dpp_mov<0x111, 0xF, 0xF, true>(0.f, v); v += (m >= 1) ? t : 0.f; }
    if constexpr (G > 2) { const float t = dpp_mov<0x112, 0xF, 0xF, true>(0.f, v); v += (m >= 2) ? t : 0.f; }
    if constexpr (G > 4) { const float t = dpp_mov<0x114, 0xF, 0xF, true>(0.f, v); v += (m >= 4) ? t : 0.f; }
    if constexpr (G > 8) { const float t = dpp_mov<0x118, 0xF, 0xF, true>(0.f, v); v += (m >= 8) ? t : 0.f; }
    // groups wider than a 16-lane row: add the running total of the previous row(s) (its last lane)
    if constexpr (G > 16) v += dpp_mov<0x142, 0xA, 0xF, false>(0.f, v);   // row_bcast:15 -> rows 1, 3
    if constexpr (G > 32) v += dpp_mov<0x143, 0xC, 0xF, false>(0.f, v);   // row_bcast:31 -> rows 2, 3
    return v;
}

// value of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// block-wide max / sum over `nw` waves; `red` is >= 2*16 floats of LDS scratch.
// Each call uses its own half of `red` selected by `slot` so two back-to-back calls need one
// barrier each.
__device__ __forceinline__ float block_max(float v, float* red, int slot, int wave, int lane, int nw) {
    v = wave_max(v);
    if (nw == 1) return v;
    float* r = red + slot * 16;
    if (lane == 0) r[wave] = v;
    __syncthreads();
    float m = r[0];
    for (int i = 1; i < nw; ++i) m = fmaxf(m, r[i]);
    return m;
}

}  // namespace psvo
