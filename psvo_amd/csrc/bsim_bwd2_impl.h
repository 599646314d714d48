// Reverse-mode pass of the backward simulation, second layout ("j on lanes").  Same mathematics, inputs and outputs as
// bsim_bwd_impl.h (read its header for the derivation); what changes is WHERE the pair work of a step lives in the wave.
//
// v1 (bsim_bwd_kernel) keeps lane = (chain, half, m) throughout.  In its pair loop a quad of lanes shares four
// sub-particles and walks the forward particles j four at a time, so the per-j sums  d F'_j = sum_{chain,m} c u,
// d W^_j = sum c  have their terms spread over the 16 quads of the wave and are brought together by a butterfly after
// every chunk of entries: 332 of the 1705 VALU instructions of a C* step, next to 125 v_readlane + 121 s_nop that restore
// spilled scalar registers (109 SGPRs spilled: ~30 array base pointers, index arithmetic, lane masks).
//
// v2 splits a step into two lane mappings that talk through a few hundred bytes of wave-private LDS:
//   * per-(chain, m) work -- proposal recompute, MLP_f / MLP_g forward + input gradients, chain reductions, row stores --
//     keeps lane = (chain, part, m), two lanes per (chain, m) with half of every MLP's hidden units each;
//   * the pair phase runs with lane = (j16, g): the 16 lanes of a DPP row hold 16 consecutive forward particles of a
//     tile, lane group g = lane >> 4 holds one ITEM = (chain, quad of four m) of the wave's 8 items per round (2 rounds).
//     Per-j sums then accumulate IN REGISTERS over the items of both rounds (f2 accumulators, one pk_fma per term) and
//     are reduced over the four lane groups once per step by two swap-add stages (v_permlane32_swap / v_permlane16_swap:
//     one swap + one add per pair of values, no selects); per-(chain, m) sums  U = sum_j p u,  V = sum_j p u^2  are
//     all-reduced over the 16 lanes of the row with four DPP row rotations.
// Addressing: every array is read / written as base (SGPR pair) + one 32-bit byte offset per array SHAPE kept in a VGPR and
// advanced by a constant per step, and the float loop constants live in VGPRs (common.h keep_in_vgpr): no index arithmetic
// in the loop, no scalar-register spills.  Arrays are therefore limited to 4 GiB each (the dispatcher falls back to v1).
//
// MFMA variant of the per-j sums (template flag JM, measured beside the VALU form):
//   d[F'|W^]_j = sum_k c_kj [x'_k | 1]  is  C^T [X' | 1]  with k = (item, m) -- v_mfma_f32_16x16x4_f32 takes the lane's
//   four c values (m = 4g + r, r = register) as the B operand (k on lane groups and registers, j on the row lanes: the
//   accumulator layout of the tile IS the operand layout) and [X' | 1]^T as the A operand; the contraction over lane
//   groups that the swap-add stages do on the VALU is part of the instruction.
#pragma once
#define PSVO_BSIM_BWD_V2_UNIT 1
#include "bsim_bwd_impl.h"

namespace psvo {
PSVO_TIMERS_DEFINE(bsim_bwd2)

// base + 32-bit byte offset: hipcc selects the saddr form (global_load_dword v, v_off, s[base:base+1])
__device__ __forceinline__ float ldf(const float* base, unsigned off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ int ldi(const int32_t* base, unsigned off) {
    return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ void stf(float* base, unsigned off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) = v;
}

// One reduce-scatter stage over lane bit 5 (bit 4): every lane passes in the value it KEEPS a sum of on the low side
// (`lo`: lanes 0-31 / even rows) and on the high side (`hi`); returned is, on a low-side lane, lo(own) + lo(partner) and
// on a high-side lane hi(own) + hi(partner), partner = lane ^ 32 (lane ^ 16).  v_permlane32_swap_b32 v0, v1 exchanges lanes
// 32-63 of v0 with lanes 0-31 of v1 (v_permlane16_swap: odd rows of v0 with even rows of v1), after which both sides add
// v0 + v1.  (s_nop 1: two wait states between a VALU write of an operand and the swap, which hipcc does not insert
// inside an asm statement.)
__device__ __forceinline__ float swap_add32(float lo, float hi) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
    return lo + hi;
}
__device__ __forceinline__ float swap_add16(float lo, float hi) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
    return lo + hi;
}
// sum over the 16 lanes of a DPP row, result in every lane (four rotations)
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_mov<0x128, 0xF, 0xF, true>(v, v);   // row_ror:8
    v += dpp_mov<0x124, 0xF, 0xF, true>(v, v);   // row_ror:4
    v += dpp_mov<0x122, 0xF, 0xF, true>(v, v);   // row_ror:2
    v += dpp_mov<0x121, 0xF, 0xF, true>(v, v);   // row_ror:1
    return v;
}

// Reduce-scatter of 4 K values over the 16 lanes of a DPP row: on return the four lanes of bank b (lanes 4 b .. 4 b + 3
// of the row) all hold the row sums of values [b K, (b + 1) K) in out[0 .. K).  The stages over lane bits 3 and 2 need no
// selects: a DPP add with a bank mask writes only the banks that keep that value (row_ror:8 reads lane ^ 8; row_shl:4 /
// row_shr:4 read the neighbouring bank), two instructions per pair of values; the last two stages (inside a quad) are
// all-reduces of the K values that are left.  4 K + 2 K + 2 K instructions against 16 K for a plain all-reduce.
// (s_nop 1: a DPP source written by the preceding VALU instruction needs two wait states, which hipcc does not insert
// inside an asm statement.)
__device__ __forceinline__ float rs_bit3(float lo, float hi) {   // banks 0,1: lo + lo(lane ^ 8);  banks 2,3: hi + hi(lane ^ 8)
    float r;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float rs_bit2(float lo, float hi) {   // banks 0,2: lo + lo(lane + 4);  banks 1,3: hi + hi(lane - 4)
    float r;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <int K>
__device__ __forceinline__ void row_reduce_scatter(const float (&v)[4 * K], float (&out)[K]) {
    float h[2 * K];
#pragma unroll
    for (int i = 0; i < 2 * K; ++i) h[i] = rs_bit3(v[i], v[2 * K + i]);
#pragma unroll
    for (int i = 0; i < K; ++i) out[i] = group_sum<4>(rs_bit2(h[i], h[K + i]));
}

typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

// f32 -> three bf16 pieces by truncation (v == hi + mid + lo exactly: 8 + 8 + 8 significand bits), returned as 16-bit
// patterns.  Two such triples (a, b) give a b = ah bh + ah bm + am bh + ah bl + al bh + am bm + am bl + al bm up to the
// al bl term (2^-32 relative): the eight K slots one lane feeds to v_mfma_f32_16x16x32_bf16.
__device__ __forceinline__ void bf16_split3(float v, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r) & 0xffff0000u;
    const float r2 = r - __uint_as_float(mb);
    h = hb >> 16;
    m = mb >> 16;
    l = __float_as_uint(r2) >> 16;
}
// K slots of one component:   A side [ah, ah, am, ah, al, am, am, al]   B side [bh, bm, bh, bl, bh, bm, bl, bm]
__device__ __forceinline__ uint4 bf16x3_a(float v) {
    unsigned h, m, l;
    bf16_split3(v, h, m, l);
    return make_uint4(h | (h << 16), m | (h << 16), l | (m << 16), m | (l << 16));
}
__device__ __forceinline__ uint4 bf16x3_b(float v) {
    unsigned h, m, l;
    bf16_split3(v, h, m, l);
    return make_uint4(h | (m << 16), h | (l << 16), h | (m << 16), l | (m << 16));
}

// JM = 0: everything on the VALU (per-j sums in registers + swap-add).
// JM = 1: the per-j sums on v_mfma_f32_16x16x4_f32 (19 % of each tile used: measured slower, kept for the A/B table).
// JM = 2 (Dx = 2): the PAIR EXPONENTS on v_mfma_f32_16x16x4_f32.  With K = Dx + 2 = 4 the whole exponent is one product
//   S_kj = [2 x'_0, 2 x'_1, 1, -|x'|^2 - lam2]_k . [F'_0, F'_1, W'_j - |F'_j|^2, 1]_j = W'_j - |x'_k - F'_j|^2 - lam2_k
//   whose accumulator layout (column j on the row lanes, rows k = 4 g + register) is exactly the pair mapping, so one MFMA
//   (16 x 16 tile fully used) replaces the ten packed VALU instructions that form the four exponents of an iteration.
//   The differences u = x' - F' are then never formed: U = sum_j p u and V = sum_j p u^2 follow from R1 = sum_j p F',
//   R2 = sum_j p F'^2 (U = x' - R1, V = x'^2 - 2 x' R1 + R2 with sum_j p = 1), and sum c u = sum c x' - F' sum c.
//   The expanded square cancels where the differenced form does not: absolute error ~ 6e-8 (|x'|^2 + |F'|^2) on S, i.e.
//   4e-6 relative on p at sigma_f = 1 and 2e-4 at sigma_f = 0.1; the gradient tests run this variant at both.
// JM = 3 (Dx = 2): the same product on v_mfma_f32_16x16x32_bf16 (half the passes of the f32 instruction): every f32
//   operand is split into three bf16 pieces (bf16_split3) and lane group g -- component g of the K = 4 product -- feeds
//   its eight K slots with the piece products listed there, so the sum carries the f32 product to 2^-32.  The B side
//   (per forward particle) is split once per step when the tile is staged and kept in LDS as the operand image: a slot is
//   [4 components][4 dwords] + F'_0, F'_1 (20 floats); padded entries use W' = -1e30 (finite: -inf has no bf16 split).
template <int DX, int DY, int H, int M, int JM>
__global__ void __launch_bounds__(256, 2) bsim_bwd2_kernel(const BsimBwdArgs a) {
    using MQ = MlpLds<DX, H, DX, 1>;
    using MG = MlpLds<DX, H, DY, 1>;
    using AC = BAcc<DX, DY>;
    constexpr int PS = (JM == 3) ? 20 : (DX == 4 && JM == 0) ? 5 : BTileSlot<DX>::kFloats;   // floats per tile entry
    constexpr int FO = (JM == 3) ? 16 : 0;    // F'_0, F'_1 inside a slot (JM >= 2)
    constexpr int NA = DX + 1;            // per-j accumulators: d F' (DX) and d W^
    constexpr int PART = 2;               // lanes per (chain, m) in the per-(chain, m) phases
    constexpr int G = M * PART;           // lanes per chain
    constexpr int CPW = 64 / G;           // chains per wave
    constexpr int CM = CPW * M;           // (chain, m) slots per wave = 32
    constexpr int NF = DX + 2;            // exchange fields per (chain, m): x' (DX), lam2, d Lambda
    constexpr int UVS = (2 * DX + 3) & ~3;   // floats per (chain, m) of the U / V hand-back, padded to float4s
    // forward-particle tiles (of 16) per chunk whose per-j sums live in registers (the MFMA accumulators take four
    // registers per tile where the VALU form takes NA: half the tiles per chunk keep it inside the 256-VGPR budget)
    // (Dx >= 3: register budget.  Dx = 4: TWO tiles per chunk -- the unrolled pair arithmetic of a tile takes ~20 registers at
    //  Dx = 4; with four tiles in flight the kernel spilled 88 registers around the pair phase, ~45 KB of scratch traffic per
    //  wave and step, and that traffic, not the ALU, set its pace at C5 (44.5 ms); with eight: 357.)
    constexpr int JC = (JM == 0 && DX == 4) ? 2 : (JM || DX >= 3) ? 4 : 8;
    // RO ("rounds outer", Dx = 4, round 3): the two rounds of items of a lane group are walked one after the other over the
    // WHOLE tile instead of inside every chunk, so only one round's U / V sums (8 Dx registers, not 16 Dx) are live; the
    // per-j sums of a chunk are then flushed once per (round, chunk) and round 1 ADDS to round 0's value in the wave's LDS
    // copy (same lane, same address: ordered).  With the other register savings below this is what lets the Dx = 4 kernel run
    // two waves per SIMD without scratch.
    // (Dx = 3 fits both rounds with chunks of four tiles and measured 5 % faster that way: C3 4.76 against 5.02 ms.)
    constexpr bool RO = (DX == 4) && (JM == 0);
    // Dx = 4, VALU form: the tile is kept as F'[NP] (float4) + W'[NP] (float) -- 20 B per forward particle instead of the
    // 32 B of the padded slot -- so that two workgroups fit a CU's LDS at N = 512 (C5)
    constexpr bool kSplitTile = (DX == 4) && (JM == 0);
    static_assert(CM == 32 && (M % 4) == 0, "two lanes per (chain, m): M in {4, 8, 16, 32}");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NTB = 256, nwv = 4;
    const int B = a.B, T = a.T, N = a.N;
    const int NP = ((N + 127) / 128) * 128;   // tile padded (W' = -inf: zero weight) to whole chunks (128 entries, both forms)
    const int nch = NP / (16 * JC);
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    constexpr int cpb = nwv * CPW;
    // per-(chain, m) mapping
    const int cl = lane / G, part = (lane / M) & 1, m = lane % M;
    const bool h0 = (part == 0);
    const int n_raw = blk * cpb + wave * CPW + cl;
    const bool valid = n_raw < N;
    const int n = valid ? n_raw : N - 1;
    const bool srow = valid && h0;                    // the lane that stores this (chain, m)'s rows
    const bool lead = srow && (m == 0);
    // pair mapping
    const int j16 = lane & 15, g = lane >> 4;

    float* wf = smem;
    float* wg = wf + MQ::kSize;
    float* wqi = wg + MG::kSize;
    float* tile = wqi + MQ::kSize;               // [2][NP][PS]
    float* jacc = tile + 2 * NP * PS;            // [nwv][NA][NP] per-wave d F' / d W^ sums of the step
    float* xch = jacc + nwv * NA * NP;           // [nwv][NF][CM]  per-(chain, m) -> pair phase
    float* uvx = xch + nwv * NF * CM;            // [nwv][CM][UVS] pair phase -> per-(chain, m)
    float* red = uvx + nwv * CM * UVS;           // 16
    float* const xw = xch + wave * NF * CM;
    float* const uw = uvx + wave * CM * UVS;
    float* const ja = jacc + wave * NA * NP;

    MQ::load(wf, a.f, tid, NTB);
    MG::load(wg, a.g, tid, NTB);
    MQ::load(wqi, a.q1inv, tid, NTB);

    // ---- constants (VGPRs: see keep_in_vgpr) ------------------------------------------------------------------------
    const float kappa = sqrtf(0.5f * kLog2e);
    float isf[DX], rp[DX], isfk[DX], isg[DY];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        isf[d] = 1.f / a.sig_f[d];
        rp[d] = isf[d] * kappa;
        isfk[d] = isf[d] / kappa;
    }
    float ikap2 = 1.f / (kappa * kappa);
#pragma unroll
    for (int e = 0; e < DY; ++e) isg[e] = 1.f / a.sig_g[e];
    float pc[DX], pic[DX], pi1[DX], pi2[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        pi1[d] = 1.f / a.sig_q1inv[d];
        pi2[d] = 1.f / a.sig_bq2[d];
        pic[d] = pi1[d] + pi2[d];
        pc[d] = 1.f / pic[d];
    }
    float s_init[DX], is_init[DX], i_isig[DX], im[DX], mi[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        s_init[d] = a.sig_init[d];
        is_init[d] = 1.f / s_init[d];
        i_isig[d] = 1.f / a.isig[d];
        im[d] = a.imean[b * DX + d];
        mi[d] = a.minit[b * DX + d];
    }
    if constexpr (DX <= 2) {
        keep_in_vgpr(isf); keep_in_vgpr(rp); keep_in_vgpr(isfk); keep_in_vgpr(isg); keep_in_vgpr(ikap2);
        keep_in_vgpr(pc); keep_in_vgpr(pic); keep_in_vgpr(pi1); keep_in_vgpr(pi2);
    }
    // (s_init, is_init, i_isig, im, mi are read in the first / last step only: scalar registers)
    const float ninf = -__builtin_huge_valf();
    const float aw = valid ? a.dscore[(size_t)b * N + n] : 0.f;  // d loss / d score of this chain

    // ---- byte offsets of step 0, one per array shape, and their per-step strides -----------------------------------
    const unsigned NM = (unsigned)N * M;
    unsigned o_nm = 4u * ((unsigned)b * NM + n * M + m);                      // (T,B,N,M)     om, lam2
    unsigned o_dnm[DX], o_knm[DY], o_dn[DX];
#pragma unroll
    for (int d = 0; d < DX; ++d) {
        o_dnm[d] = 4u * (((unsigned)b * DX + d) * NM + n * M + m);            // (T,B,DX,N,M)  eps_b, xt, dFt
        o_dn[d] = 4u * (((unsigned)b * DX + d) * N + n);                      // (T,B,DX,N)    bwX, mu1, dmu1, dbmu2_rows
    }
#pragma unroll
    for (int k = 0; k < DY; ++k) o_knm[k] = 4u * (((unsigned)b * DY + k) * NM + n * M + m);   // (T,B,DY,N,M) dGt
    unsigned o_n = 4u * ((unsigned)b * N + n);                                // (T,B,N)       sel
    unsigned o_d = 4u * ((unsigned)b * DX), o_k = 4u * ((unsigned)b * DY);    // (T,B,DX) bmu2, (T,B,DY) obs   (uniform)
    const unsigned s_nm = 4u * B * NM, s_dnm = s_nm * DX, s_knm = s_nm * DY, s_dn = 4u * B * DX * N, s_n = 4u * B * N,
                   s_d = 4u * B * DX, s_k = 4u * B * DY;

    // ---- forward-tile staging (image identical to v1 / the forward kernel) ---------------------------------------------
    float st[DX + 1], st_l = 0.f;
    int st_t = 0;
    auto put_slot = [&](float* buf, int j, const float (&raw)[DX + 1], float l) {
        float F[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) F[d] = raw[d] * rp[d];
        const float W = j < N ? (raw[DX] - l) * kLog2e : ninf;
        if constexpr (JM == 3) {     // the four components of JM = 2 as bf16 piece images + F' in f32
            const float Wf = fmaxf(W, -1e30f);
            uint4* q = reinterpret_cast<uint4*>(buf + j * PS);
            q[0] = bf16x3_b(F[0]);
            q[1] = bf16x3_b(F[DX > 1 ? 1 : 0]);
            q[2] = bf16x3_b(Wf - (F[0] * F[0] + F[DX > 1 ? 1 : 0] * F[DX > 1 ? 1 : 0]));
            q[3] = bf16x3_b(1.f);
            *reinterpret_cast<float4*>(buf + j * PS + 16) = make_float4(F[0], F[DX > 1 ? 1 : 0], 0.f, 0.f);
        } else if constexpr (JM == 2) {     // (F'_0, F'_1, W' - |F'|^2, 1): the slot IS the B operand, one component per lane group
            *reinterpret_cast<float4*>(buf + j * PS) = make_float4(F[0], F[1], W - (F[0] * F[0] + F[DX > 1 ? 1 : 0] * F[DX > 1 ? 1 : 0]), 1.f);
        } else if constexpr (DX <= 3) {
            float4 v;
            v.x = F[0];
            v.y = DX > 1 ? F[DX > 1 ? 1 : 0] : 0.f;
            v.z = DX > 2 ? F[DX > 2 ? 2 : 0] : 0.f;
            v.w = W;
            *reinterpret_cast<float4*>(buf + j * PS) = v;
        } else if constexpr (kSplitTile) {
            *reinterpret_cast<float4*>(buf + 4 * j) = make_float4(F[0], F[1], F[2], F[3]);
            buf[4 * NP + j] = W;
        } else {
            *reinterpret_cast<float4*>(buf + j * PS) = make_float4(F[0], F[1], F[2], F[3]);
            *reinterpret_cast<float4*>(buf + j * PS + 4) = make_float4(W, 0.f, 0.f, 0.f);
        }
    };
    // entry j of a staged tile
    auto tile_entry = [&](const float* buf, int j, float (&F)[DX], float& W) {
        if constexpr (kSplitTile) {
            const float4 e = *reinterpret_cast<const float4*>(buf + 4 * j);
            F[0] = e.x; F[DX > 1 ? 1 : 0] = e.y; F[DX > 2 ? 2 : 0] = e.z; F[DX > 3 ? 3 : 0] = e.w;
            W = buf[4 * NP + j];
        } else {
            read_slot<DX>(buf + j * PS, F, W);
        }
    };
    static_assert(offsetof(BsimBwdArgs, logW) - offsetof(BsimBwdArgs, Fm) == 8 &&
                  offsetof(BsimBwdArgs, lse) - offsetof(BsimBwdArgs, Fm) == 16 && offsetof(BsimBwdArgs, Fm) % 32 == 0,
                  "BsimBwdArgs: forward-tile pointer block");
    auto get_slot = [&](int tt, int j, float (&raw)[DX + 1], const float* Fm, const float* logW) {
        const unsigned tb = (unsigned)tt * B + b;
        const unsigned jc = j < N ? j : N - 1;
#pragma unroll
        for (int d = 0; d < DX; ++d) raw[d] = ldf(Fm, 4u * ((tb * DX + d) * N + jc));
        raw[DX] = ldf(logW, 4u * (tb * N + jc));
    };
    // (a second entry per thread travels through registers too when the tile has more than 256: N = 512, C5 -- copied at store
    //  time it cost an exposed round trip to memory per step)
    float st2[DX + 1];
    const bool two_slots = (DX >= 3) && NP > NTB;      // (Dx = 2 is at its register budget: kept as it was)
    auto stage_load = [&](int tt) {
        unsigned long long pt[4];
        arg_block<4>((unsigned)offsetof(BsimBwdArgs, Fm), pt);     // Fm, logW, lse (, sig_f)
        st_t = tt;
        st_l = ldf(reinterpret_cast<const float*>(pt[2]), 4u * ((unsigned)tt * B + b));
        if (tid < NP) get_slot(tt, tid, st, reinterpret_cast<const float*>(pt[0]), reinterpret_cast<const float*>(pt[1]));
        if (two_slots && tid + NTB < NP)
            get_slot(tt, tid + NTB, st2, reinterpret_cast<const float*>(pt[0]), reinterpret_cast<const float*>(pt[1]));
    };
    auto stage_store = [&](float* buf) {
        if (tid < NP) put_slot(buf, tid, st, st_l);
        if (two_slots && tid + NTB < NP) put_slot(buf, tid + NTB, st2, st_l);
        for (int j = tid + (two_slots ? 2 : 1) * NTB; j < NP; j += NTB) {
            float raw[DX + 1];
            get_slot(st_t, j, raw, PSVO_ARG(BsimBwdArgs, Fm), PSVO_ARG(BsimBwdArgs, logW));
            put_slot(buf, j, raw, st_l);
        }
    };
    if (T >= 2) {   // step t reads forward tile t-1; the first tile needed is tile(0) at t = 1
        stage_load(0);
        stage_store(tile);
    }
    __syncthreads();

    // scalar accumulators of the scale gradients: registers for Dx <= 2.  For Dx >= 3 (7 Dx + Dy of them, live across the pair
    // phase, where the kernel is at its 256-VGPR budget) wave-private LDS words updated with ds_add_f32 -- one per
    // (chain, m) for the sums every sub-particle contributes to (sigma_f, the t = 0 prior scale, sigma_g: 32 slots each) and
    // one per chain for the sums only the chain's lead lane holds (the four product-of-Gaussians sums, sigma_init).
    // (Round 2 kept one word per LANE for all of them: 29 KB at Dx = 4, a third of what a workgroup may use if two are to
    //  share a CU at N = 512.)
    constexpr bool kAccLds = (DX >= 3);
    constexpr int kRowsA = 2 * DX + DY, kRowsB = 5 * DX, kSlotsB = 8;      // (at most 8 chains per wave: M = 4)
    float acc[kAccLds ? 1 : AC::kN];
    float* const accA = red + 16 + wave * (kRowsA * 32 + kRowsB * kSlotsB);     // [nwv][kRowsA][32] + [kRowsB][8]
    float* const accB = accA + kRowsA * 32;
    auto acc_is_lead = [](int k) { return k < 4 * DX || (k >= 5 * DX && k < 6 * DX); };
    auto acc_row = [](int k) {      // row inside accA / accB
        return k < 4 * DX ? k : k < 5 * DX ? k - 4 * DX : k < 6 * DX ? 4 * DX + (k - 5 * DX)
               : k < 7 * DX ? DX + (k - 6 * DX) : 2 * DX + (k - 7 * DX);
    };
    const int s32 = cl * M + m;          // (chain, m) slot of the wave
    if constexpr (kAccLds) {
        for (int i = lane; i < kRowsA * 32 + kRowsB * kSlotsB; i += 64) accA[i] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < AC::kN; ++i) acc[i] = 0.f;
    }
    // (every caller is a part-0 lane -- one per (chain, m) -- and the per-chain sums come from the chain's lead lane alone)
    auto acc_add = [&](int k, float v) {
        if constexpr (kAccLds) {
            if (acc_is_lead(k)) atomicAdd(&accB[acc_row(k) * kSlotsB + cl], v);
            else atomicAdd(&accA[acc_row(k) * 32 + s32], v);
        } else {
            acc[k] += v;
        }
    };
    float dX[DX];  // d loss / d bwX_t of this chain (all lanes of the chain hold the same value)
#pragma unroll
    for (int d = 0; d < DX; ++d) dX[d] = 0.f;

    // per-step inputs, requested one step ahead of their use (issue only)
    struct StepIn {
        float eps[DX], bm[DX], xp[DX], mu1[DX], y[DY], om, lam2;
        int sel;
    };
    // obs, eps_b, bwX, sel, lam2_all, om_all, mu1_all (, dscore) are adjacent in BsimBwdArgs: one s_load_dwordx16 for the seven
    // base pointers of a step's inputs instead of seven dependent scalar loads (common.h: arg_block)
    static_assert(offsetof(BsimBwdArgs, dscore) - offsetof(BsimBwdArgs, obs) == 56 && offsetof(BsimBwdArgs, obs) % 64 == 0 &&
                  offsetof(BsimBwdArgs, eps_b) - offsetof(BsimBwdArgs, obs) == 8 &&
                  offsetof(BsimBwdArgs, bwX) - offsetof(BsimBwdArgs, obs) == 16 &&
                  offsetof(BsimBwdArgs, sel) - offsetof(BsimBwdArgs, obs) == 24 &&
                  offsetof(BsimBwdArgs, lam2_all) - offsetof(BsimBwdArgs, obs) == 32 &&
                  offsetof(BsimBwdArgs, om_all) - offsetof(BsimBwdArgs, obs) == 40 &&
                  offsetof(BsimBwdArgs, mu1_all) - offsetof(BsimBwdArgs, obs) == 48, "BsimBwdArgs: input pointer block");
    auto load_step = [&](bool lst, bool fst, StepIn& s) {    // (the offsets already point at the step to load)
        unsigned long long pb[8];
        arg_block<8>((unsigned)offsetof(BsimBwdArgs, obs), pb);
        const float* const p_obs = reinterpret_cast<const float*>(pb[0]);
        const float* const p_eps = reinterpret_cast<const float*>(pb[1]);
        const float* const p_bwX = reinterpret_cast<const float*>(pb[2]);
        const int32_t* const p_sel = reinterpret_cast<const int32_t*>(pb[3]);
        const float* const p_lam2 = reinterpret_cast<const float*>(pb[4]);
        const float* const p_om = reinterpret_cast<const float*>(pb[5]);
        const float* const p_mu1 = reinterpret_cast<const float*>(pb[6]);
        const float* const p_bm = PSVO_ARG(BsimBwdArgs, bmu2);
        s.sel = ldi(p_sel, o_n);
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            s.eps[d] = ldf(p_eps, o_dnm[d]);
            s.bm[d] = ldf(p_bm, o_d + 4u * d);
            s.xp[d] = lst ? 0.f : ldf(p_bwX, o_dn[d] + s_dn);
            s.mu1[d] = lst ? 0.f : ldf(p_mu1, o_dn[d]);
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) s.y[k] = ldf(p_obs, o_k + 4u * k);
        s.om = ldf(p_om, o_nm);
        s.lam2 = fst ? 0.f : ldf(p_lam2, o_nm);
    };
    auto advance = [&]() {
        o_nm += s_nm; o_n += s_n; o_d += s_d; o_k += s_k;
#pragma unroll
        for (int d = 0; d < DX; ++d) { o_dnm[d] += s_dnm; o_dn[d] += s_dn; }
#pragma unroll
        for (int k = 0; k < DY; ++k) o_knm[k] += s_knm;
    };
    StepIn cur_in, nxt_in;
    load_step(T == 1, true, cur_in);

    phase_skew(a.skew);
    SEC_INIT(bsim_bwd2)
    for (int t = 0; t < T; ++t) {
        SEC(0);
        const bool first = (t == 0), last = (t == T - 1);
        const float* cur = tile + ((t + 1) & 1) * NP * PS;  // tile(t-1), valid for t >= 1
        float* nxt = tile + (t & 1) * NP * PS;              // tile(t) for step t+1
        // (Dx = 4: the next tile's entries are requested behind the pair phase, like the next step's inputs -- the registers
        //  they land in would be live across it; the MLP phase and the chain reductions still cover the latency)
        constexpr bool kLateTile = (DX == 4);
        if (!kLateTile && t + 1 < T && t >= 1) stage_load(t);
        // the offsets are advanced to step t+1 for the prefetch; this step's stores subtract the stride again
        advance();
        // Dx <= 2: the next step's inputs are requested here, a whole step ahead.  Dx >= 3: only after the pair phase (the
        // 3 Dx + 4 registers they land in would be live across it, and the kernel is at its 256-VGPR budget there); the MLP
        // phase, the chain reductions and the two barriers that follow still cover most of their latency.
        constexpr bool kLatePrefetch = (DX >= 3);
        if (!kLatePrefetch && t + 1 < T) load_step(t + 2 == T, false, nxt_in);
        SEC(1);   // issue of the prefetch loads

        // ---- recompute the proposal ---------------------------------------------------------------------
        float xp[DX], eps[DX], bm[DX], mu1[DX], mu[DX], x[DX], y[DY];
        const int sel = cur_in.sel;
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            eps[d] = cur_in.eps[d];
            bm[d] = cur_in.bm[d];
            xp[d] = cur_in.xp[d];
            mu1[d] = cur_in.mu1[d];
            if (!last) {
                mu[d] = pc[d] * fmaf(pi1[d], mu1[d], pi2[d] * bm[d]);
                x[d] = fmaf(pc[d], eps[d], mu[d]);
            } else {
                mu[d] = mi[d];
                x[d] = fmaf(s_init[d], eps[d], mu[d]);
            }
        }
#pragma unroll
        for (int k = 0; k < DY; ++k) y[k] = cur_in.y[k];
        const float pi_m = valid ? exp2_fast(cur_in.om * kLog2e) : 0.f;
        const float issel = (m == sel) ? 1.f : 0.f;
        const float dphi = aw * pi_m;                 // = d g_m = d iota_m
        const float dlam = -aw * (issel - pi_m);

        float dxt[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dxt[d] = h0 ? issel * dX[d] : 0.f;  // dxt: this lane's PARTIAL of d x~_m

        SEC(2);   // proposal recompute, coefficients
        if (!first) {
            // ---- hand (x', lam2, d Lambda) of the wave's 32 (chain, m) to the pair mapping ---------------------------
            if (h0) {
#pragma unroll
                for (int d = 0; d < DX; ++d) xw[d * CM + cl * M + m] = x[d] * rp[d];
                xw[DX * CM + cl * M + m] = cur_in.lam2;
                xw[(DX + 1) * CM + cl * M + m] = dlam;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ---- pair phase: lane = (j16, g); item of round r: it = 4 r + g = (chain, quad) ---------------------------
            if constexpr (RO) {
#pragma unroll 1
                for (int r = 0; r < 2; ++r) {
                    const int it4 = 4 * (4 * r + g);          // first (chain, m) slot of the item
                    f2 xa[DX], xb[DX], Ua[DX], Ub[DX], Va[DX], Vb[DX];
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        const float4 v = *reinterpret_cast<const float4*>(xw + d * CM + it4);
                        xa[d] = f2{v.x, v.y};
                        xb[d] = f2{v.z, v.w};
                        Ua[d] = Ub[d] = Va[d] = Vb[d] = f2{0.f, 0.f};
                    }
                    const float4 lq = *reinterpret_cast<const float4*>(xw + DX * CM + it4);
                    const float4 dl = *reinterpret_cast<const float4*>(xw + (DX + 1) * CM + it4);
                    const f2 lqa = f2{lq.x, lq.y}, lqb = f2{lq.z, lq.w};
                    const f2 dla = f2{dl.x, dl.y}, dlb = f2{dl.z, dl.w};
#pragma unroll 1
                    for (int c = 0; c < nch; ++c) {
                        float A2[JC][NA];
#pragma unroll
                        for (int jt = 0; jt < JC; ++jt) {
                            float F[DX], W;
                            tile_entry(cur, (c * JC + jt) * 16 + j16, F, W);
                            f2 ua[DX], ub[DX], la = f2{W, W}, lb = la;
#pragma unroll
                            for (int d = 0; d < DX; ++d) {
                                const f2 Fd = f2{F[d], F[d]};
                                ua[d] = xa[d] - Fd;
                                ub[d] = xb[d] - Fd;
                                la = pk_fma(-ua[d], ua[d], la);
                                lb = pk_fma(-ub[d], ub[d], lb);
                            }
                            la -= lqa;
                            lb -= lqb;
                            const f2 pa = f2{exp2_fast(la.x), exp2_fast(la.y)}, pb = f2{exp2_fast(lb.x), exp2_fast(lb.y)};
                            const f2 ca = dla * pa, cb = dlb * pb;
#pragma unroll
                            for (int d = 0; d < DX; ++d) {
                                const f2 pua = pa * ua[d], pub = pb * ub[d];
                                Ua[d] += pua;
                                Ub[d] += pub;
                                Va[d] = pk_fma(pua, ua[d], Va[d]);
                                Vb[d] = pk_fma(pub, ub[d], Vb[d]);
                            }
                            const f2 cs = ca + cb;
                            A2[jt][DX] = cs.x + cs.y;
#pragma unroll
                            for (int d = 0; d < DX; ++d) {
                                const f2 ad = pk_fma(cb, ub[d], ca * ua[d]);
                                A2[jt][d] = ad.x + ad.y;
                            }
                        }
                        // flush: reduce over the four lane groups, one owner lane per (j, e); round 1 adds to round 0's value
                        static_assert(JC == 2, "rounds-outer flush: two tiles per chunk");
                        // lane bit 5 keeps tile 0 (groups 0, 1) or tile 1 (groups 2, 3); lane bit 4 is then summed out and
                        // the even group of each pair owns the tile
#pragma unroll
                        for (int e = 0; e < NA; ++e) {
                            float v = swap_add32(A2[0][e], A2[1][e]);
                            v += xor_lane<16>(v);
                            if ((g & 1) == 0) {
                                float* dst = ja + e * NP + (c * JC + (g >> 1)) * 16 + j16;
                                *dst = (r == 0) ? v : *dst + v;
                            }
                        }
                    }
                    // U, V of the round's item: sum over the 16 forward particles of the row, hand back per (chain, m)
                    float vv[4 * 2 * DX], o[UVS];
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        vv[0 * 2 * DX + d] = Ua[d].x; vv[1 * 2 * DX + d] = Ua[d].y;
                        vv[2 * 2 * DX + d] = Ub[d].x; vv[3 * 2 * DX + d] = Ub[d].y;
                        vv[0 * 2 * DX + DX + d] = Va[d].x; vv[1 * 2 * DX + DX + d] = Va[d].y;
                        vv[2 * 2 * DX + DX + d] = Vb[d].x; vv[3 * 2 * DX + DX + d] = Vb[d].y;
                    }
                    float red2[2 * DX];
                    row_reduce_scatter<2 * DX>(vv, red2);
#pragma unroll
                    for (int e = 0; e < UVS; ++e) o[e] = e < 2 * DX ? red2[e < 2 * DX ? e : 0] : 0.f;
                    if ((j16 & 3) == 0) {        // one lane per bank: sub-particle i = j16 >> 2 of item 4 r + g
                        float* dst = uw + (4 * (4 * r + g) + (j16 >> 2)) * UVS;
#pragma unroll
                        for (int e = 0; e < UVS; e += 4)
                            *reinterpret_cast<float4*>(dst + e) = make_float4(o[e], o[e + 1], o[e + 2], o[e + 3]);
                    }
                }
            } else {
            f2 Ua[2][DX], Ub[2][DX], Va[2][DX], Vb[2][DX];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int d = 0; d < DX; ++d) Ua[r][d] = Ub[r][d] = Va[r][d] = Vb[r][d] = f2{0.f, 0.f};

            for (int c = 0; c < nch; ++c) {
                float A2[JC][NA];       // JM = 0: per-j sums over the lane's items
                f4v D4[JC];             // JM = 1: MFMA accumulators (rows = [d F'_0.., d W^], columns = the 16 j of a tile)
#pragma unroll
                for (int jt = 0; jt < JC; ++jt) {
#pragma unroll
                    for (int e = 0; e < NA; ++e) A2[jt][e] = 0.f;
                    D4[jt] = f4v{0.f, 0.f, 0.f, 0.f};
                }
                const float* cbase = cur + (c * 16 * JC + j16) * PS;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int it4 = 4 * (4 * r + g);          // first (chain, m) slot of the item
                    f2 xa[DX], xb[DX];
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        const float4 v = *reinterpret_cast<const float4*>(xw + d * CM + it4);
                        xa[d] = f2{v.x, v.y};
                        xb[d] = f2{v.z, v.w};
                    }
                    const float4 lq = *reinterpret_cast<const float4*>(xw + DX * CM + it4);
                    const float4 dl = *reinterpret_cast<const float4*>(xw + (DX + 1) * CM + it4);
                    const f2 lqa = f2{lq.x, lq.y}, lqb = f2{lq.z, lq.w};
                    const f2 dla = f2{dl.x, dl.y}, dlb = f2{dl.z, dl.w};
                    // JM = 1: A operand of the per-j MFMA, [X' | 1]^T: lane (i = j16, kk = g) holds row i of column
                    // k = 4 g + rr for MFMA rr: i < DX: x'_i of sub-particle rr, i == DX: 1, else 0
                    float xop[4];
                    if constexpr (JM == 1) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            float v = (j16 == DX) ? 1.f : 0.f;
#pragma unroll
                            for (int d = 0; d < DX; ++d) {
                                const float xv = (rr == 0) ? xa[d].x : (rr == 1) ? xa[d].y : (rr == 2) ? xb[d].x : xb[d].y;
                                v = (j16 == d) ? xv : v;
                            }
                            xop[rr] = v;
                        }
                    }
                    // JM = 2: A operand of the exponent MFMA: lane (row = j16 -> slot 16 r + j16 of the round, component g)
                    float sop = 0.f;
                    uint4 aop = make_uint4(0u, 0u, 0u, 0u);      // JM = 3: the bf16 piece image of sop
                    if constexpr (JM >= 2) {
                        const int sl = 16 * r + j16;
                        const float x0 = xw[sl], x1 = xw[CM + sl], l2 = xw[DX * CM + sl];
                        sop = (g == 0) ? 2.f * x0 : (g == 1) ? 2.f * x1 : (g == 2) ? 1.f : -fmaf(x0, x0, fmaf(x1, x1, l2));
                        if constexpr (JM == 3) aop = bf16x3_a(sop);
                    }
                    if constexpr (JM >= 2) {
                        auto exponents = [&](const float* sp) -> f4v {
                            if constexpr (JM == 2) {
                                return __builtin_amdgcn_mfma_f32_16x16x4f32(sop, sp[g], f4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            } else {
                                const uint4 bq = *reinterpret_cast<const uint4*>(sp + 4 * g);
                                return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, aop),
                                                                               __builtin_bit_cast(bf8v, bq),
                                                                               f4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            }
                        };
                        // the MFMA of tile jt + 1 is issued before tile jt's exponentials are consumed (its passes then
                        // run beside this tile's VALU work instead of stalling the wave in front of the first v_exp)
                        f4v Sn = exponents(cbase);
#pragma unroll
                        for (int jt = 0; jt < JC; ++jt) {
                            const float* sp = cbase + jt * 16 * PS;
                            const float2 F2 = *reinterpret_cast<const float2*>(sp + FO);      // F'_0, F'_1 of slot j
                            const f4v S = Sn;
                            if (jt + 1 < JC) Sn = exponents(sp + 16 * PS);
                            const f2 pa = f2{exp2_fast(S[0]), exp2_fast(S[1])}, pb = f2{exp2_fast(S[2]), exp2_fast(S[3])};
                            const f2 ca = dla * pa, cb = dlb * pb;
                            const f2 Fv = f2{F2.x, F2.y}, Fq = Fv * Fv;
                            // (U, V accumulators hold R1 = sum p F' and R2 = sum p F'^2 in this variant)
                            Ua[r][0] = pk_fma(pa, f2{Fv.x, Fv.x}, Ua[r][0]);
                            Ub[r][0] = pk_fma(pb, f2{Fv.x, Fv.x}, Ub[r][0]);
                            Ua[r][DX > 1 ? 1 : 0] = pk_fma(pa, f2{Fv.y, Fv.y}, Ua[r][DX > 1 ? 1 : 0]);
                            Ub[r][DX > 1 ? 1 : 0] = pk_fma(pb, f2{Fv.y, Fv.y}, Ub[r][DX > 1 ? 1 : 0]);
                            Va[r][0] = pk_fma(pa, f2{Fq.x, Fq.x}, Va[r][0]);
                            Vb[r][0] = pk_fma(pb, f2{Fq.x, Fq.x}, Vb[r][0]);
                            Va[r][DX > 1 ? 1 : 0] = pk_fma(pa, f2{Fq.y, Fq.y}, Va[r][DX > 1 ? 1 : 0]);
                            Vb[r][DX > 1 ? 1 : 0] = pk_fma(pb, f2{Fq.y, Fq.y}, Vb[r][DX > 1 ? 1 : 0]);
                            const f2 cs = ca + cb;
                            A2[jt][DX] += cs.x + cs.y;
#pragma unroll
                            for (int d = 0; d < DX; ++d) {
                                const f2 ad = pk_fma(cb, xb[d], ca * xa[d]);     // sum c x' (the F' sum c term: at the flush)
                                A2[jt][d] += ad.x + ad.y;
                            }
                        }
                    } else
#pragma unroll
                    for (int jt = 0; jt < JC; ++jt) {
                        float F[DX], W;
                        if constexpr (kSplitTile) tile_entry(cur, (c * JC + jt) * 16 + j16, F, W);
                        else read_slot<DX>(cbase + jt * 16 * PS, F, W);
                        f2 ua[DX], ub[DX], la = f2{W, W}, lb = la;
#pragma unroll
                        for (int d = 0; d < DX; ++d) {
                            const f2 Fd = f2{F[d], F[d]};
                            ua[d] = xa[d] - Fd;
                            ub[d] = xb[d] - Fd;
                            la = pk_fma(-ua[d], ua[d], la);
                            lb = pk_fma(-ub[d], ub[d], lb);
                        }
                        la -= lqa;
                        lb -= lqb;
                        const f2 pa = f2{exp2_fast(la.x), exp2_fast(la.y)}, pb = f2{exp2_fast(lb.x), exp2_fast(lb.y)};
                        const f2 ca = dla * pa, cb = dlb * pb;
#pragma unroll
                        for (int d = 0; d < DX; ++d) {
                            const f2 pua = pa * ua[d], pub = pb * ub[d];
                            Ua[r][d] += pua;
                            Ub[r][d] += pub;
                            Va[r][d] = pk_fma(pua, ua[d], Va[r][d]);
                            Vb[r][d] = pk_fma(pub, ub[d], Vb[r][d]);
                        }
                        if constexpr (JM != 1) {
                            const f2 cs = ca + cb;
                            A2[jt][DX] += cs.x + cs.y;
#pragma unroll
                            for (int d = 0; d < DX; ++d) {
                                const f2 ad = pk_fma(cb, ub[d], ca * ua[d]);
                                A2[jt][d] += ad.x + ad.y;
                            }
                        } else {
                            // d F'_j wants sum c (x' - F'_j) = (sum c x') - F'_j (sum c): the MFMA forms sum c [x' | 1];
                            // the F'_j term is applied at the flush
                            D4[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xop[0], ca.x, D4[jt], 0, 0, 0);
                            D4[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xop[1], ca.y, D4[jt], 0, 0, 0);
                            D4[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xop[2], cb.x, D4[jt], 0, 0, 0);
                            D4[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xop[3], cb.y, D4[jt], 0, 0, 0);
                        }
                    }
                }
                // ---- flush the chunk's per-j sums: reduce over the four lane groups, one owner lane per (j, e) -----------
                if constexpr (JM != 1) {
                    float v[JC][NA];
#pragma unroll
                    for (int jt = 0; jt < JC; ++jt)
#pragma unroll
                        for (int e = 0; e < NA; ++e) v[jt][e] = A2[jt][e];
#pragma unroll
                    for (int jt = 0; jt < JC / 2; ++jt)
#pragma unroll
                        for (int e = 0; e < NA; ++e) v[jt][e] = swap_add32(v[jt][e], v[jt + JC / 2][e]);
#pragma unroll
                    for (int jt = 0; jt < JC / 4; ++jt)
#pragma unroll
                        for (int e = 0; e < NA; ++e) v[jt][e] = swap_add16(v[jt][e], v[jt + JC / 4][e]);
                    // lane (g, j16) now owns tiles (JC/2)(g >> 1) + (JC/4)(g & 1) + {0 .. JC/4 - 1}
                    const int jt0 = (JC / 2) * (g >> 1) + (JC / 4) * (g & 1);
#pragma unroll
                    for (int jt = 0; jt < JC / 4; ++jt) {
                        if constexpr (JM >= 2) {     // sum c u = sum c x' - F'_j sum c
                            const float2 F2 = *reinterpret_cast<const float2*>(cbase + (jt0 + jt) * 16 * PS + FO);
                            v[jt][0] = fmaf(-F2.x, v[jt][DX], v[jt][0]);
                            v[jt][DX > 1 ? 1 : 0] = fmaf(-F2.y, v[jt][DX], v[jt][DX > 1 ? 1 : 0]);
                        }
#pragma unroll
                        for (int e = 0; e < NA; ++e) ja[e * NP + (c * JC + jt0 + jt) * 16 + j16] = v[jt][e];
                    }
                } else {
                    // accumulator layout: lane (column j16, rows 4 g + reg); rows 0 .. DX are [sum c x'_d .., sum c]
                    if (g == 0 || DX > 3) {
#pragma unroll
                        for (int jt = 0; jt < JC; ++jt) {
                            float F[DX], W;
                            read_slot<DX>(cbase + jt * 16 * PS, F, W);
                            (void)W;
                            const int j = (c * JC + jt) * 16 + j16;
                            if constexpr (DX <= 3) {
                                const float sc = D4[jt][DX];
#pragma unroll
                                for (int d = 0; d < DX; ++d) ja[d * NP + j] = D4[jt][d] - F[d] * sc;
                                ja[DX * NP + j] = sc;
                            } else {
                                // DX = 4: sum c is row 4 = register 0 of lane group 1
                                const float sc = __shfl(D4[jt][0], 16 + j16);
                                if (g == 0) {
#pragma unroll
                                    for (int d = 0; d < DX; ++d) ja[d * NP + j] = D4[jt][d] - F[d] * sc;
                                    ja[DX * NP + j] = sc;
                                }
                            }
                        }
                    }
                }
            }
            // ---- U, V of every item: sum over the 16 forward particles of the row, hand back per (chain, m) -------------
            // values ordered [sub-particle i][U (DX), V (DX)]: bank i of the row ends up with sub-particle i's sums
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float vv[4 * 2 * DX], o[UVS];
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    vv[0 * 2 * DX + d] = Ua[r][d].x; vv[1 * 2 * DX + d] = Ua[r][d].y;
                    vv[2 * 2 * DX + d] = Ub[r][d].x; vv[3 * 2 * DX + d] = Ub[r][d].y;
                    vv[0 * 2 * DX + DX + d] = Va[r][d].x; vv[1 * 2 * DX + DX + d] = Va[r][d].y;
                    vv[2 * 2 * DX + DX + d] = Vb[r][d].x; vv[3 * 2 * DX + DX + d] = Vb[r][d].y;
                }
                float red2[2 * DX];
                row_reduce_scatter<2 * DX>(vv, red2);
#pragma unroll
                for (int e = 0; e < UVS; ++e) o[e] = e < 2 * DX ? red2[e < 2 * DX ? e : 0] : 0.f;
                if ((j16 & 3) == 0) {        // one lane per bank: sub-particle i = j16 >> 2 of item 4 r + g
                    float* dst = uw + (4 * (4 * r + g) + (j16 >> 2)) * UVS;
#pragma unroll
                    for (int e = 0; e < UVS; e += 4)
                        *reinterpret_cast<float4*>(dst + e) = make_float4(o[e], o[e + 1], o[e + 2], o[e + 3]);
                }
            }
            }   // (!RO)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            SEC(3);   // pair phase
            if (h0) {
                float uv[UVS];
#pragma unroll
                for (int e = 0; e < UVS; e += 4) {
                    const float4 q4 = *reinterpret_cast<const float4*>(uw + (cl * M + m) * UVS + e);
                    uv[e] = q4.x; uv[e + 1] = q4.y; uv[e + 2] = q4.z; uv[e + 3] = q4.w;
                }
                // (x~-F)/sigma^2 = u / (sigma kappa);  z^2 = u^2 / kappa^2
                if constexpr (JM >= 2) {     // R1, R2 -> U = x' - R1, V = x'^2 - 2 x' R1 + R2   (sum_j p = 1)
#pragma unroll
                    for (int d = 0; d < DX; ++d) {
                        const float xs = x[d] * rp[d], r1 = uv[d], r2 = uv[DX + d];
                        uv[d] = xs - r1;
                        uv[DX + d] = fmaf(xs, xs - 2.f * r1, r2);
                    }
                }
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    dxt[d] -= dlam * uv[d] * isfk[d];
                    acc_add(AC::kSf + d, dlam * (uv[DX + d] * ikap2 - 1.f) * isf[d]);
                }
            }
        } else {
            // t = 0: iota_m = LN(x~; imean, isig)   (reference PSVO.py:169-175)
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const float z = (x[d] - im[d]) * i_isig[d];
                const float tz = dphi * z * i_isig[d];
                if (h0) {
                    dxt[d] -= tz;
                    acc_add(AC::kSiota + d, dphi * (z * z - 1.f) * i_isig[d]);
                }
            }
        }

        if (kLatePrefetch && t + 1 < T) load_step(t + 2 == T, false, nxt_in);
        if (kLateTile && t + 1 < T && t >= 1) stage_load(t);
        SEC(4);   // hand-back
        // ---- f(x_{t+1} | x~), g(y_t | x~) ------------------------------------------------------------------------
        float dxp_part[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) dxp_part[d] = 0.f;
        {
            float dFo[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) dFo[d] = 0.f;
            if (!last) {
                float fmx[DX];
                MQ::template eval_part<PART>(wf, part, x, fmx);
#pragma unroll
                for (int d = 0; d < DX; ++d) fmx[d] += xor_lane<M>(fmx[d]);
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    const float z = (xp[d] - fmx[d]) * isf[d];
                    dFo[d] = dphi * z * isf[d];
                    if (h0) {
                        dxp_part[d] = -dFo[d];
                        acc_add(AC::kSf + d, dphi * (z * z - 1.f) * isf[d]);
                    }
                }
                MQ::template bwd_input_part<PART>(wf, part, x, dFo, dxt);
            }
            static_assert(offsetof(BsimBwdArgs, dFt) - offsetof(BsimBwdArgs, xt) == 8 &&
                          offsetof(BsimBwdArgs, dGt) - offsetof(BsimBwdArgs, xt) == 16 &&
                          offsetof(BsimBwdArgs, dmu1) - offsetof(BsimBwdArgs, xt) == 24 && offsetof(BsimBwdArgs, xt) % 32 == 0,
                          "BsimBwdArgs: row pointer block");
            unsigned long long pr[4];
            arg_block<4>((unsigned)offsetof(BsimBwdArgs, xt), pr);       // xt, dFt, dGt, dmu1
            if (srow) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    stf(reinterpret_cast<float*>(pr[1]), o_dnm[d] - s_dnm, dFo[d]);
                    stf(reinterpret_cast<float*>(pr[0]), o_dnm[d] - s_dnm, x[d]);
                }
            }
            float gm[DY], dGo[DY];
            MG::template eval_part<PART>(wg, part, x, gm);
#pragma unroll
            for (int k = 0; k < DY; ++k) gm[k] += xor_lane<M>(gm[k]);
#pragma unroll
            for (int k = 0; k < DY; ++k) {
                float dmean = 1.f;
                if (a.emission) { dmean = emis_dmean(gm[k]); gm[k] = emis_mean(gm[k]); }
                const float z = (y[k] - gm[k]) * isg[k];
                dGo[k] = dphi * z * isg[k] * dmean;
                if (h0) acc_add(AC::kSg + k, dphi * (z * z - 1.f) * isg[k]);
                if (srow) stf(reinterpret_cast<float*>(pr[2]), o_knm[k] - s_knm, dGo[k]);
            }
            MG::template bwd_input_part<PART>(wg, part, x, dGo, dxt);
        }

        SEC(5);   // MLP_f / MLP_g forward + input gradients, row stores
        // ---- reduce over the chain's M sub-particles and the two parts ---------------------------------------------
        float dmu[DX], sce[DX], dxp[DX], dim[DX];
#pragma unroll
        for (int d = 0; d < DX; ++d) {
            const float v3 = (first && h0) ? dphi * (x[d] - im[d]) * i_isig[d] * i_isig[d] : 0.f;
            dmu[d] = group_sum<G>(dxt[d]);
            sce[d] = group_sum<G>(dxt[d] * eps[d]);
            dxp[d] = group_sum<G>(dxp_part[d]);
            dim[d] = first ? group_sum<G>(v3) : 0.f;
        }
        float outv[DX];  // per-chain value that is summed over the chains afterwards: d bmu2 / d minit
        if (!last) {
            float dmu1[DX];
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                dmu1[d] = dmu[d] * pc[d] * pi1[d];
                outv[d] = dmu[d] * pc[d] * pi2[d];
                if (lead) {
                    stf(PSVO_ARG(BsimBwdArgs, dmu1), o_dn[d] - s_dn, dmu1[d]);
                    acc_add(AC::kSc + d, sce[d] + aw * pic[d]);   // -sum_m dq_m / c = a / c
                    acc_add(AC::kSmm1 + d, dmu[d] * mu1[d]);
                    acc_add(AC::kSmb + d, dmu[d] * bm[d]);
                    acc_add(AC::kSmm + d, dmu[d] * mu[d]);
                }
            }
            // MLP_q1inv's input is the same in all G lanes of the chain: spread its hidden units over kQS of them
            {
                constexpr int kQS = (H / 4 < G) ? H / 4 : G;
                float dq[DX];
#pragma unroll
                for (int d = 0; d < DX; ++d) dq[d] = 0.f;
                MQ::template bwd_input_part<kQS>(wqi, (lane % G) & (kQS - 1), xp, dmu1, dq);
#pragma unroll
                for (int d = 0; d < DX; ++d) dxp[d] += group_sum<kQS>(dq[d]);
            }
        } else {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                outv[d] = dmu[d];
                if (lead) {
                    stf(PSVO_ARG(BsimBwdArgs, dmu1), o_dn[d] - s_dn, 0.f);
                    acc_add(AC::kSinit + d, sce[d] + aw * is_init[d]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < DX; ++d) dX[d] = dxp[d];
        if (lead) {
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                stf(PSVO_ARG(BsimBwdArgs, dbmu2_rows), o_dn[d] - s_dn, last ? 0.f : outv[d]);
                if (last) PSVO_ARG(BsimBwdArgs, dminit_rows)[((size_t)b * DX + d) * N + n] = outv[d];
                if (first) PSVO_ARG(BsimBwdArgs, dimean_rows)[((size_t)b * DX + d) * N + n] = dim[d];
            }
        }
        SEC(6);   // chain reductions, MLP_q1inv input gradient
        __syncthreads();
        // ---- this workgroup's partial of d Fm[t-1] / d logW[t-1]: fold the four waves (fixed order) ------------------------
        const size_t tb = (size_t)t * B + b;
        if (!first) {
            const size_t tbm = tb - B;
            float* const pF = PSVO_ARG(BsimBwdArgs, dFm_part) + (tbm * nblk + blk) * DX * N;
            float* const pW = PSVO_ARG(BsimBwdArgs, dlogW_part) + (tbm * nblk + blk) * N;
            for (int j = tid; j < N; j += NTB) {
#pragma unroll
                for (int d = 0; d < NA; ++d) {
                    const float* col = jacc + d * NP + j;
                    const float a0 = col[0], a1 = col[NA * NP], a2 = col[2 * NA * NP], a3 = col[3 * NA * NP];
                    const float s = (a0 + a1) + (a2 + a3);
                    if (d < DX) pF[d * N + j] = s * isfk[d < DX ? d : 0];
                    else pW[j] = s;
                }
            }
        }
        if (last) {
            float* const pF = PSVO_ARG(BsimBwdArgs, dFm_part) + (tb * nblk + blk) * DX * N;
            float* const pW = PSVO_ARG(BsimBwdArgs, dlogW_part) + (tb * nblk + blk) * N;
            for (int j = tid; j < N; j += NTB) {
#pragma unroll
                for (int d = 0; d < DX; ++d) pF[d * N + j] = 0.f;
                pW[j] = 0.f;
            }
        }
        SEC(7);   // barrier, workgroup sums, d Fm / d logW flush
        if (t + 1 < T && t >= 1) stage_store(nxt);
        cur_in = nxt_in;
        __syncthreads();
        SEC(8);   // tile store + barrier
    }

    // ---- scalar accumulators: reduce over the workgroup ---------------------------------------------------------------
    for (int i = 0; i < AC::kN; ++i) {
        float v = 0.f;
        if constexpr (kAccLds) {
            if (acc_is_lead(i)) v = lane < kSlotsB ? accB[acc_row(i) * kSlotsB + lane] : 0.f;
            else v = lane < 32 ? accA[acc_row(i) * 32 + lane] : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < AC::kN; ++k) v = (k == i) ? acc[k] : v;
        }
        v = wave_sum(v);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < nwv; ++w) s += red[w];
            a.sacc_part[((size_t)b * nblk + blk) * AC::kN + i] = s;
        }
        __syncthreads();
    }
}

// v2 geometry: 256-thread workgroups, two lanes per (chain, m)
static inline void bsim2_geometry(int N, int M, int& cpb, int& nblk) {
    cpb = 4 * (64 / (2 * M));
    nblk = (N + cpb - 1) / cpb;
}

// v2 covers M in {4, 8, 16, 32} and arrays below 4 GiB (32-bit byte offsets)
static inline bool bsim2_supported(int B, int T, int N, int M, int Dx, int Dy) {
    if (!(M == 4 || M == 8 || M == 16 || M == 32)) return false;
    const long long big = (long long)T * B * (Dx > Dy ? Dx : Dy) * N * M * 4;
    return big < (1ll << 32);
}

template <int DX, int DY, int H, int M>
static int launch_bsim_bwd2(const BsimBwdArgs& a, const BsimBwdOut& o, int jm, hipStream_t stream) {
    using MQ = MlpLds<DX, H, DX, 1>;
    using MG = MlpLds<DX, H, DY, 1>;
    const int PS = (jm == 3 && DX == 2) ? 20 : (DX == 4) ? 5 : BTileSlot<DX>::kFloats;
    constexpr int UVS = (2 * DX + 3) & ~3;
    int cpb, nblk;
    bsim2_geometry(a.N, M, cpb, nblk);
    const int NP = ((a.N + 127) / 128) * 128;
    const size_t lds = sizeof(float) * (2 * MQ::kSize + MG::kSize + 2 * (size_t)NP * PS + 4 * (size_t)(DX + 1) * NP +
                                        4 * (DX + 2) * 32 + 4 * 32 * UVS + 16 +
                                        (DX >= 3 ? 4 * ((2 * DX + DY) * 32 + 5 * DX * 8) : 0));
    if (lds > 160 * 1024) return PSVO_ERR_UNSUPPORTED;
    // pair phase of one wave, alone on its SIMD: 2560 cycles at N = 128, Dx = 2 (section timers), ~ N (3 Dx + 4)
    BsimBwdArgs as = a;
    as.skew = (int)(2560.0 * NP / 128.0 * (3 * DX + 4) / 10.0 * g_tune_skew_pct / 100.0);
    clear_hip_error();
    if (jm == 3 && DX == 2) {
        if constexpr (DX == 2)
            hipLaunchKernelGGL((bsim_bwd2_kernel<DX, DY, H, M, 3>), dim3(nblk, a.B), dim3(256), lds, stream, as);
    } else if (jm == 2 && DX == 2) {
        if constexpr (DX == 2)
            hipLaunchKernelGGL((bsim_bwd2_kernel<DX, DY, H, M, 2>), dim3(nblk, a.B), dim3(256), lds, stream, as);
    } else if (jm == 1 && DX == 2) {
        if constexpr (DX == 2)       // (A/B arm only: not instantiated for the shapes whose default is v1)
            hipLaunchKernelGGL((bsim_bwd2_kernel<DX, DY, H, M, 1>), dim3(nblk, a.B), dim3(256), lds, stream, as);
    } else {
        hipLaunchKernelGGL((bsim_bwd2_kernel<DX, DY, H, M, 0>), dim3(nblk, a.B), dim3(256), lds, stream, as);
    }
    (void)o;
    return launch_status();
}

template <int DX, int DY, int H>
static int bb2_dispatch_m(const BsimBwdArgs& a, const BsimBwdOut& o, int M, int jm, hipStream_t s) {
    switch (M) {
        case 4: return launch_bsim_bwd2<DX, DY, H, 4>(a, o, jm, s);
        case 8: return launch_bsim_bwd2<DX, DY, H, 8>(a, o, jm, s);
        case 16: return launch_bsim_bwd2<DX, DY, H, 16>(a, o, jm, s);
        case 32: return launch_bsim_bwd2<DX, DY, H, 32>(a, o, jm, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

template <int DX, int DY>
static int bb2_dispatch_h(const BsimBwdArgs& a, const BsimBwdOut& o, int H, int M, int jm, hipStream_t s) {
    switch (H) {
        case 16: return bb2_dispatch_m<DX, DY, 16>(a, o, M, jm, s);
        case 32: return bb2_dispatch_m<DX, DY, 32>(a, o, M, jm, s);
        case 64: return bb2_dispatch_m<DX, DY, 64>(a, o, M, jm, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

// external linkage: explicitly instantiated once per DX in bsim_bwd2_dx{2,3,4}.hip
template <int DX>
int bb2_dispatch_dy(const BsimBwdArgs& a, const BsimBwdOut& o, int Dy, int H, int M, int jm, hipStream_t s) {
    switch (Dy) {
        case 1: return bb2_dispatch_h<DX, 1>(a, o, H, M, jm, s);
        case 2: return bb2_dispatch_h<DX, 2>(a, o, H, M, jm, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // namespace psvo
