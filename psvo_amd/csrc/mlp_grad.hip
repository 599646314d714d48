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

#include <cstdlib>

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

// The default since round 3 (PSVO_WGRAD_OLD=1 selects the kernel above for A/B): the same arithmetic per (row, hidden unit),
// restructured around what the kernel above loses.  At C* its two large launches (MLP_f and MLP_g of the backward simulation,
// 1.3e7 rows each: 0.204 and 0.153 ms) are the tail of the step's critical path together with the encoder's reverse chain, at
// 44 % of the vector-issue rate: per row and block column it spends ~270 instructions -- 205 arithmetic ones, a 64-bit
// `r / L`, 16 v_readlane + s_nop for weights that hipcc keeps in spilled SGPRs -- waits for the row's four loads before the
// first of them, and every block column (H / 16 of them) reads all rows again (2 x 210 MB at C*).
//   * the four waves of a workgroup are (row group) x (block column): the columns of a workgroup read the SAME 64 rows
//     (one trip to HBM, the other waves hit the cache of the same CU), and each wave keeps only its column's accumulators;
//   * KC = 8 hidden units per column when Din + Dout > 4, so weights (<= 64 SGPRs) and accumulators (<= 73 VGPRs) fit
//     four waves per SIMD in every instantiation (the kernel above: 16 units whatever the size);
//   * (segment, row) advance incrementally with the grid stride -- no division in the loop -- and the next row's values are
//     requested before the current row's arithmetic.
// Counters of the result (profiles/r03_sq_counters_mlp_wgrad.json): the four waves of a SIMD are each 51 % of their cycles
// in a vector instruction (the kernel above: 34 %), 6 % parked on a load (21 %).  215 vector instructions per 64 rows x 16
// units, 128 of them the FMAs of the algorithm, at 1.62 ns per instruction and SIMD; a stream of nothing but independent
// v_fma_f32 runs at 1.22 ns at four waves per SIMD (tools/micro/valu_rate.hip, profiles/r03_valu_rate.txt: 107 TFLOP/s).
// (Packed f32 pairs -- v_pk_fma_f32 over two hidden units, weights and accumulators in register pairs -- were measured on
// the old structure first: 0.219 ms against 0.204 at C*; that build carried 34 canonicalising v_max and 24 s_nop per row
// and 192 VGPRs, two waves per SIMD.  The same micro-benchmark: at four waves per SIMD a packed stream sustains 111 TFLOP/s
// against 107 scalar -- the halved instruction count buys nothing once four waves share the SIMD; not repeated here.)
template <int DIN, int DOUT>
struct WgradCols {
    static constexpr int KC = (DIN + DOUT <= 4) ? 16 : 8;
};

template <int DIN, int DOUT>
__global__ void __launch_bounds__(256) mlp_wgrad_cols_kernel(const WgradArgs a, const int H, const int cw_shift) {
    constexpr int KC = WgradCols<DIN, DOUT>::KC;
    constexpr int NPC = (DIN + 1 + DOUT) * KC + DOUT;  // per-column partial sums (+ db2, kept from column 0)
    __shared__ float red[4][NPC];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CW = 1 << cw_shift, RW = 4 >> cw_shift;      // columns / row groups of a workgroup: wave = rg * CW + c
    const int c = wave & (CW - 1), rg = wave >> cw_shift;
    const int k0 = (blockIdx.y * CW + c) * KC;

    float w1[DIN][KC], b1[KC], w2[DOUT][KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        b1[k] = a.w.b1[k0 + k];
#pragma unroll
        for (int i = 0; i < DIN; ++i) w1[i][k] = a.w.W1[i * H + k0 + k];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) w2[o][k] = a.w.W2[(k0 + k) * DOUT + o];
    }
    // w1, w2 stay in scalar registers (<= 64); pre = fma(x, w1, b1) may read ONE of them, so b1 lives in vector registers.
    // (After all the loads: hipcc turns every uniform load that FOLLOWS an asm volatile into a per-lane global_load.)
#pragma unroll
    for (int k = 0; k < KC; ++k) keep_in_vgpr(b1[k]);

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

    const int L = a.L;
    const long long R = a.S * L;
    const long long stride = (long long)gridDim.x * RW * 64;
    const long long ds = stride / L;
    const int dl = (int)(stride - ds * L);
    const long long stepX = ds * DIN * L + dl, stepD = ds * DOUT * L + dl;
    long long r = ((long long)blockIdx.x * RW + rg) * 64 + lane;
    const long long s0 = r / L;
    int l = (int)(r - s0 * L);
    long long offX = s0 * DIN * L + l, offD = s0 * DOUT * L + l;

    float xn[DIN], dn[DOUT];
    auto request = [&]() {
#pragma unroll
        for (int i = 0; i < DIN; ++i) xn[i] = a.X[offX + (long long)i * L];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) dn[o] = a.dOut[offD + (long long)o * L];
    };
    if (r < R) request();
    while (r < R) {
        float x[DIN], dout[DOUT];
#pragma unroll
        for (int i = 0; i < DIN; ++i) x[i] = xn[i];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            dout[o] = dn[o];
            gb2[o] += dout[o];
        }
        r += stride;
        l += dl;
        offX += stepX;
        offD += stepD;
        if (l >= L) {          // the stride crossed one more segment boundary
            l -= L;
            offX += (long long)(DIN - 1) * L;
            offD += (long long)(DOUT - 1) * L;
        }
        if (r < R) request();

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

    // per-wave sums (the steps of wave_sum, each applied to ALL sums before the next: one value at a time every step waits
    // out the DPP hazard of the one before -- 510 s_nop in 1 900 instructions, a third of the launch at the filter's 8e5
    // rows), then the row groups of each column in fixed order
    float acc[NPC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
#pragma unroll
        for (int i = 0; i < DIN; ++i) acc[i * KC + k] = gW1[i][k];
        acc[DIN * KC + k] = gb1[k];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) acc[(DIN + 1 + o) * KC + k] = gW2[o][k];
    }
#pragma unroll
    for (int o = 0; o < DOUT; ++o) acc[(DIN + 1 + DOUT) * KC + o] = gb2[o];
    // (eight sums at a time: enough distance for the hazard, and the steps' temporaries stay within eight registers --
    // all NPC at once cost 46 more VGPRs and the fourth wave per SIMD)
#pragma unroll
    for (int p0 = 0; p0 < NPC; p0 += 8) {
        constexpr int kStep = 8;
#pragma unroll
        for (int p = p0; p < p0 + kStep && p < NPC; ++p) acc[p] += dpp_mov<0x111, 0xF, 0xF, true>(0.f, acc[p]);      // row_shr:1
#pragma unroll
        for (int p = p0; p < p0 + kStep && p < NPC; ++p) acc[p] += dpp_mov<0x112, 0xF, 0xF, true>(0.f, acc[p]);      // row_shr:2
#pragma unroll
        for (int p = p0; p < p0 + kStep && p < NPC; ++p) acc[p] += dpp_mov<0x114, 0xF, 0xF, true>(0.f, acc[p]);      // row_shr:4
#pragma unroll
        for (int p = p0; p < p0 + kStep && p < NPC; ++p) acc[p] += dpp_mov<0x118, 0xF, 0xF, true>(0.f, acc[p]);      // row_shr:8
#pragma unroll
        for (int p = p0; p < p0 + kStep && p < NPC; ++p) acc[p] += dpp_mov<0x142, 0xA, 0xF, false>(0.f, acc[p]);     // row_bcast:15
#pragma unroll
        for (int p = p0; p < p0 + kStep && p < NPC; ++p) acc[p] += dpp_mov<0x143, 0xC, 0xF, false>(0.f, acc[p]);     // row_bcast:31
#pragma unroll
        for (int p = p0; p < p0 + kStep && p < NPC; ++p) asm volatile("" : "+v"(acc[p]));     // (groups stay apart)
    }
    if (lane == 63) {
#pragma unroll
        for (int p = 0; p < NPC; ++p) red[wave][p] = acc[p];
    }
    __syncthreads();
    // scatter into the flat keras-layout vector dW1 (DIN,H) | db1 (H) | dW2 (H,DOUT) | db2 (DOUT)
    const int NP = DIN * H + H + H * DOUT + DOUT;
    float* dst = a.partial + (size_t)blockIdx.x * NP;
    for (int q = tid; q < CW * NPC; q += 256) {
        const int cc = q / NPC, p = q - cc * NPC;
        float v = red[cc][p];
        for (int g = 1; g < RW; ++g) v += red[g * CW + cc][p];
        const int kc0 = (blockIdx.y * CW + cc) * KC;
        if (p < (DIN + 1 + DOUT) * KC) {
            const int row = p / KC, k = p - row * KC;
            if (row < DIN) dst[row * H + kc0 + k] = v;
            else if (row == DIN) dst[DIN * H + kc0 + k] = v;
            else dst[DIN * H + H + (kc0 + k) * DOUT + (row - DIN - 1)] = v;
        } else if (kc0 == 0) {
            dst[DIN * H + H + H * DOUT + (p - (DIN + 1 + DOUT) * KC)] = v;
        }
    }
}

static bool wgrad_old_kernel() {
    const char* e = getenv("PSVO_WGRAD_OLD");
    return e && e[0] == '1';
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
    if (wgrad_old_kernel()) {
        hipLaunchKernelGGL((mlp_wgrad_kernel<DIN, DOUT>), dim3(nblk, H / kKC), dim3(256), 0, s, a, H);
    } else {
        const int ncol = H / WgradCols<DIN, DOUT>::KC;
        const int sh = ncol % 4 == 0 ? 2 : ncol % 2 == 0 ? 1 : 0;
        hipLaunchKernelGGL((mlp_wgrad_cols_kernel<DIN, DOUT>), dim3(nblk, ncol >> sh), dim3(256), 0, s, a, H, sh);
    }
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

// ---------------------------------------------------------------------------------------------
// Two hidden layers (psvo_desc.layers == 2): mu = relu(relu(x W1 + b1) Wh + bh) W2 + b2.
// Per row the H x H layer costs H^2 FMAs forward, H^2 for d h1 = (d pre2) Wh^T and H^2 for dWh += h1^T (d pre2): three
// GEMMs with M = rows, N = K = H (and K = rows for the last) -- the one place of the path where the f32 matrix instruction
// has full tiles.  A workgroup walks chunks of 64 rows:
//   A (VALU)  h1 = relu(x W1 + b1) -> LDS [64][H]                      thread = (row, quarter of the units)
//   1 (MFMA)  pre2 = h1 Wh + bh    wave w owns rows 16 w .. 16 w + 15 (v_mfma_f32_16x16x4_f32, operands read from LDS);
//             in the accumulator layout: h2, d pre2 = [pre2 > 0] (dOut W2^T) -> LDS, dW2 += h2^T dOut, dbh += d pre2
//   2 (MFMA)  d h1 = (d pre2) Wh^T ; * [h1 > 0] ; dW1 += x^T d h1, db1 += d h1   (accumulator layout, registers)
//   3 (MFMA)  dWh += h1^T (d pre2)  (K = the 64 rows of the chunk), accumulators persistent over the chunks
// and writes one partial per workgroup, folded by reduce_partials_kernel (fixed order, no atomics).
// Output layout (keras order hidden_0, hidden_1, mu_layer): [dW1 (DIN,H) | db1 (H) | dWh (H,H) | dbh (H) | dW2 (H,DOUT) | db2].
// ---------------------------------------------------------------------------------------------
typedef float wg_f4 __attribute__((ext_vector_type(4)));

template <int DIN, int DOUT, int H>
__global__ void __launch_bounds__(256) mlp2_wgrad_kernel(const WgradArgs a) {
    constexpr int NCT = H / 16;               // 16-wide column tiles of an H-wide matrix
    constexpr int LDH = H + 1;                // padded LDS row
    constexpr int TPW = (NCT * NCT) / 4;      // dWh tiles per wave (H = 64: 4, H = 32: 1)
    static_assert(H == 32 || H == 64, "two-layer weight gradients: H in {32, 64}");
    constexpr int oB1 = DIN * H, oWh = oB1 + H, oBh = oWh + H * H, oW2 = oBh + H, oB2 = oW2 + H * DOUT, NP2 = oB2 + DOUT;
    constexpr int NV = DIN + 2 + DOUT;        // per-column sums: dW1 rows, db1, dbh, dW2 columns
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Whs = sm;                 // [H][LDH]
    float* H1s = Whs + H * LDH;      // [64][LDH]
    float* D2s = H1s + 64 * LDH;     // [64][LDH]   d pre2
    float* xs = D2s + 64 * LDH;      // [64][DIN]
    float* ds = xs + 64 * DIN;       // [64][DOUT]
    float* w1s = ds + 64 * DOUT;     // W1 [DIN][H] | b1 [H]
    static_assert((H + 128) * LDH >= 16 * NV * H, "final reduction reuses the tile buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    for (int i = tid; i < H * H; i += 256) Whs[(i / H) * LDH + (i % H)] = a.w.Wh[i];
    for (int i = tid; i < DIN * H; i += 256) w1s[i] = a.w.W1[i];
    for (int i = tid; i < H; i += 256) w1s[DIN * H + i] = a.w.b1[i];

    // loop-invariant per lane: columns j = c * 16 + li of W2 and bh
    float w2r[NCT][DOUT], bhr[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        bhr[c] = a.w.bh[c * 16 + li];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) w2r[c][o] = a.w.W2[(c * 16 + li) * DOUT + o];
    }
    wg_f4 accWh[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) accWh[t] = wg_f4{0.f, 0.f, 0.f, 0.f};
    float gW2[NCT][DOUT], gW1[NCT][DIN], gbh[NCT], gb1[NCT], gb2[DOUT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        gbh[c] = gb1[c] = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) gW2[c][o] = 0.f;
#pragma unroll
        for (int d = 0; d < DIN; ++d) gW1[c][d] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < DOUT; ++o) gb2[o] = 0.f;
    __syncthreads();

    const long long R = a.S * a.L;
    const long long nchunk = (R + 63) / 64;
    for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        // ---- A: rows of the chunk, first layer ---------------------------------------------------
        {
            const long long r = ch * 64 + lane;
            const bool in = r < R;
            const long long sg = in ? r / a.L : 0;
            const int l = in ? (int)(r - sg * a.L) : 0;
            float x[DIN];
#pragma unroll
            for (int d = 0; d < DIN; ++d) x[d] = in ? a.X[(sg * DIN + d) * a.L + l] : 0.f;
            if (wave == 0) {
#pragma unroll
                for (int d = 0; d < DIN; ++d) xs[lane * DIN + d] = x[d];
#pragma unroll
                for (int o = 0; o < DOUT; ++o) {
                    const float g = in ? a.dOut[(sg * DOUT + o) * a.L + l] : 0.f;   // (rows past the end contribute nothing)
                    ds[lane * DOUT + o] = g;
                    gb2[o] += g;
                }
            }
#pragma unroll
            for (int kk = 0; kk < H / 4; ++kk) {
                const int k = wave * (H / 4) + kk;
                float pre = w1s[DIN * H + k];
#pragma unroll
                for (int d = 0; d < DIN; ++d) pre = fmaf(x[d], w1s[d * H + k], pre);
                H1s[lane * LDH + k] = fmaxf(pre, 0.f);
            }
        }
        __syncthreads();
        // ---- 1: pre2 = h1 Wh + bh for rows 16 wave .. + 15 ----------------------------------------
        wg_f4 acc[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = wg_f4{bhr[c], bhr[c], bhr[c], bhr[c]};
#pragma unroll 4
        for (int k0 = 0; k0 < H; k0 += 4) {
            const float av = H1s[(wave * 16 + li) * LDH + k0 + lg];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float bv = Whs[(k0 + lg) * LDH + c * 16 + li];
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
            }
        }
        // accumulator layout: row i = 16 wave + 4 lg + r, column j = 16 c + li
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 16 + 4 * lg + r;
            float dout[DOUT];
#pragma unroll
            for (int o = 0; o < DOUT; ++o) dout[o] = ds[i * DOUT + o];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float pre2 = acc[c][r];
                const float h2 = fmaxf(pre2, 0.f);
                float dp = 0.f;
#pragma unroll
                for (int o = 0; o < DOUT; ++o) {
                    dp = fmaf(dout[o], w2r[c][o], dp);
                    gW2[c][o] = fmaf(h2, dout[o], gW2[c][o]);
                }
                dp = pre2 > 0.f ? dp : 0.f;
                gbh[c] += dp;
                D2s[i * LDH + c * 16 + li] = dp;
            }
        }
        // (rows 16 wave .. + 15 of D2s are this wave's own: no workgroup barrier before step 2)
        // ---- 2: d h1 = (d pre2) Wh^T, masked; dW1, db1 -----------------------------------------------
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = wg_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k0 = 0; k0 < H; k0 += 4) {
            const float av = D2s[(wave * 16 + li) * LDH + k0 + lg];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float bv = Whs[(c * 16 + li) * LDH + k0 + lg];     // B[k][j] = Wh[j][k]
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 16 + 4 * lg + r;
            float x[DIN];
#pragma unroll
            for (int d = 0; d < DIN; ++d) x[d] = xs[i * DIN + d];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float dh = H1s[i * LDH + c * 16 + li] > 0.f ? acc[c][r] : 0.f;
                gb1[c] += dh;
#pragma unroll
                for (int d = 0; d < DIN; ++d) gW1[c][d] = fmaf(x[d], dh, gW1[c][d]);
            }
        }
        __syncthreads();
        // ---- 3: dWh += h1^T (d pre2) over the 64 rows ------------------------------------------------
#pragma unroll 4
        for (int k0 = 0; k0 < 64; k0 += 4) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tl = wave * TPW + t, rt = tl / NCT, ct = tl % NCT;
                const float av = H1s[(k0 + lg) * LDH + rt * 16 + li];     // A[i][k] = h1[row k][unit i]
                const float bv = D2s[(k0 + lg) * LDH + ct * 16 + li];
                accWh[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accWh[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- the workgroup's partial ------------------------------------------------------------------
    float* dst = a.partial + (size_t)blockIdx.x * NP2;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tl = wave * TPW + t, rt = tl / NCT, ct = tl % NCT;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[oWh + (rt * 16 + 4 * lg + r) * H + ct * 16 + li] = accWh[t][r];
    }
    float* red = sm;     // [16 groups][NV][H]: every column sum exists once per (wave, lane group)
    const int grp = wave * 4 + lg;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int j = c * 16 + li;
        float* rg = red + (size_t)grp * NV * H;
#pragma unroll
        for (int d = 0; d < DIN; ++d) rg[d * H + j] = gW1[c][d];
        rg[DIN * H + j] = gb1[c];
        rg[(DIN + 1) * H + j] = gbh[c];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) rg[(DIN + 2 + o) * H + j] = gW2[c][o];
    }
    __syncthreads();
    for (int p = tid; p < NV * H; p += 256) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += red[(size_t)g * NV * H + p];
        const int row = p / H, j = p - row * H;
        if (row < DIN) dst[row * H + j] = v;
        else if (row == DIN) dst[oB1 + j] = v;
        else if (row == DIN + 1) dst[oBh + j] = v;
        else dst[oW2 + j * DOUT + (row - DIN - 2)] = v;
    }
    if (wave == 0) {
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            const float v = wave_sum(gb2[o]);
            if (lane == 0) dst[oB2 + o] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same three products on the bf16 matrix instructions with every f32 operand SPLIT into NP bf16 pieces
// (x ~ hi + lo (+ lo2), each piece the round-to-nearest bf16 of the remainder: 8 significand bits per piece), so that the f32
// product is carried to ~2^-17 (NP = 2: hh + hl + lh) or ~2^-24 (NP = 3: hh, hm, mh, hl, lh, mm) relative -- psvo_set_tuning(
// PSVO_TUNE_WGRAD2, 2 | 3); 0 = the f32 instruction above (default until measured faster).  Why: v_mfma_f32_16x16x4_f32 runs at
// 64 flop / cycle / SIMD and does not co-execute with the VALU on gfx950 (profiles/r02_bsim_bwd_ab.md); the bf16 instructions
// run at 1024 flop / cycle / SIMD and do -- six of them per f32 product are still 2.7 x less pipe time, three 5.3 x, and here
// K = H = 32 / 64 (phases 1, 2) or K = the 16 rows of a wave (phase 3) amortises the split.
//   * Wh is split ONCE per workgroup into two LDS operand images (K = first index for pre2 = h1 Wh, K = second index for
//     d h1 = d pre2 Wh^T): lane (j, kslot) reads its eight K values of one piece with one ds_read_b128;
//   * h1 / d pre2 rows are split on the fly when a wave reads its A operand (8 consecutive floats of its own row);
//   * phase 3 (dWh += h1^T d pre2, K = rows) runs on v_mfma_f32_16x16x16_bf16 with K = the wave's OWN 16 rows: its B operand
//     -- d pre2 of rows 4 g + e, column li -- is the accumulator layout phase 1 just left in the lane's registers, so it never
//     touches LDS, needs no workgroup barrier, and every wave keeps a partial of all H x H / 256 tiles (summed over the four
//     waves once, at the end).
// ---------------------------------------------------------------------------------------------
typedef __bf16 wg_bf8 __attribute__((ext_vector_type(8)));
typedef short wg_s4 __attribute__((ext_vector_type(4)));

// two f32 -> NP packed bf16 pairs (element 0 in the low half).  Every piece is the ROUND-TO-NEAREST bf16 of what the pieces
// before it left (v_cvt_pk_bf16_f32; the remainder x - hi is exact in f32), so the error of the sum of the pieces is
// unbiased: truncated pieces are all short of the value by up to 2^-16 (NP = 2), an error that adds up over the K = 64 products
// of a dot product and over the 10^7 rows of a weight gradient instead of averaging out (the dW1 of the first version failed
// the 1e-4 parity bar for that reason).
typedef __bf16 wg_bf2 __attribute__((ext_vector_type(2)));
typedef float wg_f2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f2{a, b}, wg_bf2));
        out[p] = pk;
        if (p + 1 < NP) {
            a -= __uint_as_float(pk << 16);
            b -= __uint_as_float(pk & 0xffff0000u);
        }
    }
}
// the piece products of one f32 product, smallest first: index pairs (piece of A, piece of B)
template <int NP> struct PieceProducts;
template <> struct PieceProducts<2> { static constexpr int N = 3; static constexpr int a[3] = {0, 1, 0}, b[3] = {1, 0, 0}; };
template <> struct PieceProducts<3> {
    static constexpr int N = 6;
    static constexpr int a[6] = {1, 0, 2, 0, 1, 0}, b[6] = {1, 2, 0, 1, 0, 0};
};

template <int DIN, int DOUT, int H, int NP>
__global__ void __launch_bounds__(256) mlp2_wgrad_bf16_kernel(const WgradArgs a) {
    using PP = PieceProducts<NP>;
    constexpr int NCT = H / 16;               // 16-wide column tiles of an H-wide matrix
    constexpr int NKB = H / 32;               // K blocks of 32 (phases 1, 2)
    constexpr int LDH = H + 4;                // padded LDS row, 16-byte aligned
    constexpr int oB1 = DIN * H, oWh = oB1 + H, oBh = oWh + H * H, oW2 = oBh + H, oB2 = oW2 + H * DOUT, NP2 = oB2 + DOUT;
    constexpr int NV = DIN + 2 + DOUT;        // per-column sums: dW1 rows, db1, dbh, dW2 columns
    constexpr int IMG = NKB * NCT * 64 * 4;   // dwords of one piece image of Wh (16 B per lane, K block and column tile)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* H1s = sm;                          // [64][LDH]
    float* D2s = H1s + 64 * LDH;              // [64][LDH]   d pre2
    unsigned* WB1 = reinterpret_cast<unsigned*>(D2s + 64 * LDH);   // [NP][NKB][NCT][64][4]  B of pre2 = h1 Wh   (K = k of Wh[k][j])
    unsigned* WB2 = WB1 + NP * IMG;                                 // [NP][NKB][NCT][64][4]  B of d h1 = d pre2 Wh^T (K = k of Wh[j][k])
    float* xs = reinterpret_cast<float*>(WB2 + NP * IMG);          // [64][DIN]
    float* ds = xs + 64 * DIN;                // [64][DOUT]
    float* w1s = ds + 64 * DOUT;              // W1 [DIN][H] | b1 [H]

    const int tid = threadIdx.x, lane = tid & 63;
    // (the wave index as a SCALAR: `if (wave == 0)` on a per-lane value becomes an EXEC-masked region, and under this kernel's
    //  register pressure hipcc 7.2 put live-range-split copies ahead of its EXEC restore -- tools/exec_restore_check.py)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // ---- the two operand images of Wh: entry (kb, c, lane) = eight K values of column / row 16 c + li ----------------
    for (int i = tid; i < NKB * NCT * 64; i += 256) {
        const int ln = i & 63, c = (i >> 6) % NCT, kb = (i >> 6) / NCT;
        const int j = 16 * c + (ln & 15), k0 = 32 * kb + 8 * (ln >> 4);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            unsigned q1[NP], q2[NP];
            split_pair<NP>(a.w.Wh[(k0 + e) * H + j], a.w.Wh[(k0 + e + 1) * H + j], q1);
            split_pair<NP>(a.w.Wh[j * H + k0 + e], a.w.Wh[j * H + k0 + e + 1], q2);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                WB1[p * IMG + i * 4 + e / 2] = q1[p];
                WB2[p * IMG + i * 4 + e / 2] = q2[p];
            }
        }
    }
    for (int i = tid; i < DIN * H; i += 256) w1s[i] = a.w.W1[i];
    for (int i = tid; i < H; i += 256) w1s[DIN * H + i] = a.w.b1[i];

    float w2r[NCT][DOUT], bhr[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        bhr[c] = a.w.bh[c * 16 + li];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) w2r[c][o] = a.w.W2[(c * 16 + li) * DOUT + o];
    }
    wg_f4 accWh[NCT][NCT];      // this wave's partial of dWh: tile (rt, ct), rows = units of h1, columns = units of d pre2
#pragma unroll
    for (int rt = 0; rt < NCT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) accWh[rt][ct] = wg_f4{0.f, 0.f, 0.f, 0.f};
    float gW2[NCT][DOUT], gW1[NCT][DIN], gbh[NCT], gb1[NCT], gb2[DOUT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        gbh[c] = gb1[c] = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) gW2[c][o] = 0.f;
#pragma unroll
        for (int d = 0; d < DIN; ++d) gW1[c][d] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < DOUT; ++o) gb2[o] = 0.f;
    __syncthreads();

    // A operand of a K block: eight consecutive floats of the lane's row, split into NP pieces of four packed words
    auto a_operand = [&](const float* row8, wg_bf8 (&out)[NP]) {
        const float4 v0 = *reinterpret_cast<const float4*>(row8), v1 = *reinterpret_cast<const float4*>(row8 + 4);
        unsigned q[4][NP];
        split_pair<NP>(v0.x, v0.y, q[0]);
        split_pair<NP>(v0.z, v0.w, q[1]);
        split_pair<NP>(v1.x, v1.y, q[2]);
        split_pair<NP>(v1.z, v1.w, q[3]);
#pragma unroll
        for (int p = 0; p < NP; ++p) out[p] = __builtin_bit_cast(wg_bf8, make_uint4(q[0][p], q[1][p], q[2][p], q[3][p]));
    };

    const long long R = a.S * a.L;
    const long long nchunk = (R + 63) / 64;
    for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
        // ---- A: rows of the chunk, first layer ---------------------------------------------------
        {
            const long long r = ch * 64 + lane;
            const bool in = r < R;
            const long long sg = in ? r / a.L : 0;
            const int l = in ? (int)(r - sg * a.L) : 0;
            float x[DIN];
#pragma unroll
            for (int d = 0; d < DIN; ++d) x[d] = in ? a.X[(sg * DIN + d) * a.L + l] : 0.f;
            if (wave == 0) {
#pragma unroll
                for (int d = 0; d < DIN; ++d) xs[lane * DIN + d] = x[d];
#pragma unroll
                for (int o = 0; o < DOUT; ++o) {
                    const float g = in ? a.dOut[(sg * DOUT + o) * a.L + l] : 0.f;
                    ds[lane * DOUT + o] = g;
                    gb2[o] += g;
                }
            }
#pragma unroll
            for (int kk = 0; kk < H / 4; ++kk) {
                const int k = wave * (H / 4) + kk;
                float pre = w1s[DIN * H + k];
#pragma unroll
                for (int d = 0; d < DIN; ++d) pre = fmaf(x[d], w1s[d * H + k], pre);
                H1s[lane * LDH + k] = fmaxf(pre, 0.f);
            }
        }
        __syncthreads();
        // ---- 1: pre2 = h1 Wh + bh for rows 16 wave .. + 15 ----------------------------------------
        wg_f4 acc[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = wg_f4{bhr[c], bhr[c], bhr[c], bhr[c]};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            wg_bf8 ap[NP];
            a_operand(H1s + (wave * 16 + li) * LDH + 32 * kb + 8 * lg, ap);
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                wg_bf8 bp[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bp[p] = __builtin_bit_cast(wg_bf8, *reinterpret_cast<const uint4*>(WB1 + p * IMG + ((kb * NCT + c) * 64 + lane) * 4));
#pragma unroll
                for (int q = 0; q < PP::N; ++q)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PP::a[q]], bp[PP::b[q]], acc[c], 0, 0, 0);
            }
        }
        // accumulator layout: row i = 16 wave + 4 lg + r, column j = 16 c + li
        float dpr[NCT][4];          // d pre2 of (row 4 lg + r, column 16 c + li): phase 3's B operand as it stands
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 16 + 4 * lg + r;
            float dout[DOUT];
#pragma unroll
            for (int o = 0; o < DOUT; ++o) dout[o] = ds[i * DOUT + o];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float pre2 = acc[c][r];
                const float h2 = fmaxf(pre2, 0.f);
                float dp = 0.f;
#pragma unroll
                for (int o = 0; o < DOUT; ++o) {
                    dp = fmaf(dout[o], w2r[c][o], dp);
                    gW2[c][o] = fmaf(h2, dout[o], gW2[c][o]);
                }
                dp = pre2 > 0.f ? dp : 0.f;
                gbh[c] += dp;
                dpr[c][r] = dp;
                D2s[i * LDH + c * 16 + li] = dp;
            }
        }
        // (rows 16 wave .. + 15 of D2s are this wave's own: no workgroup barrier before step 2)
        // ---- 2: d h1 = (d pre2) Wh^T, masked; dW1, db1 -----------------------------------------------
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = wg_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            wg_bf8 ap[NP];
            a_operand(D2s + (wave * 16 + li) * LDH + 32 * kb + 8 * lg, ap);
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                wg_bf8 bp[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bp[p] = __builtin_bit_cast(wg_bf8, *reinterpret_cast<const uint4*>(WB2 + p * IMG + ((kb * NCT + c) * 64 + lane) * 4));
#pragma unroll
                for (int q = 0; q < PP::N; ++q)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PP::a[q]], bp[PP::b[q]], acc[c], 0, 0, 0);
            }
        }
        float h1r[NCT][4];          // h1 of (row 4 lg + r, unit 16 c + li): phase 3's A operand
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 16 + 4 * lg + r;
            float x[DIN];
#pragma unroll
            for (int d = 0; d < DIN; ++d) x[d] = xs[i * DIN + d];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float h1v = H1s[i * LDH + c * 16 + li];
                h1r[c][r] = h1v;
                const float dh = h1v > 0.f ? acc[c][r] : 0.f;
                gb1[c] += dh;
#pragma unroll
                for (int d = 0; d < DIN; ++d) gW1[c][d] = fmaf(x[d], dh, gW1[c][d]);
            }
        }
        // ---- 3: dWh += h1^T (d pre2) over the wave's own 16 rows (K = 4 lg + e) --------------------------
        {
            wg_s4 ha[NCT][NP], db[NCT][NP];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                unsigned qa[2][NP], qb[2][NP];
                split_pair<NP>(h1r[c][0], h1r[c][1], qa[0]);
                split_pair<NP>(h1r[c][2], h1r[c][3], qa[1]);
                split_pair<NP>(dpr[c][0], dpr[c][1], qb[0]);
                split_pair<NP>(dpr[c][2], dpr[c][3], qb[1]);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    ha[c][p] = __builtin_bit_cast(wg_s4, make_uint2(qa[0][p], qa[1][p]));
                    db[c][p] = __builtin_bit_cast(wg_s4, make_uint2(qb[0][p], qb[1][p]));
                }
            }
#pragma unroll
            for (int rt = 0; rt < NCT; ++rt)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int q = 0; q < PP::N; ++q)
                        accWh[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha[rt][PP::a[q]], db[ct][PP::b[q]],
                                                                                  accWh[rt][ct], 0, 0, 0);
        }
        __syncthreads();      // H1s / xs / ds are rewritten by the next chunk
    }

    // ---- the workgroup's partial ------------------------------------------------------------------
    float* dst = a.partial + (size_t)blockIdx.x * NP2;
    float* red = sm;     // dWh: [4 waves][H][H]; then the column sums [16 groups][NV][H]
    static_assert(2 * 64 * LDH + 2 * NP * IMG >= 4 * H * H && 2 * 64 * LDH + 2 * NP * IMG >= 16 * NV * H,
                  "final reductions reuse the tile buffers");
#pragma unroll
    for (int rt = 0; rt < NCT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * H * H + (rt * 16 + 4 * lg + r) * H + ct * 16 + li] = accWh[rt][ct][r];
    __syncthreads();
    for (int p = tid; p < H * H; p += 256) dst[oWh + p] = (red[p] + red[H * H + p]) + (red[2 * H * H + p] + red[3 * H * H + p]);
    __syncthreads();
    const int grp = wave * 4 + lg;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int j = c * 16 + li;
        float* rg = red + (size_t)grp * NV * H;
#pragma unroll
        for (int d = 0; d < DIN; ++d) rg[d * H + j] = gW1[c][d];
        rg[DIN * H + j] = gb1[c];
        rg[(DIN + 1) * H + j] = gbh[c];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) rg[(DIN + 2 + o) * H + j] = gW2[c][o];
    }
    __syncthreads();
    for (int p = tid; p < NV * H; p += 256) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += red[(size_t)g * NV * H + p];
        const int row = p / H, j = p - row * H;
        if (row < DIN) dst[row * H + j] = v;
        else if (row == DIN) dst[oB1 + j] = v;
        else if (row == DIN + 1) dst[oBh + j] = v;
        else dst[oW2 + j * DOUT + (row - DIN - 2)] = v;
    }
    if (wave == 0) {
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            const float v = wave_sum(gb2[o]);
            if (lane == 0) dst[oB2 + o] = v;
        }
    }
}

int g_tune_wgrad2 = 0;      // psvo_set_tuning(PSVO_TUNE_WGRAD2, 0 | 2 | 3)

template <int DIN, int DOUT, int H, int NP>
static void launch_wgrad2_bf16(const WgradArgs& a, int nblk, hipStream_t s) {
    const size_t lds = sizeof(float) * ((size_t)2 * 64 * (H + 4) + 2 * NP * (H / 32) * (H / 16) * 64 * 4 + 64 * (DIN + DOUT) +
                                        (DIN + 1) * H);
    hipLaunchKernelGGL((mlp2_wgrad_bf16_kernel<DIN, DOUT, H, NP>), dim3(nblk), dim3(256), lds, s, a);
}

template <int DIN, int DOUT>
static int launch_wgrad2(const WgradArgs& a, int H, int nblk, float* out, int accumulate, hipStream_t s) {
    const int NP2 = DIN * H + H + H * H + H + H * DOUT + DOUT;
    const size_t lds = sizeof(float) * ((size_t)(H + 128) * (H + 1) + 64 * (DIN + DOUT) + (DIN + 1) * H);
    clear_hip_error();
    if (H != 32 && H != 64) return PSVO_ERR_UNSUPPORTED;
    if (g_tune_wgrad2 == 2) {
        if (H == 32) launch_wgrad2_bf16<DIN, DOUT, 32, 2>(a, nblk, s);
        else launch_wgrad2_bf16<DIN, DOUT, 64, 2>(a, nblk, s);
    } else if (g_tune_wgrad2 == 3) {
        if (H == 32) launch_wgrad2_bf16<DIN, DOUT, 32, 3>(a, nblk, s);
        else launch_wgrad2_bf16<DIN, DOUT, 64, 3>(a, nblk, s);
    } else if (H == 32) hipLaunchKernelGGL((mlp2_wgrad_kernel<DIN, DOUT, 32>), dim3(nblk), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((mlp2_wgrad_kernel<DIN, DOUT, 64>), dim3(nblk), dim3(256), lds, s, a);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(NP2), dim3(64), 0, s, a.partial, nblk, NP2, out, accumulate);
    return launch_status();
}

template <int DIN>
static int wgrad2_dispatch_out(const WgradArgs& a, int H, int Dout, int nblk, float* out, int acc, hipStream_t s) {
    switch (Dout) {
        case 1: return launch_wgrad2<DIN, 1>(a, H, nblk, out, acc, s);
        case 2: return launch_wgrad2<DIN, 2>(a, H, nblk, out, acc, s);
        case 3: return launch_wgrad2<DIN, 3>(a, H, nblk, out, acc, s);
        case 4: return launch_wgrad2<DIN, 4>(a, H, nblk, out, acc, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

}  // namespace psvo

extern "C" int psvo_mlp2_wgrad_blocks(long long rows) {
    long long nb = (rows + 64 * 4 - 1) / (64 * 4);   // >= 4 chunks of 64 rows per workgroup amortise the final reduction
    if (nb < 1) nb = 1;
    if (nb > 768) nb = 768;                          // three workgroups per CU (50 KB of LDS each at H = 64)
    return (int)nb;
}

extern "C" int psvo_mlp2_wgrad(long long S, int L, int Din, int H, int Dout, const float* X, const float* dOut,
                               const psvo_mlp* w, float* partial, float* grad, int accumulate, void* stream) {
    using namespace psvo;
    if (!X || !dOut || !w || !w->Wh || !w->bh || !partial || !grad || S <= 0 || L <= 0) return PSVO_ERR_INVALID;
    WgradArgs a{S, L, X, dOut, *w, partial};
    const int nblk = psvo_mlp2_wgrad_blocks(S * L);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (Din) {
        case 1: return wgrad2_dispatch_out<1>(a, H, Dout, nblk, grad, accumulate, s);
        case 2: return wgrad2_dispatch_out<2>(a, H, Dout, nblk, grad, accumulate, s);
        case 3: return wgrad2_dispatch_out<3>(a, H, Dout, nblk, grad, accumulate, s);
        case 4: return wgrad2_dispatch_out<4>(a, H, Dout, nblk, grad, accumulate, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

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
