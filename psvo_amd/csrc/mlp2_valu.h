// The round-2 evaluation of the two-hidden-layer per-particle MLP on the vector ALU (one lane = one row, Wh broadcast from
// LDS, 4 FMAs per ds_read_b128): kept for A/B measurements only (-DPSVO_L2_VALU; the product build uses the matrix-pipe form
// in common.h).  Included by common.h inside namespace psvo.
#pragma once
// ---------------------------------------------------------------------------------------------
// Two hidden layers of width H (reference src/transformation/MLP.py:24-38,50-54 with *_layers = "H,H"):
//     mu = relu(relu(x W1 + b1) Wh + bh) W2 + b2.
// LDS image: the one-layer image, then Wh[H][H] (keras (in, out), row-major) | bh[H].
// Same interface as the one-layer struct, so the persistent kernels only carry L as a template parameter.  The first
// layer (H * DIN FMAs) is evaluated in full by every lane; the H x H layer is walked in groups of four units of the
// second layer (one wave-uniform float4 of Wh per first-layer unit: 4 FMAs per LDS read), in a rolled loop -- h1 stays
// in H VGPRs, the reverse pass adds H accumulators for d h1.  With S lanes per evaluation (eval_part / bwd_input_part) a
// lane owns H / S units of the SECOND layer; the caller sums the outputs / input gradients over the S lanes as before.
// ---------------------------------------------------------------------------------------------
template <int DIN, int H, int DOUT>
struct MlpLds<DIN, H, DOUT, 2> {
    static constexpr int kW1 = 0;
    static constexpr int kB1 = DIN * H;
    static constexpr int kW2 = kB1 + H;
    static constexpr int kB2 = kW2 + DOUT * H;
    static constexpr int kWh = kB2 + ((DOUT + 3) & ~3);
    static constexpr int kBh = kWh + H * H;
    static constexpr int kSize = kBh + H;
    static_assert(H % 4 == 0, "hidden width must be a multiple of 4");

    __device__ static void load(float* __restrict__ w, const psvo_mlp& p, int tid, int nthreads) {
        MlpLds<DIN, H, DOUT, 1>::load(w, p, tid, nthreads);
        for (int i = tid; i < H * H; i += nthreads) w[kWh + i] = p.Wh[i];
        for (int i = tid; i < H; i += nthreads) w[kBh + i] = p.bh[i];
    }

    // h = relu(x W1 + b1), all H units, kept as H / 2 register pairs (h[2k], h[2k+1]): the H x H layer multiplies by a splat
    // of one half, which v_pk_fma_f32 takes with op_sel -- a plain float h[] makes hipcc build the {h_i, h_i} pairs with
    // v_mov and hoist all H of them out of the rolled loop over the second layer (2 H extra registers)
    __device__ __forceinline__ static void hidden1(const float* __restrict__ w, const float (&x)[DIN], f2 (&h)[H / 2]) {
        const f2 zero = f2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 b = *reinterpret_cast<const float4*>(w + kB1 + k);
            f2 ha = f2{b.x, b.y}, hb = f2{b.z, b.w};
#pragma unroll
            for (int i = 0; i < DIN; ++i) {
                const float4 wi = *reinterpret_cast<const float4*>(w + kW1 + i * H + k);
                const f2 xi = f2{x[i], x[i]};
                ha = pk_fma(xi, f2{wi.x, wi.y}, ha);
                hb = pk_fma(xi, f2{wi.z, wi.w}, hb);
            }
            h[k / 2] = pk_max(ha, zero);
            h[k / 2 + 1] = pk_max(hb, zero);
        }
    }

    // pre-activations of second-layer units j0 .. j0+3
    __device__ __forceinline__ static void pre2_group4(const float* __restrict__ w, int j0, const f2 (&h)[H / 2], f2& pa,
                                                       f2& pb) {
        const float4 b = *reinterpret_cast<const float4*>(w + kBh + j0);
        pa = f2{b.x, b.y};
        pb = f2{b.z, b.w};
        const float* wh = w + kWh + j0;
#pragma unroll
        for (int i = 0; i < H; i += 2) {
            const float4 w0 = *reinterpret_cast<const float4*>(wh + i * H);
            const float4 w1 = *reinterpret_cast<const float4*>(wh + (i + 1) * H);
            pa = pk_fma_bcast<0>(h[i / 2], f2{w0.x, w0.y}, pa);
            pb = pk_fma_bcast<0>(h[i / 2], f2{w0.z, w0.w}, pb);
            pa = pk_fma_bcast<1>(h[i / 2], f2{w1.x, w1.y}, pa);
            pb = pk_fma_bcast<1>(h[i / 2], f2{w1.z, w1.w}, pb);
            // (the scheduler otherwise issues all H float4 reads ahead of the FMAs: 4 H registers in flight)
            if ((i & 7) == 6) __builtin_amdgcn_sched_barrier(0);
        }
    }

    // The weights are loop-invariant LDS data of a persistent kernel: left visible, hipcc hoists the (DIN + 1) * H reads of the
    // first layer of EVERY MLP out of the time loop (fine for one narrow layer; here it is 3 x 192 registers at H = 64 on top
    // of h and d h: 960 spilled VGPRs in filter_bwd).  An opaque zero offset per call keeps the reads where they are used.
    // What remains: a result that is only consumed after the NEXT MLP call (dx is the running sum of several calls) has its
    // tail -- the reads of h and d h -- sunk behind that call's loop, which keeps 2 H registers of every earlier call alive in
    // it (+230 live values per call at H = 64).  Pinning the results with an `asm volatile("" : "+v"(v))` stops the sinking
    // (-DPSVO_PIN_RESULTS: 177 registers whatever the number of calls); it is off because these units are built with the
    // basic register allocator (build.py: L2_FLAGS), where it changes little.
#if defined(PSVO_PIN_RESULTS)
    __device__ __forceinline__ static void pin(float& v) { asm volatile("" : "+v"(v)); }
    __device__ __forceinline__ static void pin_out(float& v) { asm volatile("" : "+v"(v)); }
#else
    __device__ __forceinline__ static void pin(float&) {}
    __device__ __forceinline__ static void pin_out(float&) {}
#endif
    __device__ __forceinline__ static const float* opaque(const float* w) {
        int zo = 0;
        asm volatile("" : "+v"(zo));
        return w + zo;
    }

    template <int HP>
    __device__ __forceinline__ static void eval_range(const float* __restrict__ w_, int jb, bool bias,
                                                      const float (&x)[DIN], float (&out)[DOUT]) {
        const float* w = opaque(w_);
        f2 h[H / 2];
        hidden1(w, x, h);
        f2 acc[DOUT];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) acc[o] = f2{bias ? w[kB2 + o] : 0.f, 0.f};
        const f2 zero = f2{0.f, 0.f};
#pragma unroll 1
        for (int jj = 0; jj < HP; jj += 4) {
            const int j0 = jb + jj;
            f2 pa, pb;
            pre2_group4(w, j0, h, pa, pb);
            pa = pk_max(pa, zero);
            pb = pk_max(pb, zero);
#pragma unroll
            for (int o = 0; o < DOUT; ++o) {
                const float4 wo = *reinterpret_cast<const float4*>(w + kW2 + o * H + j0);
                acc[o] = pk_fma(pa, f2{wo.x, wo.y}, acc[o]);
                acc[o] = pk_fma(pb, f2{wo.z, wo.w}, acc[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            out[o] = acc[o].x + acc[o].y;
            pin_out(out[o]);
        }
    }

    template <bool ROLLED = false>
    __device__ __forceinline__ static void eval(const float* __restrict__ w, const float (&x)[DIN],
                                                float (&out)[DOUT]) {
        eval_range<H>(w, 0, true, x, out);
    }

    template <int S>
    __device__ __forceinline__ static void eval_part(const float* __restrict__ w, int part, const float (&x)[DIN],
                                                     float (&out)[DOUT]) {
        constexpr int HP = H / S;
        static_assert(HP % 4 == 0, "hidden slice must be a multiple of 4");
        eval_range<HP>(w, part * HP, part == 0, x, out);
    }

    // dx += (d out / d x)^T dout through the second-layer units [jb, jb + HP); everything is recomputed
    template <int HP>
    __device__ __forceinline__ static void bwd_range(const float* __restrict__ w_, int jb, const float (&x)[DIN],
                                                     const float (&dout)[DOUT], float (&dx)[DIN]) {
        const float* w = opaque(w_);
        f2 h[H / 2];
        float dh[H];
        hidden1(w, x, h);
#pragma unroll
        for (int i = 0; i < H; ++i) dh[i] = 0.f;
#pragma unroll 1
        for (int jj = 0; jj < HP; jj += 4) {
            const int j0 = jb + jj;
            f2 pa, pb;
            pre2_group4(w, j0, h, pa, pb);
            f2 da = f2{0.f, 0.f}, db = da;
#pragma unroll
            for (int o = 0; o < DOUT; ++o) {
                const float4 wo = *reinterpret_cast<const float4*>(w + kW2 + o * H + j0);
                const f2 go = f2{dout[o], dout[o]};
                da = pk_fma(go, f2{wo.x, wo.y}, da);
                db = pk_fma(go, f2{wo.z, wo.w}, db);
            }
            const float d0 = pa.x > 0.f ? da.x : 0.f, d1 = pa.y > 0.f ? da.y : 0.f;
            const float d2 = pb.x > 0.f ? db.x : 0.f, d3 = pb.y > 0.f ? db.y : 0.f;
            // (a second read of the four columns of Wh: through the pointer pre2_group4 used, hipcc keeps all 4 H values of its
            //  reads alive -- in AGPRs at one wave per SIMD, in scratch at two -- instead of re-reading them)
            const float* wh = opaque(w_) + kWh + j0;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const float4 wi = *reinterpret_cast<const float4*>(wh + i * H);
                dh[i] = fmaf(wi.x, d0, fmaf(wi.y, d1, fmaf(wi.z, d2, fmaf(wi.w, d3, dh[i]))));
                if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        f2 dxa[DIN];
#pragma unroll
        for (int i = 0; i < DIN; ++i) dxa[i] = f2{dx[i], 0.f};
        // (W1 is read a second time here: through the same pointer hipcc keeps the DIN * H values of hidden1() in registers
        //  -- or scratch -- across the loop above instead of re-reading them)
        const float* wt = opaque(w_);
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const f2 ga = f2{h[k / 2].x > 0.f ? dh[k] : 0.f, h[k / 2].y > 0.f ? dh[k + 1] : 0.f};
            const f2 gb = f2{h[k / 2 + 1].x > 0.f ? dh[k + 2] : 0.f, h[k / 2 + 1].y > 0.f ? dh[k + 3] : 0.f};
#pragma unroll
            for (int i = 0; i < DIN; ++i) {
                const float4 wi = *reinterpret_cast<const float4*>(wt + kW1 + i * H + k);
                dxa[i] = pk_fma(ga, f2{wi.x, wi.y}, dxa[i]);
                dxa[i] = pk_fma(gb, f2{wi.z, wi.w}, dxa[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < DIN; ++i) {
            dx[i] = dxa[i].x + dxa[i].y;
            pin(dx[i]);
        }
    }

    template <int S>
    __device__ __forceinline__ static void bwd_input_part(const float* __restrict__ w, int part, const float (&x)[DIN],
                                                          const float (&dout)[DOUT], float (&dx)[DIN]) {
        constexpr int HP = H / S;
        static_assert(HP % 4 == 0, "hidden slice must be a multiple of 4");
        bwd_range<HP>(w, part * HP, x, dout, dx);
    }

    template <bool ROLLED = false>
    __device__ __forceinline__ static void bwd_input(const float* __restrict__ w, const float (&x)[DIN],
                                                     const float (&dout)[DOUT], float (&dx)[DIN]) {
        bwd_range<H>(w, 0, x, dout, dx);
    }
};
