// C-ABI entry points of the backward-simulation reverse pass; the kernels are instantiated per Dx in
// bsim_bwd_dx{2,3,4}.hip (one translation unit each so that hipcc compiles them in parallel).
#include "bsim_bwd2_impl.h"

namespace psvo {
extern template int bb2_dispatch_dy<2>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, int, hipStream_t);
extern template int bb2_dispatch_dy<3>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, int, hipStream_t);
extern template int bb2_dispatch_dy<4>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, int, hipStream_t);
extern template int bb_dispatch_dy<2>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
extern template int bb_dispatch_dy<3>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
extern template int bb_dispatch_dy<4>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
// two hidden layers per MLP (psvo_desc.layers == 2): the v1 kernel family compiled with PSVO_L = 2 (bsim_bwd_dx{2,3,4}_l2.hip)
namespace l2 {
template <int DX>
int bb_dispatch_dy(const BsimBwdArgs& a, const BsimBwdOut& o, int Dy, int H, int M, hipStream_t s);
extern template int bb_dispatch_dy<2>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
extern template int bb_dispatch_dy<3>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
extern template int bb_dispatch_dy<4>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
}  // namespace l2
}  // namespace psvo

// Which reverse kernel psvo_bsim_backward launches (psvo_set_tuning(PSVO_TUNE_BSIM_BWD, v)):
//   0 = v1 (lane = (chain, half, m), per-j sums by butterfly),  1 = v2 VALU (bsim_bwd2_impl.h: j on lanes, per-j sums in
//   registers + swap-add),  2 = v2 with the per-j sums on v_mfma_f32_16x16x4_f32,  3 = v2 with the pair exponents on
//   v_mfma_f32_16x16x4_f32 (Dx = 2; other Dx run as 1),  4 = v2 with the pair exponents on v_mfma_f32_16x16x32_bf16, operands
//   split into three bf16 pieces (Dx = 2; other Dx run as 1),  -1 = the measured default.  The workspace geometry (psvo_bsim_blocks) follows the choice, so set it before sizing buffers.
static int g_bsim_bwd_variant = -1;

static int bsim_bwd_variant(int B, int T, int N, int M, int Dx, int Dy, int layers = 1) {
    if (layers == 2) return 0;    // (v2's register budget has no room for a second hidden layer: v1 only)
    // measured default (profiles/r02_bsim_bwd_ab.md): v2 wins where its working set fits the 256 VGPRs
    // of two waves per SIMD -- Dx = 2 (C* -8 %, C4 -29 %) and, with the scalar accumulators in LDS, Dx = 3 (C3 -12 %); at
    // Dx = 4 it still spills 65 registers to scratch and v1 (one wave per SIMD, 512 VGPRs) is faster
    int v = g_bsim_bwd_variant < 0 ? (Dx <= 3 ? 1 : 0) : g_bsim_bwd_variant;
    if (v != 0 && !psvo::bsim2_supported(B, T, N, M, Dx, Dy)) v = 0;
    return v;
}

namespace psvo {
int set_rows_bwd_rb(int v);     // rows_mlp.hip
int get_rows_bwd_rb();
}  // namespace psvo

namespace psvo { int g_tune_l2_split = 0; int g_tune_skew_pct = 0; int g_tune_filter_bwd_scan = 1; }

namespace psvo { extern int g_tune_wgrad2; }

extern "C" int psvo_set_tuning(int key, int value) {
    if (key == PSVO_TUNE_ROWS_BWD) return psvo::set_rows_bwd_rb(value);
    if (key == PSVO_TUNE_FILTER_BWD && value >= 0 && value <= 2) {
        psvo::g_tune_filter_bwd_scan = value;
        return PSVO_OK;
    }
    if (key == PSVO_TUNE_WGRAD2 && (value == 0 || value == 2 || value == 3)) {
        psvo::g_tune_wgrad2 = value;
        return PSVO_OK;
    }
    if (key == PSVO_TUNE_SKEW && value >= 0 && value <= 1000) {
        psvo::g_tune_skew_pct = value;
        return PSVO_OK;
    }
    if (key == PSVO_TUNE_L2_SPLIT && (value == 0 || value == 1)) {
        psvo::g_tune_l2_split = value;
        return PSVO_OK;
    }
    if (key == PSVO_TUNE_BSIM_BWD && value >= -1 && value <= 4) {
        g_bsim_bwd_variant = value;
        return PSVO_OK;
    }
    return PSVO_ERR_INVALID;
}

extern "C" int psvo_get_tuning(int key) {
    if (key == PSVO_TUNE_ROWS_BWD) return psvo::get_rows_bwd_rb();
    if (key == PSVO_TUNE_L2_SPLIT) return psvo::g_tune_l2_split;
    if (key == PSVO_TUNE_SKEW) return psvo::g_tune_skew_pct;
    if (key == PSVO_TUNE_WGRAD2) return psvo::g_tune_wgrad2;
    if (key == PSVO_TUNE_FILTER_BWD) return psvo::g_tune_filter_bwd_scan;
    return key == PSVO_TUNE_BSIM_BWD ? g_bsim_bwd_variant : PSVO_ERR_INVALID;
}

extern "C" int psvo_bsim_blocks(const psvo_desc* desc) {
    if (!desc) return PSVO_ERR_INVALID;
    if (!psvo::desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;
    int HS, NTB, cpb, nblk;
    if (bsim_bwd_variant(desc->B, desc->T, desc->N, desc->M, desc->Dx, desc->Dy, desc->layers) != 0)
        psvo::bsim2_geometry(desc->N, desc->M, cpb, nblk);
    else psvo::bsim_geometry(desc->B, desc->N, desc->M, desc->H, desc->Dx, HS, NTB, cpb, nblk, desc->layers == 2 ? 2 : 1);
    return nblk;
}

extern "C" int psvo_bsim_acc_size(int Dx, int Dy) { return 7 * Dx + Dy; }

extern "C" int psvo_bsim_backward(
    const psvo_desc* desc, const float* Fm, const float* logW, const float* lse, const psvo_mlp* f,
    const psvo_mlp* g, const psvo_mlp* q1_inv, const float* sig_f, const float* sig_g, const float* sig_q1inv,
    const float* sig_bq2, const float* bmu2, const float* minit, const float* sig_init, const float* imean,
    const float* isig, const float* obs, const float* eps_b, const float* bwX, const int32_t* sel,
    const float* lam2_all, const float* om_all, const float* mu1_all, const float* dscore, float* xt, float* dFt,
    float* dGt, float* dmu1, float* dFm_part, float* dlogW_part, float* dbmu2_rows, float* dminit_rows,
    float* dimean_rows, float* sacc_part, void* stream) {
    using namespace psvo;
    if (!desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;
    if (!desc || !Fm || !logW || !lse || !f || !g || !q1_inv || !sig_f || !sig_g || !sig_q1inv || !sig_bq2 || !bmu2 ||
        !minit || !sig_init || !imean || !isig || !obs || !eps_b || !bwX || !sel || !lam2_all || !om_all ||
        !mu1_all || !dscore || !xt || !dFt || !dGt || !dmu1 || !dFm_part || !dlogW_part || !dbmu2_rows ||
        !dminit_rows || !dimean_rows || !sacc_part)
        return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T < 2 || desc->N <= 0 || desc->M <= 0) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535 || !desc_layers_ok(desc)) return PSVO_ERR_UNSUPPORTED;

    BsimBwdArgs a;
    a.skew = 0;
    a.B = desc->B; a.T = desc->T; a.N = desc->N; a.emission = desc->emission;
    a.f = *f; a.g = *g; a.q1inv = *q1_inv;
    a.Fm = Fm; a.logW = logW; a.lse = lse;
    a.sig_f = sig_f; a.sig_g = sig_g; a.sig_q1inv = sig_q1inv; a.sig_bq2 = sig_bq2;
    a.bmu2 = bmu2; a.minit = minit; a.sig_init = sig_init; a.imean = imean; a.isig = isig;
    a.obs = obs; a.eps_b = eps_b; a.bwX = bwX; a.sel = sel;
    a.lam2_all = lam2_all; a.om_all = om_all; a.mu1_all = mu1_all; a.dscore = dscore;
    a.xt = xt; a.dFt = dFt; a.dGt = dGt; a.dmu1 = dmu1;
    a.dFm_part = dFm_part; a.dlogW_part = dlogW_part; a.dbmu2_rows = dbmu2_rows; a.dminit_rows = dminit_rows;
    a.dimean_rows = dimean_rows; a.sacc_part = sacc_part;
    BsimBwdOut o{};      // (the folds are psvo_bsim_backward_fold's)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int variant = bsim_bwd_variant(desc->B, desc->T, desc->N, desc->M, desc->Dx, desc->Dy, desc->layers);
    if (variant != 0) {
        switch (desc->Dx) {
            case 2: return bb2_dispatch_dy<2>(a, o, desc->Dy, desc->H, desc->M, variant - 1, s);
            case 3: return bb2_dispatch_dy<3>(a, o, desc->Dy, desc->H, desc->M, variant - 1, s);
            case 4: return bb2_dispatch_dy<4>(a, o, desc->Dy, desc->H, desc->M, variant - 1, s);
            default: return PSVO_ERR_UNSUPPORTED;
        }
    }
    if (desc->layers == 2) {
        if (!f->Wh || !f->bh || !g->Wh || !g->bh || !q1_inv->Wh || !q1_inv->bh) return PSVO_ERR_INVALID;
        switch (desc->Dx) {
            case 2: return l2::bb_dispatch_dy<2>(a, o, desc->Dy, desc->H, desc->M, s);
            case 3: return l2::bb_dispatch_dy<3>(a, o, desc->Dy, desc->H, desc->M, s);
            case 4: return l2::bb_dispatch_dy<4>(a, o, desc->Dy, desc->H, desc->M, s);
            default: return PSVO_ERR_UNSUPPORTED;
        }
    }
    switch (desc->Dx) {
        case 2: return bb_dispatch_dy<2>(a, o, desc->Dy, desc->H, desc->M, s);
        case 3: return bb_dispatch_dy<3>(a, o, desc->Dy, desc->H, desc->M, s);
        case 4: return bb_dispatch_dy<4>(a, o, desc->Dy, desc->H, desc->M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}

namespace psvo {
template <int DX>
static int fold_dispatch_dy(const BsimBwdArgs& a, const BsimBwdOut& o, int Dy, int nblk, hipStream_t s) {
    clear_hip_error();
    switch (Dy) {
        case 1: launch_bsim_fold_finalize<DX, 1>(a, o, nblk, s); break;
        case 2: launch_bsim_fold_finalize<DX, 2>(a, o, nblk, s); break;
        default: return PSVO_ERR_UNSUPPORTED;
    }
    return launch_status();
}
}  // namespace psvo

extern "C" int psvo_bsim_backward_fold(const psvo_desc* desc, const float* dFm_part, const float* dlogW_part,
                                       const float* sacc_part, const float* sig_q1inv, const float* sig_bq2, float* dFm,
                                       float* dlogW, float* dsig_f, float* dsig_g, float* dsig_q1inv, float* dsig_bq2,
                                       float* dsig_init, float* disig, float* dlse, void* stream) {
    using namespace psvo;
    if (!desc || !dFm_part || !dlogW_part || !sacc_part || !sig_q1inv || !sig_bq2 || !dFm || !dlogW || !dsig_f ||
        !dsig_g || !dsig_q1inv || !dsig_bq2 || !dsig_init || !disig)
        return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T < 2 || desc->N <= 0 || desc->M <= 0) return PSVO_ERR_INVALID;
    BsimBwdArgs a{};
    a.B = desc->B; a.T = desc->T; a.N = desc->N;
    a.dFm_part = const_cast<float*>(dFm_part); a.dlogW_part = const_cast<float*>(dlogW_part);
    a.sacc_part = const_cast<float*>(sacc_part);
    a.sig_q1inv = sig_q1inv; a.sig_bq2 = sig_bq2;
    BsimBwdOut o{dsig_f, dsig_g, dsig_q1inv, dsig_bq2, dsig_init, disig, dFm, dlogW, dlse};
    const int nblk = psvo_bsim_blocks(desc);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return fold_dispatch_dy<2>(a, o, desc->Dy, nblk, s);
        case 3: return fold_dispatch_dy<3>(a, o, desc->Dy, nblk, s);
        case 4: return fold_dispatch_dy<4>(a, o, desc->Dy, nblk, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
