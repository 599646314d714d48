// C-ABI entry points of the backward-simulation reverse pass; the kernels are instantiated per Dx in
// bsim_bwd_dx{2,3,4}.hip (one translation unit each so that hipcc compiles them in parallel).
#include "bsim_bwd_impl.h"

namespace psvo {
extern template int bb_dispatch_dy<2>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
extern template int bb_dispatch_dy<3>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
extern template int bb_dispatch_dy<4>(const BsimBwdArgs&, const BsimBwdOut&, int, int, int, hipStream_t);
}  // namespace psvo

extern "C" int psvo_bsim_blocks(int B, int N, int M, int H, int Dx) {
    int HS, NTB, cpb, nblk;
    psvo::bsim_geometry(B, N, M, H, Dx, HS, NTB, cpb, nblk);
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
    float* dimean_rows, float* dsig_f, float* dsig_g, float* dsig_q1inv, float* dsig_bq2, float* dsig_init,
    float* disig, float* sacc_part, void* stream) {
    using namespace psvo;
    if (!desc || !Fm || !logW || !lse || !f || !g || !q1_inv || !sig_f || !sig_g || !sig_q1inv || !sig_bq2 || !bmu2 ||
        !minit || !sig_init || !imean || !isig || !obs || !eps_b || !bwX || !sel || !lam2_all || !om_all ||
        !mu1_all || !dscore || !xt || !dFt || !dGt || !dmu1 || !dFm_part || !dlogW_part || !dbmu2_rows ||
        !dminit_rows || !dimean_rows || !dsig_f || !dsig_g || !dsig_q1inv || !dsig_bq2 || !dsig_init || !disig ||
        !sacc_part)
        return PSVO_ERR_INVALID;
    if (desc->B <= 0 || desc->T < 2 || desc->N <= 0 || desc->M <= 0) return PSVO_ERR_INVALID;
    if (desc->N > 1024 || desc->B > 65535) return PSVO_ERR_UNSUPPORTED;

    BsimBwdArgs a;
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
    BsimBwdOut o{dsig_f, dsig_g, dsig_q1inv, dsig_bq2, dsig_init, disig};
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (desc->Dx) {
        case 2: return bb_dispatch_dy<2>(a, o, desc->Dy, desc->H, desc->M, s);
        case 3: return bb_dispatch_dy<3>(a, o, desc->Dy, desc->H, desc->M, s);
        case 4: return bb_dispatch_dy<4>(a, o, desc->Dy, desc->H, desc->M, s);
        default: return PSVO_ERR_UNSUPPORTED;
    }
}
