"""PSVOwR -- mirror of reference src/SMC/PSVOwR.py:8-198: PSVO whose backward simulation resamples
ACROSS the chains after every step and accumulates a per-step ELBO.  The whole loop
(PSVOwR.py:65-198) is ONE persistent HIP kernel with one workgroup per sequence
(psvo_bsimwr_forward, psvo_amd/csrc/psvowr_fwd.hip); its reverse pass is psvo_bsimwr_backward."""
import math

import torch

from .. import autograd
from ..autograd import BsimWRFunction, Overlap, side_stream
from .PSVO import PSVO


class PSVOwR(PSVO):
    def __init__(self, model, FLAGS, name="log_ZSMC"):
        PSVO.__init__(self, model, FLAGS, name=name)

    def get_log_ZSMC(self, obs, hidden, noise=None):
        """PSVOwR.py:38-62.  Extra `noise` keys on top of PSVO's: u_r (T,B,N) uniforms of the cross-chain
        draw or anc_r (T,B,N) int32 teacher-forced ancestors."""
        batch_size, time, _ = obs.shape
        self.Dx, self.batch_size, self.time = self.model.Dx, batch_size, time
        if self.model.output_cov:
            return self._get_log_ZSMC_cov(obs, noise)

        log = {}
        # (second side stream for the bsim weight gradients only in the default wiring: otherwise the hoisted
        #  f.mean(mu_0) of the t = 0 term accumulates into the same gradient slice on the main stream)
        both = self.model.use_bootstrap and self.model.use_2_q
        self._release()
        self._ov = (Overlap(side_stream(obs.device), side_stream(obs.device, 1) if both else None)
                    if (autograd.OVERLAP and obs.is_cuda) else None)
        self._sigmas = self.model.sigmas()
        filt = self.SMC(hidden, obs, noise=noise)
        bs = self.backward_simulation_w_resampling(filt, obs, noise=noise)
        self._ov = None
        # PSVOwR.py:144-148, 184-185: log_ZSMC = sum_t [logsumexp_n bw_log_W_t - log N], averaged over the batch
        log_ZSMC = (bs["lseW"].sum(0) - time * math.log(float(self.n_particles))).mean()
        log["Xs"] = bs["bwXanc"].permute(1, 0, 3, 2)                   # (B, T, N, Dx), PSVOwR.py:198
        log["filter"], log["bsim"] = filt, bs
        self._release()
        return log_ZSMC, log

    def _get_log_ZSMC_cov(self, obs, noise):
        """get_log_ZSMC with state-dependent diagonal scales (FLAGS.output_cov and FLAGS.diag_cov): psvo_filter_forward_cov,
        then psvo_bsimwr_forward_cov (one launch per time step: the kernel boundary is the cross-chain exchange); one stream."""
        from ..autograd import BsimWRCovFunction
        model = self.model
        noise = noise or {}
        log = {}
        self._release()
        self._ov = None
        self._sigmas = model.sigmas()
        filt = self.SMC(None, obs, noise=noise)
        Dx, T, N, B = self.Dx, self.time, self.n_particles, self.batch_size
        M = self.n_particles_for_BSim_proposal
        dev = obs.device
        _, preprocessed_obs = self.BS_preprocess_obs(obs)
        bmu2, bsig2 = (v.transpose(0, 1).contiguous()
                       for v in self.BSim_q2.mean_and_sigma(preprocessed_obs, self._sigma(self.BSim_q2)))
        minit, sinit = self.BSim_q_init.mean_and_sigma(preprocessed_obs[:, -1], self._sigma(self.BSim_q_init))
        mu_0 = self.preprocessed_X0
        if not (model.use_bootstrap and model.use_2_q):
            imean, isig = self.f.mean_and_sigma(mu_0, self._sigma(self.f))       # PSVOwR.py:167
        else:
            imean, isig = self._m0, self._sig0                                   # PSVOwR.py:169
        eps_b = noise.get("eps_b")
        if eps_b is None:
            eps_b = self._randn(T, B, Dx, N, M, device=dev)
        u_b, sel_in = noise.get("u_b"), noise.get("sel_b")
        if u_b is None and sel_in is None:
            u_b = self._rand(T, B, N, device=dev)
        u_r, anc_in = noise.get("u_r"), noise.get("anc_r")
        if u_r is None and anc_in is None:
            u_r = self._rand(T, B, N, device=dev)
        obs_TB = getattr(self, "_obs_TB", None)
        if obs_TB is None or obs_TB.shape[:2] != (T, B):
            obs_TB = obs.transpose(0, 1).contiguous().float()
        lseW, bwXanc, bwX, bwW, sel, anc = BsimWRCovFunction.apply(
            self._desc(M), obs_TB, eps_b, u_b, u_r, sel_in, anc_in, filt["Fm"], filt["Fs"], filt["logW"], filt["lse"],
            *self._mlp_params_cov(model.f_tran), *self._mlp_params_cov(model.g_tran), *self._mlp_params_cov(model.q1_inv_tran),
            self._sigma(self.f), self._sigma(self.g), self._sigma(self.q1_inv),
            bmu2, bsig2, minit, sinit.expand(B, Dx), imean, isig.expand(B, Dx))
        bs = {"lseW": lseW, "bwXanc": bwXanc, "bwX": bwX, "bwW": bwW, "sel": sel, "anc": anc}
        log_ZSMC = (lseW.sum(0) - T * math.log(float(N))).mean()               # PSVOwR.py:144-148, 184-185
        log["Xs"] = bwXanc.permute(1, 0, 3, 2)
        log["filter"], log["bsim"] = filt, bs
        self._release()
        return log_ZSMC, log

    def backward_simulation_w_resampling(self, filt, obs, noise=None):
        model = self.model
        Dx, T, N, B = self.Dx, self.time, self.n_particles, self.batch_size
        M = self.n_particles_for_BSim_proposal
        dev = obs.device
        noise = noise or {}

        _, preprocessed_obs = self.BS_preprocess_obs(obs)
        bmu2 = self.BSim_q2.mean(preprocessed_obs).transpose(0, 1).contiguous()
        minit = self.BSim_q_init.mean(preprocessed_obs[:, -1])
        mu_0 = self.preprocessed_X0
        if not (model.use_bootstrap and model.use_2_q):
            imean, isig = self.f.mean(mu_0), self._sigma(self.f)                 # PSVOwR.py:167
        else:
            imean, isig = self._m0, self._sig0                                   # PSVOwR.py:169

        eps_b = noise.get("eps_b")
        if eps_b is None:
            eps_b = self._randn(T, B, Dx, N, M, device=dev)
        u_b, sel_in = noise.get("u_b"), noise.get("sel_b")
        if u_b is None and sel_in is None:
            u_b = self._rand(T, B, N, device=dev)
        u_r, anc_in = noise.get("u_r"), noise.get("anc_r")
        if u_r is None and anc_in is None:
            u_r = self._rand(T, B, N, device=dev)
        obs_TB = getattr(self, "_obs_TB", None)                                  # cached by SMC() for this batch
        if obs_TB is None or obs_TB.shape[:2] != (T, B):
            obs_TB = obs.transpose(0, 1).contiguous().float()

        desc = self._desc(M)
        gb = (self._gbuf(model.f_tran), self._gbuf(model.g_tran), self._gbuf(model.q1_inv_tran))
        desc._gbufs = gb if all(v is not None for v in gb) else None
        if getattr(self, "_sticky", None) is None or self._sticky.device != dev:
            self._sticky = torch.zeros(1, dtype=torch.int32, device=dev)    # exchange time-outs of ALL launches so far
        desc._sticky = self._sticky
        first, extra = self._mlp_args(self._mlp_params(model.f_tran), self._mlp_params(model.g_tran),
                                      self._mlp_params(model.q1_inv_tran))
        lseW, bwXanc, bwX, bwW, sel, anc, ws = BsimWRFunction.apply(
            desc, obs_TB, eps_b, u_b, u_r, sel_in, anc_in, filt["Fm"], filt["logW"], filt["lse"],
            *first, self._sigma(self.f), self._sigma(self.g), self._sigma(self.q1_inv), self._sigma(self.BSim_q2),
            bmu2, minit, self._sigma(self.BSim_q_init), imean, isig, *extra)
        # ws[-1] (as int32) is nonzero iff an exchange poll of the kernel timed out (check_exchange, the tests)
        self._last_ws = ws
        return {"lseW": lseW, "bwXanc": bwXanc, "bwX": bwX, "bwW": bwW, "sel": sel, "anc": anc, "ws": ws}

    def check_exchange(self):
        """Raise if the workgroups of a sequence lost each other in ANY evaluation or reverse pass since the last call (a
        bounded poll of psvo_bsimwr_forward / psvo_bsimwr_backward timed out and the kernel drained with garbage: the
        objective or its gradients were wrong).  Every launch ORs its flag into one sticky device word, so a time-out in
        an earlier mini-batch is not erased by the launches after it.  Synchronises: the trainer calls it once an epoch."""
        sticky = getattr(self, "_sticky", None)
        if sticky is None:
            return
        v = int(sticky.item())
        if v:
            sticky.zero_()
            which = " and ".join(n for b, n in ((1, "psvo_bsimwr_forward"), (2, "psvo_bsimwr_backward")) if v & b)
            raise RuntimeError("%s: an exchange poll between the workgroups of a sequence timed out" % which)
