"""PSVO -- mirror of reference src/SMC/PSVO.py:8-216: forward filter, then backward simulation
with proposal.  The backward simulation (PSVO.py:69-203, the (M, N, N, B) transition tile) is
ONE persistent HIP kernel (psvo_bsim_forward, psvo_amd/csrc/bsim_fwd.hip)."""
import math

import torch

from .. import ops
from .. import autograd
from ..autograd import BsimCovFunction, BsimFunction, ElboBsimFunction, Overlap, side_stream
from .SVO import SVO


class PSVO(SVO):
    def __init__(self, model, FLAGS, name="log_ZSMC"):
        SVO.__init__(self, model, FLAGS, name="log_ZSMC")

        self.n_particles_for_BSim_proposal = FLAGS.n_particles_for_BSim_proposal

        self.smooth_obs = False
        self.BSim_use_single_RNN = FLAGS.BSim_use_single_RNN

        self.q1_inv = model.q1_inv_dist
        self.BSim_q_init = model.Bsim_q_init_dist
        self.BSim_q2 = model.BSim_q2_dist

    def get_log_ZSMC(self, obs, hidden, noise=None):
        """PSVO.py:21-50.  Extra `noise` keys: eps_b (T,B,Dx,N,M), u_b (T,B,N) or sel_b (T,B,N) int32."""
        batch_size, time, _ = obs.shape
        self.Dx, self.batch_size, self.time = self.model.Dx, batch_size, time
        if self.model.output_cov:
            return self._get_log_ZSMC_cov(obs, noise)

        log = {}
        # the filter (one workgroup per sequence) is issued on a side stream and overlaps with the encoder,
        # the backward-proposal means and the noise draws; events order it against the backward simulation
        # (second side stream for the bsim weight gradients only in the default wiring: otherwise the hoisted
        #  f.mean(mu_0) of the t = 0 term accumulates into the same gradient slice on the main stream)
        both = self.model.use_bootstrap and self.model.use_2_q
        self._release()
        self._ov = (Overlap(side_stream(obs.device), side_stream(obs.device, 1) if both else None)
                    if (autograd.OVERLAP and obs.is_cuda) else None)
        self._sigmas = self.model.sigmas()          # every scale vector of this evaluation, one fused launch
        filt = self.SMC(hidden, obs, noise=noise)                      # pre-resampling X and log_Ws
        bs = self.backward_simulation_w_proposal(filt, obs, noise=noise)
        self._ov = None
        log_ZSMC = self.compute_log_ZSMC_bsim(bs["score"])
        log["Xs"] = bs["bwX"].permute(1, 0, 3, 2)                      # (B, T, N, Dx)
        log["filter"], log["bsim"] = filt, bs
        self._release()
        return log_ZSMC, log

    def _get_log_ZSMC_cov(self, obs, noise):
        """get_log_ZSMC with state-dependent diagonal scales (FLAGS.output_cov and FLAGS.diag_cov): psvo_filter_forward_cov,
        then psvo_bsim_forward_cov on the filter's means AND scales; one stream (no overlap wiring on this path)."""
        log = {}
        self._release()
        self._ov = None
        self._sigmas = self.model.sigmas()
        filt = self.SMC(None, obs, noise=noise)
        bs = self._backward_simulation_cov(filt, obs, noise or {})
        log_ZSMC = self.compute_log_ZSMC_bsim(bs["score"])
        log["Xs"] = bs["bwX"].permute(1, 0, 3, 2)
        log["filter"], log["bsim"] = filt, bs
        self._release()
        return log_ZSMC, log

    def _backward_simulation_cov(self, filt, obs, noise):
        model = self.model
        Dx, T, N, B = self.Dx, self.time, self.n_particles, self.batch_size
        M = self.n_particles_for_BSim_proposal
        dev = obs.device
        _, preprocessed_obs = self.BS_preprocess_obs(obs)                        # (B, T, 2Dh)
        bmu2, bsig2 = (v.transpose(0, 1).contiguous()
                       for v in self.BSim_q2.mean_and_sigma(preprocessed_obs, self._sigma(self.BSim_q2)))    # (T, B, Dx)
        minit, sinit = self.BSim_q_init.mean_and_sigma(preprocessed_obs[:, -1], self._sigma(self.BSim_q_init))   # (B, Dx)
        mu_0 = self.preprocessed_X0                                              # cached by SMC()
        if not (model.use_bootstrap and model.use_2_q):
            imean, isig = self.f.mean_and_sigma(mu_0, self._sigma(self.f))       # PSVO.py:171
        else:
            imean, isig = self._m0, self._sig0                                   # PSVO.py:173: q0's density at mu_0
        eps_b = noise.get("eps_b")
        if eps_b is None:
            eps_b = self._randn(T, B, Dx, N, M, device=dev)
        u_b, sel_in = noise.get("u_b"), noise.get("sel_b")
        if u_b is None and sel_in is None:
            u_b = self._rand(T, B, N, device=dev)
        obs_TB = getattr(self, "_obs_TB", None)
        if obs_TB is None or obs_TB.shape[:2] != (T, B):
            obs_TB = obs.transpose(0, 1).contiguous().float()
        score, bwX, flp, glp, Omega, sel = BsimCovFunction.apply(
            self._desc(M), obs_TB, eps_b, u_b, sel_in, filt["Fm"], filt["Fs"], filt["logW"], filt["lse"],
            *self._mlp_params_cov(model.f_tran), *self._mlp_params_cov(model.g_tran), *self._mlp_params_cov(model.q1_inv_tran),
            self._sigma(self.f), self._sigma(self.g), self._sigma(self.q1_inv),
            bmu2, bsig2, minit, sinit.expand(B, Dx), imean, isig.expand(B, Dx))
        return {"score": score, "bwX": bwX, "flp": flp, "glp": glp, "Omega": Omega, "sel": sel}

    def compute_log_ZSMC_bsim(self, score):
        """PSVO.py:52-67: mean_b [ logsumexp_n( sum_t(f+g) - sum_t Omega ) - log N ]."""
        return ElboBsimFunction.apply(self._desc(self.n_particles_for_BSim_proposal), score)

    def backward_simulation_w_proposal(self, filt, obs, noise=None):
        model = self.model
        Dx, T, N, B = self.Dx, self.time, self.n_particles, self.batch_size
        M = self.n_particles_for_BSim_proposal
        dev = obs.device
        noise = noise or {}

        _, preprocessed_obs = self.BS_preprocess_obs(obs)                        # (B, T, 2Dh)
        bmu2 = self.BSim_q2.mean(preprocessed_obs).transpose(0, 1).contiguous()   # (T, B, Dx)
        minit = self.BSim_q_init.mean(preprocessed_obs[:, -1])                   # (B, Dx)
        mu_0 = self.preprocessed_X0                                              # cached by SMC()
        if not (model.use_bootstrap and model.use_2_q):
            imean, isig = self.f.mean(mu_0), self._sigma(self.f)                 # PSVO.py:171
        else:
            # PSVO.py:173: q0.log_prob(mu_0, .) -- the very MLP_q0(mu_0) / sigma_q0 the filter used at t = 0
            imean, isig = self._m0, self._sig0

        eps_b = noise.get("eps_b")
        if eps_b is None:
            eps_b = self._randn(T, B, Dx, N, M, device=dev)
        u_b, sel_in = noise.get("u_b"), noise.get("sel_b")
        if u_b is None and sel_in is None:
            u_b = self._rand(T, B, N, device=dev)
        obs_TB = getattr(self, "_obs_TB", None)                                  # cached by SMC() for this batch
        if obs_TB is None or obs_TB.shape[:2] != (T, B):
            obs_TB = obs.transpose(0, 1).contiguous().float()

        # one opaque autograd node: psvo_bsim_forward / psvo_bsim_backward
        desc = self._desc(M)
        gb = (self._gbuf(model.f_tran), self._gbuf(model.g_tran), self._gbuf(model.q1_inv_tran))
        desc._gbufs = gb if all(v is not None for v in gb) else None
        first, extra = self._mlp_args(self._mlp_params(model.f_tran), self._mlp_params(model.g_tran),
                                      self._mlp_params(model.q1_inv_tran))
        score, bwX, flp, glp, Omega, sel = BsimFunction.apply(
            desc, obs_TB, eps_b, u_b, sel_in, filt["Fm"], filt["logW"], filt["lse"],
            *first, self._sigma(self.f), self._sigma(self.g), self._sigma(self.q1_inv), self._sigma(self.BSim_q2),
            bmu2, minit, self._sigma(self.BSim_q_init), imean, isig, *extra)
        return {"score": score, "bwX": bwX, "flp": flp, "glp": glp, "Omega": Omega, "sel": sel}

    def BS_preprocess_obs(self, obs):
        """PSVO.py:205-216."""
        if self.BSim_use_single_RNN:
            # static_rnn over the forward cells only (PSVO.py:208-212); the final state it also returns is never read
            return None, self.model.y_smoother(obs)
        # the X0 feature of this encoder pass is never read (PSVO.py:80): do not compute it
        return self.preprocess_obs_w_bRNN(obs, need_X0=False)
