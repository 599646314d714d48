"""SVO -- mirror of reference src/SMC/SVO.py:6-409.

The forward particle filter (SVO.SMC, :60-180, with sample_from_2_dist :182-232 and
resample_X / get_resample_idx :243-300) runs as ONE persistent HIP kernel
(psvo_filter_forward, psvo_amd/csrc/filter_fwd.hip).  This class only prepares what does not
depend on the particles -- encoder features and the hoisted q0 / q2 means, B*T rows -- and
permutes the result to the reference's (batch, time, particles, Dx).
"""
import torch

from .. import ops
from ..autograd import FilterCovFunction, FilterFunction


class SVO:
    def __init__(self, model, FLAGS, name="log_ZSMC"):
        self.model = model

        # SSM distributions
        self.q0 = model.q0_dist
        self.q1 = model.q1_dist
        self.q2 = model.q2_dist
        self.f = model.f_dist
        self.g = model.g_dist

        self.n_particles = FLAGS.n_particles
        self.q_uses_true_X = FLAGS.q_uses_true_X

        # bidirectional RNN as full sequence observations encoder
        self.X0_use_separate_RNN = FLAGS.X0_use_separate_RNN
        self.use_stack_rnn = FLAGS.use_stack_rnn

        self.smooth_obs = True
        self.resample_particles = True

        self.name = name
        self.generator = None        # torch.Generator on the compute device; None = global RNG

    # ------------------------------------------------------------------------------------------
    def get_log_ZSMC(self, obs, hidden, noise=None):
        """obs (batch, time, Dy), hidden (batch, time, Dx) -> (log_ZSMC scalar, {"Xs": (B,T,N,Dx)}).

        `noise` optionally injects the random draws in the HBM layout of include/psvo_hip.h:
        eps_f (T,B,Dx,N), u_f (T,B,N) or idx_f (T,B,N) int32.
        """
        batch_size, time, _ = obs.shape
        self.Dx, self.batch_size, self.time = self.model.Dx, batch_size, time

        log = {}
        self._release()
        self._sigmas = self.model.sigmas()          # every scale vector of this evaluation, one fused launch
        filt = self.SMC(hidden, obs, noise=noise)
        log_ZSMC = self.compute_log_ZSMC(filt["lse"])
        # (T, B, Dx, N) -> (batch_size, time, n_particles, Dx)
        log["Xs"] = filt["Xanc"].permute(1, 0, 3, 2)
        log["filter"] = filt
        self._release()
        return log_ZSMC, log

    def _release(self):
        """drop what one evaluation cached on the object (scale vectors, hoisted means, features, stream context).

        These tensors carry autograd history; kept until the next call they keep the previous evaluation's graph alive, and
        with it the parameters' gradient accumulators, which remember the stream they were created on.  The next evaluation
        would reuse them: issued under a hipGraph capture on another stream the engine then synchronises that old stream
        into the capture, where it is never joined back (hipStreamEndCapture crashes on ROCm 7.2 instead of reporting it)."""
        self._sigmas = self._m0 = self._sig0 = self._obs_TB = self._ov = None
        self.preprocessed_X0 = self.preprocessed_obs = None

    # hidden widths the kernels are instantiated for (template parameter H of csrc/*.hip)
    _KERNEL_H = (16, 32, 64)

    def _particle_mlps(self):
        """the MLPs evaluated per particle INSIDE the kernels (one or two hidden layers each; hip_params() raises otherwise)"""
        m = self.model
        trans = [m.q1_tran, m.g_tran]
        if not m.use_bootstrap:
            trans.append(m.f_tran)
        if getattr(m, "q1_inv_tran", None) is not None and hasattr(self, "q1_inv"):
            trans.append(m.q1_inv_tran)
        return trans

    def _kernel_width(self):
        """(H the kernels run at, whether any per-particle MLP layer is narrower than that, hidden layers per MLP).

        The kernels take ONE hidden width and ONE depth (1 or 2 hidden layers: psvo_desc.layers) for all per-particle MLPs
        of a launch.  Other widths the flags can reach (q1_layers=50, g_layers=16 beside q1_layers=32, "64,32", ...) run at
        the next instantiated width with zero-padded hidden units: a padded unit is relu(x . 0 + 0) = 0 times a zero row of
        the next kernel, so values and the gradients of the real entries are unchanged (the padding is
        `torch.nn.functional.pad` on the parameters: autograd slices the padded gradient back).  Wider than 64 has no
        kernel, and MLPs of different depth cannot be padded to a common one (an extra relu layer is not the identity):
        ValueError."""
        cached = self.__dict__.get("_kw")
        if cached is None:
            params = [t.hip_params() for t in self._particle_mlps()]
            depths = sorted(set(1 if len(p) == 4 else 2 for p in params))
            if len(depths) != 1:
                raise ValueError("per-particle MLPs (q1 / f / g / q1_inv) with different numbers of hidden layers %s: the "
                                 "kernels take one depth per launch; no fallback path exists"
                                 % [t.Dhs for t in self._particle_mlps()])
            layers = depths[0]
            widths = [w for p in params for w in ((p[0].shape[1],) if layers == 1 else (p[0].shape[1], p[4].shape[1]))]
            avail = self._KERNEL_H if layers == 1 else self._KERNEL_H[1:]      # (two layers: 32 and 64)
            fit = [H for H in avail if H >= max(widths)]
            if not fit:
                raise ValueError("per-particle MLP hidden width %d exceeds the widest kernel instantiation (%d); "
                                 "no fallback path exists" % (max(widths), self._KERNEL_H[-1]))
            cached = self._kw = (fit[0], any(w != fit[0] for w in widths), layers)
        return cached

    def _mlp_params(self, tran):
        """(W1, b1, W2, b2[, Wh, bh]) of a per-particle MLP at the kernels' hidden width"""
        p = tran.hip_params()
        H = self._kernel_width()[0]
        F = torch.nn.functional
        W1, b1, W2, b2 = p[:4]
        if len(p) == 4:
            pad = H - W1.shape[1]
            if pad:
                W1, b1, W2 = F.pad(W1, (0, pad)), F.pad(b1, (0, pad)), F.pad(W2, (0, 0, 0, pad))
            return W1, b1, W2, b2
        Wh, bh = p[4:]
        p1, p2 = H - W1.shape[1], H - Wh.shape[1]
        if p1 or p2:
            W1, b1 = F.pad(W1, (0, p1)), F.pad(b1, (0, p1))
            Wh, bh = F.pad(Wh, (0, p2, 0, p1)), F.pad(bh, (0, p2))
            W2 = F.pad(W2, (0, 0, 0, p2))
        return W1, b1, W2, b2, Wh, bh

    @staticmethod
    def _mlp_args(*mlps):
        """positional layout of the autograd nodes: four tensors per MLP (None for an absent one), then -- two hidden
        layers -- (Wh, bh) per MLP behind everything else.  Returns (first, extra)."""
        first, extra = [], []
        two = any(p is not None and len(p) == 6 for p in mlps)
        for p in mlps:
            first += list(p[:4]) if p is not None else [None] * 4
            if two:
                extra += list(p[4:6]) if p is not None else [None] * 2
        return first, extra

    def _desc(self, M=1):
        d = self._make_desc(M, self._kernel_width()[0])
        d._ov = getattr(self, "_ov", None)          # stream-overlap context of this evaluation (PSVO only)
        return d

    def _sigma(self, dist):
        sig = getattr(self, "_sigmas", None)
        if sig is None or id(dist) not in sig:      # (tf_poisson has no scale vector: constant ones)
            return dist.get_sigma()
        return sig[id(dist)]

    def _gbuf(self, tran):
        """slice of the flat gradient buffer a native backward may accumulate this MLP's gradient into (None when the
        kernels run on padded copies of the weights: the padded gradient then goes back through autograd)"""
        if tran is None or self._kernel_width()[1]:
            return None
        return tran.__dict__.get("_flat_grad")

    def _make_desc(self, M, H):
        return ops.make_desc(self.batch_size, self.time, self.n_particles, M, self.model.Dx, self.model.Dy, H,
                             resample=self.resample_particles, two_q=self.model.use_2_q,
                             bootstrap=self.model.use_bootstrap,
                             emission=int(getattr(self.model, "poisson_emission", False)),
                             layers=self._kernel_width()[2])

    def _randn(self, *shape, device):
        return torch.randn(*shape, device=device, dtype=torch.float32, generator=self.generator)

    def _rand(self, *shape, device):
        return torch.rand(*shape, device=device, dtype=torch.float32, generator=self.generator)

    def SMC(self, hidden, obs, q_cov=1.0, noise=None):
        """SVO.py:60-180.  Returns the kernel's output dict: X (pre-resampling, Xs_ta),
        Xanc (X_ancestors_ta), Fm, logW (log_Ws_ta), idx, lse -- all (T, B, ...)-major."""
        if self.q_uses_true_X:
            # debug proposal around the true latents (SVO.py:73-77,127-131); run_flag forces it off
            # whenever use_2_q is set (runner.py:38-39)
            raise NotImplementedError("q_uses_true_X is a debugging aid outside the MI355X hot-path scope")
        model = self.model
        if model.output_cov:
            return self._SMC_cov(obs, noise)
        Dx, T, N, B = self.Dx, self.time, self.n_particles, self.batch_size
        dev = obs.device
        noise = noise or {}

        preprocessed_X0, preprocessed_obs = self.preprocess_obs(obs)
        both = model.use_bootstrap and model.use_2_q
        obs_TB = obs.transpose(0, 1).contiguous().float()
        self._obs_TB = obs_TB                                                  # (reused by the backward simulation)
        if not self.smooth_obs and both and obs.dtype == torch.float32:
            # obs[:, 0] itself is the X0 feature (SVO.py:322-324, no X0_transformer in this wiring): row 0 of the
            # (T, B, Dy) copy the kernels need anyway is the same data, contiguous -- no second copy launch
            preprocessed_X0 = obs_TB[0]
        self.preprocessed_X0, self.preprocessed_obs = preprocessed_X0, preprocessed_obs

        # ---- hoisted, particle-independent terms (B*T rows) ------------------------------------
        m0 = self.q0.mean(preprocessed_X0)                                    # (B, Dx)
        sig0 = self._sigma(self.q0)
        if both:
            fm0, fsig0 = m0, sig0                                             # f_0 is q0's own density
        else:
            fm0, fsig0 = self.f.mean(preprocessed_X0), self._sigma(self.f)    # SVO.py:91-92
        mu2 = sig_q2 = None
        if model.use_2_q:
            if preprocessed_obs is obs:
                # raw observations as features (AESMC / IWAE / PSVO): rows in (T, B) order give mu2 in the kernels' layout
                # directly -- no transposed copy of the output, none of its gradient
                mu2 = self.q2.mean(obs_TB)                                     # (T, B, Dx)
            else:
                mu2 = self.q2.mean(preprocessed_obs).transpose(0, 1).contiguous()
            sig_q2 = self._sigma(self.q2)

        eps = noise.get("eps_f")
        if eps is None:
            eps = self._randn(T, B, Dx, N, device=dev)
        u, idx_in = noise.get("u_f"), noise.get("idx_f")
        if self.resample_particles and u is None and idx_in is None:
            u = self._rand(T, B, N, device=dev)

        first, extra = self._mlp_args(self._mlp_params(model.q1_tran),
                                      None if model.use_bootstrap else self._mlp_params(model.f_tran),
                                      self._mlp_params(model.g_tran))
        sig_f = None if model.use_bootstrap else self._sigma(self.f)
        self._m0, self._sig0 = m0, sig0
        desc = self._desc()
        gb = (self._gbuf(model.q1_tran), None if model.use_bootstrap else self._gbuf(model.f_tran), self._gbuf(model.g_tran))
        desc._gbufs = gb if (gb[0] is not None and gb[2] is not None and (model.use_bootstrap or gb[1] is not None)) else None
        # one opaque autograd node: psvo_filter_forward / psvo_filter_backward
        lse, Fm, logW, X, Xanc, idx = FilterFunction.apply(
            desc, obs_TB, eps, u, idx_in,
            *first, self._sigma(self.q1), sig_q2, sig_f, self._sigma(self.g), mu2, m0, sig0, fm0, fsig0, *extra)
        return {"lse": lse, "Fm": Fm, "logW": logW, "X": X, "Xanc": Xanc, "idx": idx, "eps": eps, "u": u}

    def _mlp_params_cov(self, tran):
        """(W1, b1, W_mu, b_mu, W_sigma, b_sigma) of a per-particle MLP with the covariance head, at the kernels' width"""
        W1, b1, Wm, bm, Ws, bs = tran.hip_params_cov()
        pad = self._kernel_width()[0] - W1.shape[1]
        if pad:
            F = torch.nn.functional
            W1, b1 = F.pad(W1, (0, pad)), F.pad(b1, (0, pad))
            Wm, Ws = F.pad(Wm, (0, 0, 0, pad)), F.pad(Ws, (0, 0, 0, pad))
        return W1, b1, Wm, bm, Ws, bs

    def _SMC_cov(self, obs, noise):
        """SVO.SMC with state-dependent diagonal scales (FLAGS.output_cov and FLAGS.diag_cov; src/transformation/MLP.py:40-46,
        58-61, src/distribution/mvn.py:66-71): psvo_filter_forward_cov / psvo_filter_backward_cov.  The hoisted distributions
        (q0, q2 and, outside the default wiring, f on the X0 feature) hand over mean AND scale per row; the per-particle
        MLPs carry their second head into the kernel."""
        model = self.model
        Dx, T, N, B = self.Dx, self.time, self.n_particles, self.batch_size
        dev = obs.device
        noise = noise or {}
        if self._kernel_width()[2] != 1:
            raise ValueError("output_cov with two hidden layers per particle MLP: the state-dependent-scale kernels take one "
                             "hidden layer (psvo_filter_forward_cov); no fallback path exists")

        preprocessed_X0, preprocessed_obs = self.preprocess_obs(obs)
        both = model.use_bootstrap and model.use_2_q
        obs_TB = obs.transpose(0, 1).contiguous().float()
        self._obs_TB = obs_TB
        self.preprocessed_X0, self.preprocessed_obs = preprocessed_X0, preprocessed_obs

        m0, sig0 = self.q0.mean_and_sigma(preprocessed_X0, self._sigma(self.q0))            # (B, Dx) each
        if both:
            fm0, fsig0 = m0, sig0
        else:
            fm0, fsig0 = self.f.mean_and_sigma(preprocessed_X0, self._sigma(self.f))        # SVO.py:91-92
        mu2 = sig2 = None
        if model.use_2_q:
            if preprocessed_obs is obs:
                mu2, sig2 = self.q2.mean_and_sigma(obs_TB, self._sigma(self.q2))            # (T, B, Dx)
            else:
                mu2, sig2 = (v.transpose(0, 1).contiguous()
                             for v in self.q2.mean_and_sigma(preprocessed_obs, self._sigma(self.q2)))

        eps = noise.get("eps_f")
        if eps is None:
            eps = self._randn(T, B, Dx, N, device=dev)
        u, idx_in = noise.get("u_f"), noise.get("idx_f")
        if self.resample_particles and u is None and idx_in is None:
            u = self._rand(T, B, N, device=dev)

        q1p = self._mlp_params_cov(model.q1_tran)
        fp = (None,) * 6 if model.use_bootstrap else self._mlp_params_cov(model.f_tran)
        gp = self._mlp_params_cov(model.g_tran)
        self._m0, self._sig0 = m0, sig0
        lse, Fm, Fs, logW, X, Xanc, idx = FilterCovFunction.apply(
            self._desc(), obs_TB, eps, u, idx_in, *q1p, *fp, *gp,
            self._sigma(self.q1), None if model.use_bootstrap else self._sigma(self.f), self._sigma(self.g),
            mu2, sig2, m0, sig0.expand(B, Dx), fm0, fsig0.expand(B, Dx))
        return {"lse": lse, "Fm": Fm, "Fs": Fs, "logW": logW, "X": X, "Xanc": Xanc, "idx": idx, "eps": eps, "u": u}

    def compute_log_ZSMC(self, lse):
        """SVO.py:302-311: mean_b sum_t logsumexp_n log_Ws[t, n, b]; `lse` (T, B) is the per-step
        logsumexp the filter kernel already produced."""
        if lse.requires_grad:
            return lse.sum(0).mean()
        return ops.elbo_filter(self._desc(), lse).mean()

    def preprocess_obs(self, obs):
        """SVO.py:313-331 -> (preprocessed_X0 (B, E0), preprocessed_obs (B, T, E))."""
        if not self.smooth_obs:
            preprocessed_obs = obs
            preprocessed_X0 = obs[:, 0]
        else:
            preprocessed_X0, preprocessed_obs = self.preprocess_obs_w_bRNN(obs)
        if not (self.model.use_bootstrap and self.model.use_2_q):
            preprocessed_X0 = self.model.X0_transformer(preprocessed_X0)
        return preprocessed_X0, preprocessed_obs

    def preprocess_obs_w_bRNN(self, obs, need_X0=True):
        """SVO.py:333-369 (use_stack_rnn): bi-LSTM encodings (B, T, 2Dh) and the X0 feature
        concat(out[:, -1], out[:, 0]) (B, 4Dh).  `need_X0=False` skips the separate X0 encoder
        when the caller discards its output (PSVO.BS_preprocess_obs, reference PSVO.py:80)."""
        y_smoother, X0_smoother = self.model.bRNN
        outputs = y_smoother(obs)
        preprocessed_obs = outputs
        if not need_X0:
            return None, preprocessed_obs
        if self.X0_use_separate_RNN:
            outputs = X0_smoother(obs)
        if self.use_stack_rnn:
            preprocessed_X0 = torch.cat([outputs[:, -1], outputs[:, 0]], dim=-1)            # (B, 4Dh), SVO.py:360-367
        else:
            Dh = outputs.shape[-1] // 2                                                     # outputs = [fw | bw]
            preprocessed_X0 = torch.cat([outputs[:, -1, :Dh], outputs[:, 0, Dh:]], dim=-1)  # (B, 2Dh)
        return preprocessed_X0, preprocessed_obs

    def n_step_prediction(self, n_steps, hidden, obs):
        """SVO.py:371-404: k-step predictions from the particle mean."""
        batch_size, time, _, _ = hidden.shape
        assert n_steps < time, "n_steps = {} >= time".format(n_steps)
        x_BxTmkxDz = hidden.mean(dim=2)
        y_hat_N_BxTxDy = []
        for k in range(n_steps):
            y_hat_N_BxTxDy.append(self.g.mean(x_BxTmkxDz))
            x_BxTmkxDz = self.f.mean(x_BxTmkxDz[:, :-1])
        y_hat_N_BxTxDy.append(self.g.mean(x_BxTmkxDz))
        y_N_BxTxDy = [obs[:, k:, :] for k in range(n_steps + 1)]
        return y_hat_N_BxTxDy, y_N_BxTxDy

    def get_nextX(self, X):
        """SVO.py:406-409 (quiver plots)."""
        return self.f.mean(X)


def _c(x):
    """detach + make contiguous fp32 (tensors or tuples of tensors); None passes through."""
    if x is None:
        return None
    if isinstance(x, (tuple, list)):
        return tuple(_c(v) for v in x)
    return x.detach().float().contiguous()
