"""CPU oracle for the PSVO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PARITY UNPINNED: the reference (amoretti86/PSVO, TensorFlow 1.12 + TFP 0.5) ships no
tests, golden vectors or known-answer fixtures for this path, and TensorFlow is not
installable in the build container, so this restatement cannot be checked against the
reference's own outputs.  It is an op-for-op CPU restatement of the reference's
arithmetic with every random draw *injected* (normal eps, resampling uniforms u or
teacher-forced indices).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product path (psvo_amd/) never does.

Third-party arithmetic restated from its published closed forms (TF 1.12 / TFP 0.5,
pinned only by print statements at reference src/runner_flag.py:12-14):
  * tfd.MultivariateNormalDiag.log_prob(x) = -1/2 sum(((x-mu)/sigma)^2)
        - sum(log sigma) - D/2 log(2 pi);  sample = mu + sigma * eps
  * tf.reduce_logsumexp = max-shifted log-sum-exp
  * tfd.Categorical(logits).sample: inverse-CDF multinomial.  TF's RNG stream cannot
    be replayed, so the oracle DEFINES  idx = #{k : cumsum(exp(logit-max))_k <= u*total}
    (clamped to K-1) with u ~ U[0,1) injected.
  * keras Dense: y = x @ kernel + bias, kernel shape (in, out); relu hidden layers.
  * tf.contrib.rnn.LSTMBlockCell: gates (i, j, f, o) = split(concat(x, h) @ W + b);
    c' = c*sigmoid(f + 1) + sigmoid(i)*tanh(j);  h' = tanh(c')*sigmoid(o).

Layout is the reference's own internal layout: (T, N, B, D) particle-major
(reference src/SMC/SVO.py:176-178); the boundary output log["Xs"] is (B, T, N, Dx).

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
import math

import torch

LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- #
# L1: transformation / distribution
# --------------------------------------------------------------------------- #
def mlp_transform(p, x, with_cov=False):
    """MLP_transformation.transform -- src/transformation/MLP.py:48-68.

    p = {"layers": [(W, b), ...], "mu": (W, b)[, "sigma": (W, b)]}; W is the keras kernel (in, out).
    output_cov and diag_cov (p has a "sigma" head, MLP.py:40-46): the second return value is
    cov = exp(hidden @ W_sigma + b_sigma) + 1e-6 (MLP.py:58-61), else None.  (The full-covariance head,
    output_cov without diag_cov, is out of scope: SURVEY.md section 2 row 2.)
    """
    h = x
    for W, b in p["layers"]:
        h = torch.relu(h @ W + b)
    W, b = p["mu"]
    mu = h @ W + b
    if not with_cov:
        return mu
    cov = None
    if p.get("sigma") is not None:
        Ws, bs = p["sigma"]
        cov = torch.exp(h @ Ws + bs) + 1e-6
    return mu, cov


def get_sigma(p):
    """tf_mvn.get_sigma -- src/distribution/mvn.py:80-90."""
    s = torch.nn.functional.softplus(p["sigma_raw"])
    s = torch.where(torch.isnan(s), torch.zeros_like(s), s)
    return torch.maximum(s, torch.as_tensor(p["sigma_min"], dtype=s.dtype))


def diag_log_prob(x, mu, sigma):
    """tfd.MultivariateNormalDiag(mu, sigma).log_prob(x) (TFP 0.5, third-party)."""
    z = (x - mu) / sigma
    D = x.shape[-1]
    return -0.5 * (z * z).sum(-1) - torch.log(sigma).sum(-1) - 0.5 * D * LOG2PI


class OracleMVN:
    """tf_mvn, diagonal branches -- src/distribution/mvn.py:23-117.

    scale = sigma_con (state-independent, get_sigma) when the transformation has no covariance head, else
    sigma_con + 0.1 * cov(Input) with cov the MLP's exp head (mvn.py:66-71, output_cov and diag_cov)."""

    def __init__(self, p):
        self.p = p

    def mean(self, Input):                       # mvn.py:104-117
        return mlp_transform(self.p, Input)

    def mean_and_sigma(self, Input):             # get_mvn_from_transformation, mvn.py:51-78
        mu, cov = mlp_transform(self.p, Input, with_cov=True)
        s = get_sigma(self.p)
        if cov is not None:
            s = s + 0.1 * cov
        return mu, s

    def sigma(self, Input=None):
        if self.p.get("sigma") is None:
            return get_sigma(self.p)
        return self.mean_and_sigma(Input)[1]

    def log_prob(self, Input, output):           # mvn.py:99-102
        mu, s = self.mean_and_sigma(Input)
        return diag_log_prob(output, mu, s)

    def sample_and_log_prob(self, Input, eps):   # mvn.py:92-97; eps has the full sample shape
        mu, s = self.mean_and_sigma(Input)
        x = mu + s * eps
        return x, diag_log_prob(x, mu, s)


class OraclePoisson:
    """tf_poisson -- src/distribution/poisson.py:27-50.  The reference builds MultivariateNormalDiag(lambdas) with
    lambdas = softplus(MLP(Input)) + 1e-6 and no scale (:33-38): a unit-scale normal, emission only."""

    def __init__(self, p):
        self.p = p

    def mean(self, Input):                       # poisson.py:45-48
        return torch.nn.functional.softplus(mlp_transform(self.p, Input)) + 1e-6

    def sigma(self):
        return torch.ones_like(self.p["mu"][1])

    def log_prob(self, Input, output):           # poisson.py:40-43
        return diag_log_prob(output, self.mean(Input), self.sigma())


def logsumexp(x, dim, keepdim=False):
    return torch.logsumexp(x, dim=dim, keepdim=keepdim)


def multinomial_idx(log_W, u):
    """get_resample_idx -- src/SMC/SVO.py:266-300, with the draw defined by inverse CDF.

    log_W: (K, *batch); u: (*out) where out = (S, *batch) [sample_size=S] or batch
    [sample_size=()].  Returns int64 idx of shape u.shape with classes on axis 0 of log_W.
    """
    w = torch.exp(log_W - log_W.max(dim=0, keepdim=True).values)
    cdf = torch.cumsum(w, dim=0)                                  # (K, *batch)
    total = cdf[-1]
    target = u * total                                            # broadcast over leading S
    if u.dim() == log_W.dim():                                    # (S, *batch)
        cnt = (cdf.unsqueeze(0) <= target.unsqueeze(1)).sum(1)
    else:                                                         # batch only
        cnt = (cdf <= target.unsqueeze(0)).sum(0)
    return cnt.clamp(max=log_W.shape[0] - 1)


def draw_distance(log_W, u, idx):
    """How far each index `idx` is from being the inverse-CDF draw of `multinomial_idx(log_W, u)`: 0 where
    cdf[idx-1] <= u*total < cdf[idx], else the distance of u*total to that interval divided by total (so an index
    that differs only because u*total sits within rounding of a CDF edge has a distance of a few ulp).
    Shapes as in multinomial_idx; returns a tensor of idx's shape."""
    w = torch.exp(log_W - log_W.max(dim=0, keepdim=True).values)
    cdf = torch.cumsum(w, dim=0)
    total = cdf[-1]
    K = log_W.shape[0]
    target = u * total
    ix = idx if u.dim() == log_W.dim() else idx.unsqueeze(0)
    hi = torch.gather(cdf, 0, ix)
    lo = torch.where(ix > 0, torch.gather(cdf, 0, (ix - 1).clamp(min=0)), torch.zeros_like(hi))
    hi = torch.where(ix == K - 1, torch.full_like(hi, float("inf")), hi)
    if u.dim() != log_W.dim():
        hi, lo = hi[0], lo[0]
    d = torch.clamp(lo - target, min=0) + torch.clamp(target - hi, min=0)
    return d / total


def gather_particles(X, idx):
    """tf.gather_nd(X, resample_idx) for sample_size=N -- src/SMC/SVO.py:255-257,295-298.

    X: (K, B, ...); idx: (S, B) -> out[s, b] = X[idx[s, b], b].
    """
    B = X.shape[1]
    b = torch.arange(B).unsqueeze(0).expand_as(idx)
    return X[idx, b]


def gather_sub(X, idx):
    """gather_nd for sample_size=() over a leading M axis -- src/SMC/PSVO.py:99-102.

    X: (M, N, B, ...); idx: (N, B) -> out[n, b] = X[idx[n, b], n, b].
    """
    N, B = idx.shape
    n = torch.arange(N).unsqueeze(1).expand(N, B)
    b = torch.arange(B).unsqueeze(0).expand(N, B)
    return X[idx, n, b]


# --------------------------------------------------------------------------- #
# encoder (upstream of the particle path)
# --------------------------------------------------------------------------- #
def lstm_block_cell(x, h, c, W, b):
    """tf.contrib.rnn.LSTMBlockCell(forget_bias=1.0) -- used at src/model.py:164-176."""
    z = torch.cat([x, h], dim=-1) @ W + b
    i, j, f, o = z.chunk(4, dim=-1)
    c2 = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.tanh(c2) * torch.sigmoid(o)
    return h2, c2


def run_rnn(x_BTD, W, b, reverse=False):
    B, T, _ = x_BTD.shape
    Dh = W.shape[1] // 4
    h = x_BTD.new_zeros(B, Dh)
    c = x_BTD.new_zeros(B, Dh)
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h, c = lstm_block_cell(x_BTD[:, t], h, c, W, b)
        outs[t] = h
    return torch.stack(outs, dim=1)


def stack_bidirectional_rnn(x_BTD, layers):
    """tf.contrib.rnn.stack_bidirectional_dynamic_rnn -- src/SMC/SVO.py:337-341.

    layers: list of {"fw": (W, b), "bw": (W, b)}; each layer's input is the concat of
    both directions of the previous layer.
    """
    h = x_BTD
    for L in layers:
        fw = run_rnn(h, *L["fw"], reverse=False)
        bw = run_rnn(h, *L["bw"], reverse=True)
        h = torch.cat([fw, bw], dim=-1)
    return h


def bidirectional_rnn(x_BTD, layers):
    """tf.nn.bidirectional_dynamic_rnn over two MultiRNNCells (use_stack_rnn=False) -- src/SMC/SVO.py:342-346,
    src/model.py:168-170: the layers are stacked inside each direction, the directions only meet at the end.
    Returns (outputs_fw, outputs_bw), both (B, T, Dh_last)."""
    fw = bw = x_BTD
    for L in layers:
        fw = run_rnn(fw, *L["fw"], reverse=False)
        bw = run_rnn(bw, *L["bw"], reverse=True)
    return fw, bw


def static_rnn(x_BTD, layers):
    """tf.nn.static_rnn(MultiRNNCell(y_smoother_f), ...) -- src/SMC/PSVO.py:208-212 (BSim_use_single_RNN):
    the forward cells only, stacked; outputs (B, T, Dh_last)."""
    h = x_BTD
    for L in layers:
        h = run_rnn(h, *L["fw"], reverse=False)
    return h


# --------------------------------------------------------------------------- #
# L2: objectives
# --------------------------------------------------------------------------- #
class OracleSVO:
    """SVO -- src/SMC/SVO.py:6-409.  AESMC/IWAE/PSVO toggle smooth_obs / resample_particles."""

    def __init__(self, params, flags, smooth_obs=True, resample_particles=True):
        self.params = params
        self.q0 = OracleMVN(params["q0"])
        self.q1 = OracleMVN(params["q1"])
        self.q2 = OracleMVN(params["q2"]) if flags["use_2_q"] else None
        self.f = self.q1 if flags["use_bootstrap"] else OracleMVN(params["f"])   # src/model.py:145-151
        self.g = (OraclePoisson if flags.get("poisson_emission", False) else OracleMVN)(params["g"])   # model.py:153-160
        self.use_bootstrap = flags["use_bootstrap"]
        self.use_2_q = flags["use_2_q"]
        self.n_particles = flags["n_particles"]
        self.smooth_obs = smooth_obs
        self.resample_particles = resample_particles
        self.use_stack_rnn = flags.get("use_stack_rnn", True)

    # -- SVO.py:313-331
    def preprocess_obs(self, obs):
        if not self.smooth_obs:
            preprocessed_obs = list(obs.unbind(1))
            preprocessed_X0 = preprocessed_obs[0]
        else:
            preprocessed_X0, preprocessed_obs = self.preprocess_obs_w_bRNN(obs)
        if not (self.use_bootstrap and self.use_2_q):
            W, b = self.params["X0_transformer"]
            preprocessed_X0 = preprocessed_X0 @ W + b
        return preprocessed_X0, preprocessed_obs

    # -- SVO.py:333-369 (use_stack_rnn / X0_use_separate_RNN as flagged)
    def preprocess_obs_w_bRNN(self, obs):
        enc = self.params["bRNN"]
        if self.use_stack_rnn:
            outputs = stack_bidirectional_rnn(obs, enc["y_smoother"])
            preprocessed_obs = list(outputs.unbind(1))
            if enc.get("X0_smoother") is not None:
                outputs = stack_bidirectional_rnn(obs, enc["X0_smoother"])
            outputs_fw = outputs_bw = outputs                            # SVO.py:360-361
        else:
            outputs_fw, outputs_bw = bidirectional_rnn(obs, enc["y_smoother"])
            preprocessed_obs = list(torch.cat([outputs_fw, outputs_bw], dim=-1).unbind(1))
            if enc.get("X0_smoother") is not None:
                outputs_fw, outputs_bw = bidirectional_rnn(obs, enc["X0_smoother"])
        preprocessed_X0 = torch.cat([outputs_fw[:, -1], outputs_bw[:, 0]], dim=-1)
        return preprocessed_X0, preprocessed_obs

    # -- SVO.py:182-232, diagonal branch
    def sample_from_2_dist(self, dist1, dist2, d1_input, d2_input, eps):
        m1, s1 = dist1.mean_and_sigma(d1_input)          # (.mean(), .stddev() of the two MultivariateNormalDiag)
        m2, s2 = dist2.mean_and_sigma(d2_input)
        s1_inv, s2_inv = 1 / s1, 1 / s2
        combined_cov = 1 / (s1_inv + s2_inv)
        combined_mean = combined_cov * (s1_inv * m1 + s2_inv * m2)
        X = combined_mean + combined_cov * eps                       # mvn.sample(sample_size)
        q_lp = diag_log_prob(X, combined_mean, combined_cov)
        f_lp = diag_log_prob(X, m1, s1)
        return X, q_lp, f_lp

    # -- SVO.py:243-264
    def resample_X(self, X, log_W, u=None, idx=None, sample_size=(), resample_particles=True):
        if not resample_particles:
            return X, None
        if log_W.shape[0] == 1:
            # the reference asserts sample_size == 1 here (SVO.py:259); identity is the
            # only consistent definition for K == 1 and is what the build defines.
            idx0 = torch.zeros(log_W.shape[1:], dtype=torch.long)
            if sample_size == ():                 # one class, one draw per chain: drop the class axis
                return ([x[0] for x in X] if isinstance(X, list) else X[0]), idx0
            return X, idx0.unsqueeze(0)
        if idx is None:
            idx = multinomial_idx(log_W, u)
        if getattr(self, "draw_log", None) is not None:   # test hook: every categorical draw (logits, uniform, index taken)
            self.draw_log.append((log_W.detach(), u, idx))
        g = gather_particles if sample_size != () else gather_sub
        if isinstance(X, list):
            return [g(item, idx) for item in X], idx
        return g(X, idx), idx

    # -- SVO.py:60-180
    def SMC(self, obs, noise):
        """obs (B,T,Dy). noise: eps_f (T,N,B,Dx), u_f (T,N,B) and/or idx_f (T,N,B)."""
        B, T, _ = obs.shape
        N = self.n_particles
        q0, q1, f = self.q0, self.q1, self.f
        eps, u, idx_tf = noise["eps_f"], noise.get("u_f"), noise.get("idx_f")
        logN = math.log(float(N))

        preprocessed_X0, preprocessed_obs = self.preprocess_obs(obs)
        self.preprocessed_X0, self.preprocessed_obs = preprocessed_X0, preprocessed_obs
        q_f_0_feed = preprocessed_X0

        def step(t, feed):
            first = t == 0
            d1 = q0 if first else q1
            if self.use_2_q:
                X_t, q_lp, f_lp = self.sample_from_2_dist(d1, self.q2, feed, preprocessed_obs[t], eps[t])
            else:
                X_t, q_lp = d1.sample_and_log_prob(feed, eps[t])
            if not (self.use_bootstrap and self.use_2_q):
                f_lp = f.log_prob(feed, X_t)
            g_lp = self.g.log_prob(X_t, obs[:, t])
            return X_t, f_lp + g_lp - q_lp

        Xs, X_ancs, log_Ws, idxs = [], [], [], []
        X_anc, log_norm_W = None, None
        for t in range(T):
            X_t, log_alpha = step(t, q_f_0_feed if t == 0 else X_anc)
            log_W = log_alpha - logN if t == 0 else log_alpha + log_norm_W
            X_anc, idx = self.resample_X(X_t, log_W,
                                         u=None if u is None else u[t],
                                         idx=None if idx_tf is None else idx_tf[t],
                                         sample_size=N, resample_particles=self.resample_particles)
            log_norm_W = log_W - logsumexp(log_W, 0)
            if self.resample_particles:
                log_norm_W = torch.full_like(log_W, -logN)
            Xs.append(X_t); X_ancs.append(X_anc); log_Ws.append(log_W); idxs.append(idx)
        self.idx_f = None if idxs[0] is None else torch.stack(idxs)
        return torch.stack(Xs), torch.stack(X_ancs), torch.stack(log_Ws)

    @staticmethod
    def compute_log_ZSMC(log_Ws):                # SVO.py:302-311
        return logsumexp(log_Ws, 1).sum(0).mean()

    def get_log_ZSMC(self, obs, noise):          # SVO.py:31-58
        X_prevs, X_ancestors, log_Ws = self.SMC(obs, noise)
        log_ZSMC = self.compute_log_ZSMC(log_Ws)
        log = {"Xs": X_ancestors.permute(2, 0, 1, 3), "X_prevs": X_prevs,
               "X_ancestors": X_ancestors, "log_Ws": log_Ws, "idx_f": self.idx_f}
        return log_ZSMC, log

    def n_step_prediction(self, n_steps, hidden, obs):   # SVO.py:371-404
        x = hidden.mean(2)
        y_hat = []
        for _ in range(n_steps):
            y_hat.append(self.g.mean(x))
            x = self.f.mean(x[:, :-1])
        y_hat.append(self.g.mean(x))
        y = [obs[:, k:] for k in range(n_steps + 1)]
        return y_hat, y

    def get_nextX(self, X):                      # SVO.py:406-409
        return self.f.mean(X)


class OracleAESMC(OracleSVO):                    # src/SMC/AESMC.py:8-11
    def __init__(self, params, flags):
        super().__init__(params, flags, smooth_obs=False, resample_particles=True)


class OracleIWAE(OracleSVO):                     # src/SMC/IWAE.py:8-12
    def __init__(self, params, flags):
        super().__init__(params, flags, smooth_obs=False, resample_particles=False)


class OraclePSVO(OracleSVO):
    """PSVO -- src/SMC/PSVO.py:8-216 (bidirectional-RNN backward proposals)."""

    def __init__(self, params, flags):
        super().__init__(params, flags, smooth_obs=False, resample_particles=True)
        self.M = flags["n_particles_for_BSim_proposal"]
        self.q1_inv = OracleMVN(params["q1_inv"])
        self.BSim_q_init = OracleMVN(params["BSim_q_init"])
        self.BSim_q2 = OracleMVN(params["BSim_q2"])
        self.BSim_use_single_RNN = flags.get("BSim_use_single_RNN", False)

    def BS_preprocess_obs(self, obs):            # PSVO.py:205-216
        if self.BSim_use_single_RNN:
            outputs = static_rnn(obs, self.params["bRNN"]["y_smoother"])
            return None, list(outputs.unbind(1))     # (the final LSTM state the reference returns here is never read)
        return self.preprocess_obs_w_bRNN(obs)

    @staticmethod
    def compute_log_ZSMC_bsim(f_lps, g_lps, Omegas):   # PSVO.py:52-67
        N = f_lps.shape[1]
        joint = (f_lps + g_lps).sum(0)
        proposal = Omegas.sum(0)
        return (logsumexp(joint - proposal, 0) - math.log(float(N))).mean()

    def backward_simulation_w_proposal(self, Xs, log_Ws, obs, noise):   # PSVO.py:69-203
        T, N, B, Dx = Xs.shape
        M = self.M
        eps, u, idx_tf = noise["eps_b"], noise.get("u_b"), noise.get("idx_b")
        logM = math.log(float(M))
        _, enc = self.BS_preprocess_obs(obs)

        def pick(t):
            return (None if u is None else u[t]), (None if idx_tf is None else idx_tf[t])

        def filter_term(x_t, tm1):
            # PSVO.py:128-133: (M, N_i, N_j, B) tile, materialised like the reference
            tiled = x_t.unsqueeze(2).expand(M, N, N, B, Dx)
            f_tm1 = self.f.log_prob(Xs[tm1], tiled)
            log_W_tm1 = log_Ws[tm1] - logsumexp(log_Ws[tm1], 0)
            return logsumexp(f_tm1 + log_W_tm1, 2)

        bw_Xs, f_lps, g_lps, Omegas, sels = [None] * T, [None] * T, [None] * T, [None] * T, [None] * T

        # t = T-1 (PSVO.py:82-108)
        t = T - 1
        x, q_lp = self.BSim_q_init.sample_and_log_prob(enc[t], eps[t])        # (M,N,B,Dx),(M,N,B)
        Lam = filter_term(x, t - 1)
        g_lp = self.g.log_prob(x, obs[:, t])
        omega = Lam + g_lp - q_lp
        omega = omega - logsumexp(omega, 0, keepdim=True)
        ut, it = pick(t)
        (x, omega, g_lp, q_lp), sel = self.resample_X([x, omega, g_lp, q_lp], omega, u=ut, idx=it, sample_size=())
        bw_Xs[t], g_lps[t], Omegas[t], sels[t] = x, g_lp, omega + q_lp + logM, sel
        x_tp1 = x

        # t = T-2 .. 1 (PSVO.py:116-151)
        for t in range(T - 2, 0, -1):
            x, q_lp, _ = self.sample_from_2_dist(self.q1_inv, self.BSim_q2, x_tp1, enc[t], eps[t])
            f_lp = self.f.log_prob(x, x_tp1)
            Lam = filter_term(x, t - 1)
            g_lp = self.g.log_prob(x, obs[:, t])
            omega = Lam + f_lp + g_lp - q_lp
            omega = omega - logsumexp(omega, 0)
            ut, it = pick(t)
            (x, omega, f_lp, g_lp, q_lp), sel = self.resample_X([x, omega, f_lp, g_lp, q_lp], omega,
                                                                u=ut, idx=it, sample_size=())
            bw_Xs[t], f_lps[t + 1], g_lps[t], Omegas[t], sels[t] = x, f_lp, g_lp, omega + q_lp + logM, sel
            x_tp1 = x

        # t = 0 (PSVO.py:158-190)
        x, q_lp, _ = self.sample_from_2_dist(self.q1_inv, self.BSim_q2, x_tp1, enc[0], eps[0])
        f_lp = self.f.log_prob(x, x_tp1)
        g_lp = self.g.log_prob(x, obs[:, 0])
        mu_0 = self.preprocessed_X0                                            # cached by SMC()
        if not (self.use_bootstrap and self.use_2_q):
            f_init = self.f.log_prob(mu_0, x)
        else:
            f_init = self.q0.log_prob(mu_0, x)
        omega = f_init + f_lp + g_lp - q_lp
        omega = omega - logsumexp(omega, 0)
        ut, it = pick(0)
        (x, omega, f_lp, f_init, g_lp, q_lp), sel = self.resample_X(
            [x, omega, f_lp, f_init, g_lp, q_lp], omega, u=ut, idx=it, sample_size=())
        bw_Xs[0], f_lps[1], f_lps[0], g_lps[0], Omegas[0], sels[0] = x, f_lp, f_init, g_lp, omega + q_lp + logM, sel

        self.idx_b = torch.stack(sels)
        return torch.stack(bw_Xs), torch.stack(f_lps), torch.stack(g_lps), torch.stack(Omegas)

    def get_log_ZSMC(self, obs, noise):          # PSVO.py:21-50
        X_prevs, X_ancestors, log_Ws = self.SMC(obs, noise)
        bw_Xs, f_lps, g_lps, Omegas = self.backward_simulation_w_proposal(X_prevs, log_Ws, obs, noise)
        log_ZSMC = self.compute_log_ZSMC_bsim(f_lps, g_lps, Omegas)
        log = {"Xs": bw_Xs.permute(2, 0, 1, 3), "X_prevs": X_prevs, "X_ancestors": X_ancestors,
               "log_Ws": log_Ws, "idx_f": self.idx_f, "bw_Xs": bw_Xs, "f_log_probs": f_lps,
               "g_log_probs": g_lps, "bw_log_Omegas": Omegas, "idx_b": self.idx_b}
        return log_ZSMC, log


def evaluate_R_square(y_hat, y):
    """trainer.evaluate_R_square -- src/trainer.py:322-335 (numpy in the reference)."""
    out = []
    for y_hat_i, y_i in zip(y_hat, y):
        mse = ((y_hat_i - y_i) ** 2).sum()
        var = ((y_i - y_i.mean(0, keepdim=True)) ** 2).sum()
        out.append(1 - mse / var)
    return torch.stack(out)


OBJECTIVES = {"SVO": OracleSVO, "AESMC": OracleAESMC, "IWAE": OracleIWAE, "PSVO": OraclePSVO}


# --------------------------------------------------------------------------- #
# parameter / noise factories used by tests, fixtures and the cpu_baseline leg
# --------------------------------------------------------------------------- #
def he_normal(gen, fan_in, fan_out, dtype):
    """keras he_normal: truncated normal (|z|<2), stddev sqrt(2/fan_in)/.87962566."""
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    z = torch.empty(fan_in, fan_out, dtype=torch.float64)
    torch.nn.init.trunc_normal_(z, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return (z * std).to(dtype)


def make_mlp(gen, Din, Dhs, Dout, sigma_init, sigma_min, dtype, bias_scale=0.0, cov_head=False):
    layers, d = [], Din
    for Dh in Dhs:
        b = torch.zeros(Dh, dtype=dtype)
        if bias_scale:
            b = (torch.randn(Dh, generator=gen, dtype=torch.float64) * bias_scale).to(dtype)
        layers.append((he_normal(gen, d, Dh, dtype), b))
        d = Dh
    b = torch.zeros(Dout, dtype=dtype)
    if bias_scale:
        b = (torch.randn(Dout, generator=gen, dtype=torch.float64) * bias_scale).to(dtype)
    out = {"layers": layers, "mu": (he_normal(gen, d, Dout, dtype), b),
           "sigma_raw": torch.full((Dout,), float(sigma_init), dtype=dtype), "sigma_min": float(sigma_min)}
    if cov_head:      # sigma_layer: he_normal kernel, bias Constant(1.0) (MLP.py:40-46), diag_cov: Dout outputs
        out["sigma"] = (he_normal(gen, d, Dout, dtype), torch.ones(Dout, dtype=dtype))
    return out


def make_lstm(gen, Din, Dh, dtype):
    lim = math.sqrt(6.0 / (Din + Dh + 4 * Dh))          # glorot_uniform on the (in+h, 4h) kernel
    W = ((torch.rand(Din + Dh, 4 * Dh, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)
    return (W, torch.zeros(4 * Dh, dtype=dtype))


def make_params(flags, seed=0, dtype=torch.float64, bias_scale=0.0):
    """Which nets exist and how sigma is wired -- src/model.py:17-192."""
    gen = torch.Generator().manual_seed(seed)
    Dx, Dy = flags["Dx"], flags["Dy"]
    smooth = flags.get("objective") == "SVO"
    psvo = flags.get("objective") in ("PSVO", "PSVOwR")
    Dhs_y = flags.get("y_smoother_Dhs", [32])
    Dhs_x0 = flags.get("X0_smoother_Dhs", [32])
    stacked = flags.get("use_stack_rnn", True)
    single = psvo and flags.get("BSim_use_single_RNN", False)
    E = 2 * Dhs_y[-1] if smooth else Dy
    sep = flags.get("X0_use_separate_RNN", True)
    # X0 feature: concat(outputs[-1], outputs[0]) of the (B,T,2Dh) stack outputs, or concat(fw[-1], bw[0])
    E0 = (4 if stacked else 2) * (Dhs_x0[-1] if sep else Dhs_y[-1]) if smooth else Dy
    both = flags["use_bootstrap"] and flags["use_2_q"]
    q0_in = E0 if both else Dx
    si, sm = flags.get("sigma_init", 5.0), flags.get("sigma_min", 1.0)
    H = flags.get("layers", [32])
    cov_head = bool(flags.get("output_cov", False))
    if cov_head and not flags.get("diag_cov", False):
        raise NotImplementedError("full-covariance head (output_cov without diag_cov): out of scope, SURVEY.md section 2")
    mk = lambda i, o: make_mlp(gen, i, H, o, si, sm, dtype, bias_scale, cov_head=cov_head)
    P = {"q0": mk(q0_in, Dx), "q1": mk(Dx, Dx)}
    if flags["use_2_q"]:
        P["q2"] = mk(E, Dx)
    if not flags["use_bootstrap"]:
        P["f"] = mk(Dx, Dx)
    P["g"] = mk(Dx, Dy)                                   # g_sigma_init = f_sigma_init quirk: model.py:37
    if not both:
        lim = math.sqrt(6.0 / E0)                         # he_uniform Dense, model.py:188-192
        W = ((torch.rand(E0, Dx, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)
        P["X0_transformer"] = (W, torch.zeros(Dx, dtype=dtype))
    if smooth or psvo:
        def stack(Dhs, chained):
            # layer input: concat of both directions of the previous layer (stack_bidirectional), or the
            # previous layer of the SAME direction (MultiRNNCell: use_stack_rnn=False / static_rnn)
            out, d = [], Dy
            for Dh in Dhs:
                out.append({"fw": make_lstm(gen, d, Dh, dtype), "bw": make_lstm(gen, d, Dh, dtype)})
                d = Dh if chained else 2 * Dh
            return out
        P["bRNN"] = {"y_smoother": stack(Dhs_y, (not stacked) or single),
                     "X0_smoother": stack(Dhs_x0, not stacked) if sep else None}
    if psvo:
        Eb = Dhs_y[-1] if single else 2 * Dhs_y[-1]
        P["BSim_q_init"] = mk(Eb, Dx)
        P["q1_inv"] = mk(Dx, Dx)
        P["BSim_q2"] = mk(Eb, Dx)
    return P


def make_noise(flags, B, T, seed=1234, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    N, Dx = flags["n_particles"], flags["Dx"]
    noise = {"eps_f": torch.randn(T, N, B, Dx, generator=gen, dtype=torch.float64).to(dtype),
             "u_f": torch.rand(T, N, B, generator=gen, dtype=torch.float64).to(dtype)}
    if flags.get("objective") in ("PSVO", "PSVOwR"):
        M = flags["n_particles_for_BSim_proposal"]
        noise["eps_b"] = torch.randn(T, M, N, B, Dx, generator=gen, dtype=torch.float64).to(dtype)
        noise["u_b"] = torch.rand(T, N, B, generator=gen, dtype=torch.float64).to(dtype)
    if flags.get("objective") == "PSVOwR":
        noise["u_r"] = torch.rand(T, N, B, generator=gen, dtype=torch.float64).to(dtype)
    return noise


def params_to(P, dtype):
    def cv(x):
        if torch.is_tensor(x):
            return x.to(dtype)
        if isinstance(x, dict):
            return {k: cv(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(cv(v) for v in x)
        return x
    return cv(P)


def fhn_synthetic(n, T, seed=0, dt=0.15, obs_cov=0.01, dtype=torch.float64):
    """RK4 restatement of the FHN generator -- src/transformation/fhn.py:26-35,
    src/utils/data_generator.py:38-45 (the reference integrates with scipy odeint)."""
    gen = torch.Generator().manual_seed(seed)
    a, b, c, I = 1.0, 0.95, 0.05, 1.0

    def rhs(x):
        V, w = x[..., 0], x[..., 1]
        return torch.stack([V - V ** 3 / 3 - w + I, a * (b * V - c * w)], -1)
    x = (torch.rand(n, 2, generator=gen, dtype=torch.float64) * 5.0) - 2.5
    xs = [x]
    sub = 4
    h = dt / sub
    for _ in range(T - 1):
        for _ in range(sub):
            k1 = rhs(x); k2 = rhs(x + 0.5 * h * k1); k3 = rhs(x + 0.5 * h * k2); k4 = rhs(x + h * k3)
            x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        xs.append(x)
    hidden = torch.stack(xs, 1)
    obs = hidden[..., :1] + math.sqrt(obs_cov) * torch.randn(n, T, 1, generator=gen, dtype=torch.float64)
    return hidden.to(dtype), obs.to(dtype)


class OraclePSVOwR(OraclePSVO):
    """PSVOwR -- src/SMC/PSVOwR.py:8-211: PSVO whose backward chains are additionally resampled
    across the particle axis every step, with a per-step ELBO.

    Extra noise: u_r (T, N, B) uniforms (or idx_r (T, N, B) teacher-forced ancestors) for the
    cross-chain multinomial draw (PSVOwR.py:103,145,185)."""

    @staticmethod
    def compute_log_ZSMC_wr(bw_log_W):                     # PSVOwR.py:52-63
        N = bw_log_W.shape[1]
        return (logsumexp(bw_log_W, 1) - math.log(float(N))).sum(0).mean()

    def backward_simulation_w_proposal(self, Xs, log_Ws, obs, noise):   # PSVOwR.py:65-198
        T, N, B, Dx = Xs.shape
        M = self.M
        eps, u, idx_tf = noise["eps_b"], noise.get("u_b"), noise.get("idx_b")
        u_r, idx_r = noise.get("u_r"), noise.get("idx_r")
        logM = math.log(float(M))
        _, enc = self.BS_preprocess_obs(obs)

        def pick(t):
            return (None if u is None else u[t]), (None if idx_tf is None else idx_tf[t])

        def pick_r(t):
            return (None if u_r is None else u_r[t]), (None if idx_r is None else idx_r[t])

        def filter_term(x_t, tm1):
            tiled = x_t.unsqueeze(2).expand(M, N, N, B, Dx)
            f_tm1 = self.f.log_prob(Xs[tm1], tiled)
            log_W_tm1 = log_Ws[tm1] - logsumexp(log_Ws[tm1], 0)
            return logsumexp(f_tm1 + log_W_tm1, 2)

        bw_Xs, bw_Xanc, bw_W, sels, ancs = [None] * T, [None] * T, [None] * T, [None] * T, [None] * T

        # t = T-1 (PSVOwR.py:77-104)
        t = T - 1
        x, q_lp = self.BSim_q_init.sample_and_log_prob(enc[t], eps[t])
        Lam = filter_term(x, t - 1)
        g_lp = self.g.log_prob(x, obs[:, t])
        omega = Lam + g_lp - q_lp
        omega = omega - logsumexp(omega, 0, keepdim=True)
        W = Lam + g_lp - q_lp - omega - logM
        ut, it = pick(t)
        (x, W, omega), sel = self.resample_X([x, W, omega], omega, u=ut, idx=it, sample_size=())
        ur, ir = pick_r(t)
        x_anc, anc = self.resample_X(x, omega, u=ur, idx=ir, sample_size=N)
        bw_Xs[t], bw_W[t], bw_Xanc[t], sels[t], ancs[t] = x, W, x_anc, sel, anc

        # t = T-2 .. 1 (PSVOwR.py:113-149)
        for t in range(T - 2, 0, -1):
            x_tp1 = bw_Xanc[t + 1]
            x, q_lp, _ = self.sample_from_2_dist(self.q1_inv, self.BSim_q2, x_tp1, enc[t], eps[t])
            f_lp = self.f.log_prob(x, x_tp1)
            Lam = filter_term(x, t - 1)
            g_lp = self.g.log_prob(x, obs[:, t])
            omega = Lam + f_lp + g_lp - q_lp
            omega = omega - logsumexp(omega, 0)
            W = Lam + g_lp
            ut, it = pick(t)
            (x, omega, W, q_lp), sel = self.resample_X([x, omega, W, q_lp], omega, u=ut, idx=it, sample_size=())
            W = W - (q_lp + omega + logM)
            ur, ir = pick_r(t)
            x_anc, anc = self.resample_X(x, omega, u=ur, idx=ir, sample_size=N)
            bw_Xs[t], bw_W[t], bw_Xanc[t], sels[t], ancs[t] = x, W, x_anc, sel, anc

        # t = 0 (PSVOwR.py:155-187)
        x_tp1 = bw_Xanc[1]
        x, q_lp, _ = self.sample_from_2_dist(self.q1_inv, self.BSim_q2, x_tp1, enc[0], eps[0])
        f_lp = self.f.log_prob(x, x_tp1)
        g_lp = self.g.log_prob(x, obs[:, 0])
        mu_0 = self.preprocessed_X0
        if not (self.use_bootstrap and self.use_2_q):
            f_init = self.f.log_prob(mu_0, x)
        else:
            f_init = self.q0.log_prob(mu_0, x)
        omega = f_init + f_lp + g_lp - q_lp
        omega = omega - logsumexp(omega, 0)
        W = f_init + g_lp
        ut, it = pick(0)
        (x, omega, W, q_lp), sel = self.resample_X([x, omega, W, q_lp], omega, u=ut, idx=it, sample_size=())
        W = W - (q_lp + omega + logM)
        ur, ir = pick_r(0)
        x_anc, anc = self.resample_X(x, omega, u=ur, idx=ir, sample_size=N)
        bw_Xs[0], bw_W[0], bw_Xanc[0], sels[0], ancs[0] = x, W, x_anc, sel, anc

        self.idx_b, self.idx_r = torch.stack(sels), torch.stack(ancs)
        return torch.stack(bw_Xs), torch.stack(bw_Xanc), torch.stack(bw_W)

    def get_log_ZSMC(self, obs, noise):                   # PSVOwR.py:21-50
        X_prevs, X_ancestors, log_Ws = self.SMC(obs, noise)
        bw_Xs, bw_Xanc, bw_W = self.backward_simulation_w_proposal(X_prevs, log_Ws, obs, noise)
        log_ZSMC = self.compute_log_ZSMC_wr(bw_W)
        log = {"Xs": bw_Xanc.permute(2, 0, 1, 3), "X_prevs": X_prevs, "X_ancestors": X_ancestors,
               "log_Ws": log_Ws, "idx_f": self.idx_f, "bw_Xs": bw_Xs, "bw_X_ancestors": bw_Xanc,
               "bw_log_W": bw_W, "idx_b": self.idx_b, "idx_r": self.idx_r}
        return log_ZSMC, log


OBJECTIVES["PSVOwR"] = OraclePSVOwR
