"""SSM -- mirror of reference src/model.py:10-192: which networks exist, their shapes, what is
shared (f == q1 under use_bootstrap) and how sigma is wired (g_sigma_init = f_sigma_init quirk,
model.py:37).  TF placeholders do not exist here: `obs` / `hidden` are fed as tensors.
"""
import math

import torch
from torch import nn

from .distribution.mvn import tf_mvn
from .distribution.poisson import tf_poisson
from .transformation.MLP import MLP_transformation


class LSTMBlockCellLayer(nn.Module):
    """One direction of tf.contrib.rnn.LSTMBlockCell (model.py:164-176) run over a sequence.

    Parameters keep the TF layout: kernel (in + h, 4h) with gate order (i, j, f, o), bias (4h),
    forget_bias = 1 added to f.  The sequence loop itself is upstream of the particle path
    (SURVEY section 8f-1); it runs through torch's fused LSTM (MIOpen) after a layout permute.
    """

    def __init__(self, Din, Dh, name="lstm"):
        super().__init__()
        self.Din, self.Dh, self.name = Din, Dh, name
        lim = math.sqrt(6.0 / (Din + Dh + 4 * Dh))          # glorot_uniform, TF default
        self.kernel = nn.Parameter((torch.rand(Din + Dh, 4 * Dh) * 2 - 1) * lim)
        self.bias = nn.Parameter(torch.zeros(4 * Dh))

    def torch_weights(self):
        Dh = self.Dh
        i, j, f, o = self.kernel.split(Dh, dim=1)
        w = torch.cat([i, f, j, o], dim=1)                  # torch gate order (i, f, g, o)
        bi, bj, bf, bo = self.bias.split(Dh)
        b = torch.cat([bi, bf + 1.0, bj, bo])
        w_ih = w[:self.Din].t().contiguous()
        w_hh = w[self.Din:].t().contiguous()
        return w_ih, w_hh, b, torch.zeros_like(b)


class StackBiRNN(nn.Module):
    """The observation encoder over lists of LSTMBlockCells, in the reference's three wirings:

    mode "stack": tf.contrib.rnn.stack_bidirectional_dynamic_rnn (src/SMC/SVO.py:337-341) -- each layer
                  consumes concat(fw, bw) of the previous one; output (B, T, 2 Dh).
    mode "multi": tf.nn.bidirectional_dynamic_rnn over two MultiRNNCells (use_stack_rnn=False,
                  src/model.py:168-170, src/SMC/SVO.py:342-346) -- layers are chained inside each direction;
                  output concat(fw, bw) (B, T, 2 Dh) of the last layer.
    mode "uni":   tf.nn.static_rnn(MultiRNNCell(y_smoother_f)) (BSim_use_single_RNN, src/SMC/PSVO.py:208-212)
                  -- forward cells only; output (B, T, Dh).
    On the GPU every layer is ONE persistent launch (psvo_bilstm_forward / psvo_bilstm_backward) and nothing else: a cell
    width the kernels do not cover (Dh not in 8/16/32/64) raises ValueError from the launch, there is no PyTorch fallback for
    tensors in HBM.  CPU tensors run torch's LSTM on the re-ordered TF weights: the host-side mirror used by the CPU tests of
    the host logic on a machine without a GPU, never part of the GPU path.
    """

    def __init__(self, Din, Dhs, name, mode="stack"):
        super().__init__()
        assert mode in ("stack", "multi", "uni")
        self.mode = mode
        self.fw, self.bw = nn.ModuleList(), nn.ModuleList()
        d = Din
        for i, Dh in enumerate(Dhs):
            self.fw.append(LSTMBlockCellLayer(d, Dh, "{}_f_{}".format(name, i)))
            if mode != "uni":           # (TF builds a cell's variables on first use: the unused bw cells have none)
                self.bw.append(LSTMBlockCellLayer(d, Dh, "{}_b_{}".format(name, i)))
            d = 2 * Dh if mode == "stack" else Dh
        self.Dh_last = Dhs[-1]
        self.Dout = self.Dh_last if mode == "uni" else 2 * self.Dh_last

    @staticmethod
    def _gbufs(fw, bw):
        gf, gb = fw.__dict__.get("_flat_grad"), bw.__dict__.get("_flat_grad")
        return (gf, gb) if (gf is not None and gb is not None) else None

    def forward(self, x_BTD):
        h = x_BTD
        if h.is_cuda:
            from .autograd import BiLSTMFunction
            if self.mode == "stack":
                for fw, bw in zip(self.fw, self.bw):
                    h = BiLSTMFunction.apply(self._gbufs(fw, bw), h, fw.kernel, fw.bias, bw.kernel, bw.bias)
                return h
            if self.mode == "uni":
                # the bidirectional kernel with the forward cell in both slots; only the forward half is kept
                for fw in self.fw:
                    h = BiLSTMFunction.apply(None, h, fw.kernel, fw.bias, fw.kernel, fw.bias)[..., :fw.Dh]
                return h
            # "multi": layer i >= 1 reads only its own direction of the previous layer: embed its kernel into one
            # that takes concat(fw, bw) with zero rows for the other direction (one launch per layer)
            for i, (fw, bw) in enumerate(zip(self.fw, self.bw)):
                if i == 0:
                    h = BiLSTMFunction.apply(self._gbufs(fw, bw), h, fw.kernel, fw.bias, bw.kernel, bw.bias)
                else:
                    Dp = fw.Din
                    z = fw.kernel.new_zeros(Dp, 4 * fw.Dh)
                    kf = torch.cat([fw.kernel[:Dp], z, fw.kernel[Dp:]], dim=0)
                    kb = torch.cat([z, bw.kernel[:Dp], bw.kernel[Dp:]], dim=0)
                    h = BiLSTMFunction.apply(None, h, kf, fw.bias, kb, bw.bias)
            return h
        B = h.shape[0]

        def run(cell, x, reverse):
            z = x.new_zeros(1, B, cell.Dh)
            if reverse:
                x = x.flip(1)
            y, _, _ = torch._VF.lstm(x, (z, z), list(cell.torch_weights()), True, 1, 0.0, self.training, False, True)
            return y.flip(1) if reverse else y
        if self.mode == "stack":
            for fw, bw in zip(self.fw, self.bw):
                flat = list(fw.torch_weights()) + list(bw.torch_weights())
                z = h.new_zeros(2, B, fw.Dh)
                h, _, _ = torch._VF.lstm(h, (z, z), flat, True, 1, 0.0, self.training, True, True)
            return h
        if self.mode == "uni":
            for fw in self.fw:
                h = run(fw, h, False)
            return h
        f = b = h
        for fw, bw in zip(self.fw, self.bw):
            f, b = run(fw, f, False), run(bw, b, True)
        return torch.cat([f, b], dim=-1)


class SSM(nn.Module):
    """state space model: keeps q, f, g and the obs smoothers (model.py:10-62)."""

    def __init__(self, FLAGS):
        super().__init__()
        self.Dx, self.Dy = FLAGS.Dx, FLAGS.Dy
        self.time, self.batch_size = FLAGS.time, FLAGS.batch_size

        split = lambda s: [int(x) for x in str(s).split(",")]
        self.q0_layers, self.q1_layers, self.q2_layers = split(FLAGS.q0_layers), split(FLAGS.q1_layers), split(FLAGS.q2_layers)
        self.f_layers, self.g_layers = split(FLAGS.f_layers), split(FLAGS.g_layers)

        self.q0_sigma_init, self.q0_sigma_min = FLAGS.q0_sigma_init, FLAGS.q0_sigma_min
        self.q1_sigma_init, self.q1_sigma_min = FLAGS.q1_sigma_init, FLAGS.q1_sigma_min
        self.q2_sigma_init, self.q2_sigma_min = FLAGS.q2_sigma_init, FLAGS.q2_sigma_min
        self.f_sigma_init, self.f_sigma_min = FLAGS.f_sigma_init, FLAGS.f_sigma_min
        self.g_sigma_init, self.g_sigma_min = FLAGS.f_sigma_init, FLAGS.g_sigma_min    # sic, model.py:37

        self.y_smoother_Dhs = split(FLAGS.y_smoother_Dhs)
        self.X0_smoother_Dhs = split(FLAGS.X0_smoother_Dhs)

        self.output_cov, self.diag_cov = FLAGS.output_cov, FLAGS.diag_cov
        self.use_bootstrap, self.use_2_q = FLAGS.use_bootstrap, FLAGS.use_2_q
        self.poisson_emission = FLAGS.poisson_emission
        self.X0_use_separate_RNN, self.use_stack_rnn = FLAGS.X0_use_separate_RNN, FLAGS.use_stack_rnn
        self.PSVO, self.PSVOwR, self.SVO = FLAGS.PSVO, getattr(FLAGS, "PSVOwR", False), FLAGS.SVO
        self.BSim_use_single_RNN = FLAGS.BSim_use_single_RNN

        # placeholders of the reference (model.py:63-65) have no equivalent; kept as feed keys
        self.obs, self.hidden = "obs", "hidden"

        self.init_trans()
        self.init_dist()
        self.init_RNNs()

    # feature widths of the proposal inputs (SVO.py:313-331,362-367)
    def _feature_dims(self):
        if self.SVO:
            E = 2 * self.y_smoother_Dhs[-1]
            Dh0 = self.X0_smoother_Dhs[-1] if self.X0_use_separate_RNN else self.y_smoother_Dhs[-1]
            # concat(outputs[-1], outputs[0]) of the stacked (B,T,2Dh) outputs, or concat(fw[-1], bw[0]) (SVO.py:360-367)
            E0 = (4 if self.use_stack_rnn else 2) * Dh0
        else:
            E = E0 = self.Dy
        return E, E0

    def init_trans(self):                                    # model.py:67-110
        E, E0 = self._feature_dims()
        both = self.use_bootstrap and self.use_2_q
        q0_in = E0 if both else self.Dx                      # X0_transformer maps E0 -> Dx otherwise
        mk = lambda layers, Dout, Din, name: MLP_transformation(layers, Dout, Din, output_cov=self.output_cov,
                                                                diag_cov=self.diag_cov, name=name)
        self.q0_tran = mk(self.q0_layers, self.Dx, q0_in, "q0_tran")
        self.q1_tran = mk(self.q1_layers, self.Dx, self.Dx, "q1_tran")
        self.q2_tran = mk(self.q2_layers, self.Dx, E, "q2_tran") if self.use_2_q else None
        if self.PSVO or self.PSVOwR:
            single = self.BSim_use_single_RNN                # features of static_rnn(y_smoother_f): (B, Dh)
            Eb = (1 if single else 2) * self.y_smoother_Dhs[-1]
            self.BSim_q_init_tran = mk(self.q0_layers, self.Dx, Eb, "BSim_q_init_tran")
            self.q1_inv_tran = mk(self.q1_layers, self.Dx, self.Dx, "q1_inv_tran")
            self.BSim_q2_tran = mk(self.q2_layers, self.Dx, Eb, "BSim_q2_tran")
        self.f_tran = self.q1_tran if self.use_bootstrap else mk(self.f_layers, self.Dx, self.Dx, "f_tran")
        self.g_tran = mk(self.g_layers, self.Dy, self.Dx, "g_tran")

    def init_dist(self):                                     # model.py:112-160
        self.q0_dist = tf_mvn(self.q0_tran, self.q0_sigma_init, self.q0_sigma_min, "q0_dist")
        self.q1_dist = tf_mvn(self.q1_tran, self.q1_sigma_init, self.q1_sigma_min, "q1_dist")
        self.q2_dist = tf_mvn(self.q2_tran, self.q2_sigma_init, self.q2_sigma_min, "q2_dist") if self.use_2_q else None
        if self.PSVO or self.PSVOwR:
            self.Bsim_q_init_dist = tf_mvn(self.BSim_q_init_tran, self.q0_sigma_init, self.q0_sigma_min, "BSim_q_init_dist")
            self.q1_inv_dist = tf_mvn(self.q1_inv_tran, self.q1_sigma_init, self.q1_sigma_min, "q1_inv_dist")
            self.BSim_q2_dist = tf_mvn(self.BSim_q2_tran, self.q2_sigma_init, self.q2_sigma_min, "BSim_q2_dist")
        if self.use_bootstrap:
            self.f_dist = self.q1_dist
        else:
            self.f_dist = tf_mvn(self.f_tran, self.f_sigma_init, self.f_sigma_min, "f_dist")
        if self.poisson_emission:                            # model.py:153-155
            self.g_dist = tf_poisson(self.g_tran, "g_dist")
        else:
            self.g_dist = tf_mvn(self.g_tran, self.g_sigma_init, self.g_sigma_min, "g_dist")

    def init_RNNs(self):                                     # model.py:162-192
        if self.SVO or self.PSVO or self.PSVOwR:
            mode = "stack" if self.use_stack_rnn else "multi"
            single = (self.PSVO or self.PSVOwR) and self.BSim_use_single_RNN
            self.y_smoother = StackBiRNN(self.Dy, self.y_smoother_Dhs, "y_smoother", "uni" if single else mode)
            self.X0_smoother = (StackBiRNN(self.Dy, self.X0_smoother_Dhs, "X0_smoother", mode)
                                if self.X0_use_separate_RNN else None)
            self.bRNN = (self.y_smoother, self.X0_smoother)
        else:
            self.y_smoother = self.X0_smoother = None
            self.bRNN = None
        if not (self.use_bootstrap and self.use_2_q):
            _, E0 = self._feature_dims()
            lim = math.sqrt(6.0 / E0)                        # Dense(Dx, he_uniform)
            self.X0_transformer_kernel = nn.Parameter((torch.rand(E0, self.Dx) * 2 - 1) * lim)
            self.X0_transformer_bias = nn.Parameter(torch.zeros(self.Dx))

    def sigmas(self):
        """{distribution: scale vector} for every tf_mvn of the model.  With a flat parameter buffer
        (optim.FlatParams) all of them come from ONE fused launch; otherwise each distribution computes its own."""
        blk = self.__dict__.get("_sigma_block")
        if blk is not None and blk["raw"].is_cuda:
            from .autograd import SigmaFunction
            sig = SigmaFunction.apply(blk, *[d.sigma_con for d in blk["dists"]])
            return {id(d): s for d, s in zip(blk["dists"], sig.split(blk["sizes"]))}
        out = {}
        for mod in self.modules():
            if isinstance(mod, tf_mvn) and id(mod) not in out:
                out[id(mod)] = mod.get_sigma()
        return out

    def X0_transformer(self, x):
        return x @ self.X0_transformer_kernel + self.X0_transformer_bias

    # ------------------------------------------------------------------------------------------
    def export_reference_layout(self, dtype=torch.float64):
        """All variables in the reference's (keras / TF) layout, as plain CPU tensors.
        Used to hand the same parameters to a checker or to a reference-format checkpoint."""
        cv = lambda t: t.detach().to("cpu", dtype).clone()

        def dist(d):
            tr = d.transformation
            out = {"layers": [(cv(W), cv(b)) for W, b in zip(tr.kernels, tr.biases)],
                   "mu": (cv(tr.mu_kernel), cv(tr.mu_bias))}
            if tr.output_cov:                                # sigma_layer (MLP.py:40-46)
                out["sigma"] = (cv(tr.sigma_kernel), cv(tr.sigma_bias))
            if isinstance(d, tf_mvn):                        # (tf_poisson has no scale variable)
                out.update({"sigma_raw": cv(d.sigma_con), "sigma_min": float(d.sigma_min)})
            return out

        P = {"q0": dist(self.q0_dist), "q1": dist(self.q1_dist), "g": dist(self.g_dist)}
        if self.use_2_q:
            P["q2"] = dist(self.q2_dist)
        if not self.use_bootstrap:
            P["f"] = dist(self.f_dist)
        if not (self.use_bootstrap and self.use_2_q):
            P["X0_transformer"] = (cv(self.X0_transformer_kernel), cv(self.X0_transformer_bias))
        if self.bRNN is not None:
            def stack(s):
                if s is None:
                    return None
                bws = list(s.bw) + [None] * (len(s.fw) - len(s.bw))      # mode "uni": forward cells only
                return [{"fw": (cv(f.kernel), cv(f.bias)), "bw": None if b is None else (cv(b.kernel), cv(b.bias))}
                        for f, b in zip(s.fw, bws)]
            P["bRNN"] = {"y_smoother": stack(self.y_smoother), "X0_smoother": stack(self.X0_smoother)}
        if self.PSVO or self.PSVOwR:
            P["BSim_q_init"] = dist(self.Bsim_q_init_dist)
            P["q1_inv"] = dist(self.q1_inv_dist)
            P["BSim_q2"] = dist(self.BSim_q2_dist)
        return P

    def load_reference_layout(self, P):
        """Inverse of export_reference_layout: copy variables given in the reference's (keras / TF)
        layout into this model (checkpoint import; tests load golden-fixture parameters with it)."""
        def put(dst, src):
            with torch.no_grad():
                dst.copy_(torch.as_tensor(src).to(dst.device, dst.dtype))

        def dist(d, p):
            tr = d.transformation
            for (W, b), (Ws, bs) in zip(zip(tr.kernels, tr.biases), p["layers"]):
                put(W, Ws); put(b, bs)
            put(tr.mu_kernel, p["mu"][0]); put(tr.mu_bias, p["mu"][1])
            if tr.output_cov:
                put(tr.sigma_kernel, p["sigma"][0]); put(tr.sigma_bias, p["sigma"][1])
            if isinstance(d, tf_mvn):
                put(d.sigma_con, p["sigma_raw"])
                d.sigma_min = float(p["sigma_min"])

        dist(self.q0_dist, P["q0"]); dist(self.q1_dist, P["q1"]); dist(self.g_dist, P["g"])
        if self.use_2_q:
            dist(self.q2_dist, P["q2"])
        if not self.use_bootstrap:
            dist(self.f_dist, P["f"])
        if not (self.use_bootstrap and self.use_2_q):
            put(self.X0_transformer_kernel, P["X0_transformer"][0]); put(self.X0_transformer_bias, P["X0_transformer"][1])
        if self.bRNN is not None:
            for nm, s in (("y_smoother", self.y_smoother), ("X0_smoother", self.X0_smoother)):
                if s is None:
                    continue
                for i, (L, f) in enumerate(zip(P["bRNN"][nm], s.fw)):
                    put(f.kernel, L["fw"][0]); put(f.bias, L["fw"][1])
                    if i < len(s.bw):
                        put(s.bw[i].kernel, L["bw"][0]); put(s.bw[i].bias, L["bw"][1])
        if self.PSVO or self.PSVOwR:
            dist(self.Bsim_q_init_dist, P["BSim_q_init"]); dist(self.q1_inv_dist, P["q1_inv"])
            dist(self.BSim_q2_dist, P["BSim_q2"])
        return self
