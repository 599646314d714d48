"""MLP_transformation -- mirror of reference src/transformation/MLP.py:8-86, incl. the diagonal covariance head
(output_cov and diag_cov: `sigma_layer`, MLP.py:40-46,58-61); the full-covariance head (output_cov without diag_cov) is out of
scope.

Weights keep the keras Dense layout (kernel (in, out), y = x @ kernel + bias) so they can be
handed to the HIP kernels (psvo_mlp in include/psvo_hip.h) without a transpose.
"""
import math

import torch
from torch import nn


def _he_normal_(w):
    """keras he_normal: truncated normal, stddev sqrt(2 / fan_in) (MLP.py:29-38)."""
    fan_in = w.shape[0]
    std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
    with torch.no_grad():
        nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std)
    return w


class MLP_transformation(nn.Module):
    def __init__(self, Dhs, Dout, Din, use_residual=False, output_cov=False, diag_cov=False,
                 name="MLP_transformation"):
        super().__init__()
        if output_cov and not diag_cov:
            # reference MLP.py:41,63-66: Dout^2 outputs reshaped to a matrix, cov = A A^T -> MultivariateNormalFullCovariance
            raise NotImplementedError("output_cov without diag_cov (full covariance) is outside the MI355X hot-path scope "
                                      "(SURVEY.md section 2 row 2); output_cov with diag_cov is built")
        self.Dhs, self.Dout, self.Din = list(Dhs), Dout, Din
        self.use_residual = use_residual
        self.output_cov, self.diag_cov = output_cov, diag_cov
        self.name = name
        self.kernels = nn.ParameterList()
        self.biases = nn.ParameterList()
        d = Din
        for Dh in self.Dhs:                                # hidden_{i}: Dense(relu, he_normal)
            self.kernels.append(nn.Parameter(_he_normal_(torch.empty(d, Dh))))
            self.biases.append(nn.Parameter(torch.zeros(Dh)))
            d = Dh
        self.mu_kernel = nn.Parameter(_he_normal_(torch.empty(d, Dout)))   # mu_layer: Dense(linear)
        self.mu_bias = nn.Parameter(torch.zeros(Dout))
        if output_cov:                                     # sigma_layer: Dense(linear, he_normal, bias Constant(1.0)), MLP.py:40-46
            self.sigma_kernel = nn.Parameter(_he_normal_(torch.empty(d, Dout)))
            self.sigma_bias = nn.Parameter(torch.ones(Dout))

    # limits of the fused one-hidden-layer row kernels (psvo_rows_mlp_*, csrc/rows_mlp.hip); anything else that is a plain
    # chain of Dense layers goes layer by layer through psvo_dense_* (csrc/dense.hip, f32 MFMA)
    _NATIVE_H, _NATIVE_DIN, _NATIVE_DOUT = (16, 32, 64), 128, 4

    def _native_refusal(self, Input):
        """why the fused psvo_rows_mlp_* pair cannot evaluate this MLP on `Input` (None if it can)"""
        if self.output_cov:
            return "two output heads (mu_layer, sigma_layer)"
        if len(self.Dhs) != 1:
            return "%d hidden layers %s (the fused kernels take exactly one)" % (len(self.Dhs), self.Dhs)
        if self.use_residual:
            return "use_residual=True"
        if self.Dhs[0] not in self._NATIVE_H:
            return "hidden width %d not in %s" % (self.Dhs[0], self._NATIVE_H)
        if Input.shape[-1] > self._NATIVE_DIN:
            return "input width %d > %d" % (Input.shape[-1], self._NATIVE_DIN)
        if self.mu_kernel.shape[1] > self._NATIVE_DOUT:
            return "output width %d > %d" % (self.mu_kernel.shape[1], self._NATIVE_DOUT)
        if Input.dtype != torch.float32:
            return "dtype %s (float32 only)" % Input.dtype
        return None

    def transform(self, Input):
        """reference MLP.py:48-68; returns (mu, cov): cov = None without the covariance head, else exp(sigma_layer(hidden)) + 1e-6
        (diag_cov; MLP.py:58-61).

        Tensors in HBM go through native kernels only: the fused one-hidden-layer pair psvo_rows_mlp_* where it applies
        (one launch forward, one backward), otherwise one psvo_dense_* launch per Dense layer (any number of hidden layers,
        widths up to 4096) -- there is no PyTorch-op fallback on the GPU.  CPU tensors take the plain
        matmul form below; that is the host-side mirror the CPU tests of the host logic (encoder wiring, k-step
        prediction, R-square) run on a machine without a GPU, never part of the GPU path."""
        if Input.is_cuda:
            if Input.dtype != torch.float32:
                raise ValueError("%s: the native kernels compute in float32, got %s (no PyTorch fallback exists)"
                                 % (self.name, Input.dtype))
            X = Input.reshape(-1, Input.shape[-1])
            if self._native_refusal(Input) is not None:
                # several hidden layers / wide layers: one native Dense launch per layer (f32 MFMA GEMM)
                from ..autograd import DenseFunction
                hidden = X
                for W, b in zip(self.kernels, self.biases):
                    hidden = DenseFunction.apply(hidden, W, b, True)
                mu = DenseFunction.apply(hidden, self.mu_kernel, self.mu_bias, False)
                if self.use_residual:
                    mu = mu + X
                cov = None
                if self.output_cov:
                    cov = torch.exp(DenseFunction.apply(hidden, self.sigma_kernel, self.sigma_bias, False)) + 1e-6
                    cov = cov.reshape(Input.shape[:-1] + (cov.shape[-1],))
                return mu.reshape(Input.shape[:-1] + (mu.shape[-1],)), cov
            from ..autograd import RowsMLPFunction
            mu = RowsMLPFunction.apply(self.__dict__.get("_flat_grad"), X, self.kernels[0], self.biases[0],
                                       self.mu_kernel, self.mu_bias)
            return mu.reshape(Input.shape[:-1] + (mu.shape[-1],)), None
        hidden = Input
        for W, b in zip(self.kernels, self.biases):
            hidden = torch.relu(hidden @ W + b)
        mu = hidden @ self.mu_kernel + self.mu_bias
        if self.use_residual:
            mu = mu + Input
        cov = None
        if self.output_cov:
            cov = torch.exp(hidden @ self.sigma_kernel + self.sigma_bias) + 1e-6
        return mu, cov

    def hip_params(self):
        """the per-particle form the persistent kernels take: (W1, b1, W2, b2) for one hidden layer,
        (W1, b1, W2, b2, Wh, bh) for two (hidden_0, mu_layer, then hidden_1: psvo_mlp in include/psvo_hip.h)."""
        if len(self.Dhs) not in (1, 2) or self.use_residual:
            raise ValueError("%s: the fused HIP kernels take one or two hidden layers per particle MLP and no residual "
                             "connection, got layers %s, use_residual=%s (no fallback path exists)"
                             % (self.name, self.Dhs, self.use_residual))
        p = (self.kernels[0], self.biases[0], self.mu_kernel, self.mu_bias)
        if len(self.Dhs) == 2:
            p = p + (self.kernels[1], self.biases[1])
        return p

    def hip_params_cov(self):
        """the per-particle form of the state-dependent-scale kernels (psvo_filter_forward_cov): (W1, b1, W_mu, b_mu, W_sigma,
        b_sigma), one hidden layer; the kernels take the two heads as ONE output layer [mu_layer | sigma_layer]."""
        if not self.output_cov or len(self.Dhs) != 1 or self.use_residual:
            raise ValueError("%s: the state-dependent-scale kernels take one hidden layer, two heads and no residual "
                             "connection, got layers %s, output_cov=%s, use_residual=%s (no fallback path exists)"
                             % (self.name, self.Dhs, self.output_cov, self.use_residual))
        return (self.kernels[0], self.biases[0], self.mu_kernel, self.mu_bias, self.sigma_kernel, self.sigma_bias)

    def get_variables(self):
        """reference MLP.py:70-86."""
        res = {}
        for i, (W, b) in enumerate(zip(self.kernels, self.biases)):
            res["hidden_{}/weights".format(i)] = W
            res["hidden_{}/bias".format(i)] = b
        res["mu_layer/weights"] = self.mu_kernel
        res["mu_layer/bias"] = self.mu_bias
        if self.output_cov:
            res["sigma_layer/weights"] = self.sigma_kernel
            res["sigma_layer/bias"] = self.sigma_bias
        return res
