"""tf_mvn -- mirror of reference src/distribution/mvn.py:23-117, diagonal branch.

mean = MLP(Input); scale = max(softplus(sigma_con), sigma_min) with sigma_con a trainable
state-independent vector (mvn.py:80-90), plus 0.1 * cov(Input) when the transformation has the diagonal
covariance head (output_cov and diag_cov, mvn.py:66-71).  The per-particle evaluations run inside the HIP
kernels; this module serves the hoisted per-(b, t) calls and the k-step prediction.
"""
import math

import torch
from torch import nn

LOG2PI = math.log(2.0 * math.pi)


class MultivariateNormalDiag:
    """The slice of tfd.MultivariateNormalDiag the path uses (TFP 0.5, third-party)."""

    def __init__(self, loc, scale_diag):
        self.loc, self.scale_diag = loc, scale_diag

    def mean(self):
        return self.loc

    def stddev(self):
        return self.scale_diag

    def sample(self, sample_shape=(), eps=None, generator=None):
        if isinstance(sample_shape, int):
            sample_shape = (sample_shape,)
        shape = tuple(sample_shape) + tuple(self.loc.shape)
        if eps is None:
            eps = torch.randn(shape, device=self.loc.device, dtype=self.loc.dtype, generator=generator)
        return self.loc + self.scale_diag * eps

    def log_prob(self, x):
        z = (x - self.loc) / self.scale_diag
        D = x.shape[-1]
        return -0.5 * (z * z).sum(-1) - torch.log(self.scale_diag).sum(-1) - 0.5 * D * LOG2PI


class tf_mvn(nn.Module):
    def __init__(self, transformation, sigma_init=5, sigma_min=1, name="tf_mvn"):
        super().__init__()
        self.transformation = transformation
        self.sigma_init, self.sigma_min = sigma_init, sigma_min
        self.name = name
        # one `sigma_con` per distribution name (AUTO_REUSE, mvn.py:52,82-86)
        self.sigma_con = nn.Parameter(torch.full((transformation.Dout,), float(sigma_init)))

    def get_sigma(self, mu=None):
        s = torch.nn.functional.softplus(self.sigma_con)
        s = torch.where(torch.isnan(s), torch.zeros_like(s), s)
        return torch.clamp(s, min=float(self.sigma_min))

    def mean_and_sigma(self, Input, sigma_con=None):
        """(mean, scale) of the distribution given Input (get_mvn_from_transformation, mvn.py:51-71, diagonal branches);
        sigma_con: this distribution's clipped scale vector when the caller already has it (SSM.sigmas())."""
        mu, cov = self.transformation.transform(Input)
        sigma = self.get_sigma(mu) if sigma_con is None else sigma_con
        if cov is not None:
            assert cov.shape == mu.shape, "diagonal covariance head only"
            sigma = sigma + 0.1 * cov
        return mu, sigma

    def get_mvn(self, Input):
        mu, sigma = self.mean_and_sigma(Input)
        return MultivariateNormalDiag(mu, sigma)

    def sample_and_log_prob(self, Input, sample_shape=(), name=None, eps=None):
        mvn = self.get_mvn(Input)
        sample = mvn.sample(sample_shape, eps=eps)
        return sample, mvn.log_prob(sample)

    def log_prob(self, Input, output, name=None):
        return self.get_mvn(Input).log_prob(output)

    def mean(self, Input, name=None):
        # (mvn.py:104-117 builds the distribution and takes its mean: the scale plays no part, so its four small
        #  kernels are not launched -- the hoisted networks call this several times per step)
        mu, _ = self.transformation.transform(Input)
        return mu
