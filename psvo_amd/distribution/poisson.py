"""tf_poisson -- mirror of reference src/distribution/poisson.py:27-50 (emission distribution only).

Despite its name the reference builds `tfd.MultivariateNormalDiag(lambdas)` with
`lambdas = softplus(MLP_g(x)) + 1e-6` and NO scale argument (poisson.py:33-38), i.e. a unit-scale
normal around a positive mean.  That is what is reproduced here and inside the HIP kernels
(`psvo_desc.emission = 1`, include/psvo_hip.h); there is no trainable scale vector.
"""
import torch
from torch import nn

from .mvn import MultivariateNormalDiag


class tf_poisson(nn.Module):
    def __init__(self, transformation, name="tf_poisson"):
        super().__init__()
        self.transformation = transformation
        self.name = name

    def get_sigma(self, mu=None):
        """the implicit identity scale (what the kernels receive as sig_g)"""
        p = next(self.transformation.parameters())
        return torch.ones(self.transformation.Dout, device=p.device, dtype=p.dtype)

    def get_poisson(self, Input):
        lambdas, _ = self.transformation.transform(Input)
        lambdas = torch.nn.functional.softplus(lambdas) + 1e-6
        return MultivariateNormalDiag(lambdas, torch.ones_like(lambdas))

    def log_prob(self, Input, output, name=None):
        return self.get_poisson(Input).log_prob(output)

    def mean(self, Input, name=None):
        lambdas, _ = self.transformation.transform(Input)
        return torch.nn.functional.softplus(lambdas) + 1e-6
