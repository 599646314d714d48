"""Test-side glue between the oracle (reference layout (T, N, B, D), fp64) and the product
(HBM layout (T, B, D, N), fp32).  Only tests import this."""
import torch

from oracle import psvo_oracle as O
from psvo_amd.flags import Flags


def make_flags(objective, **kw):
    base = dict(PSVO=False, SVO=False, AESMC=False, IWAE=False, PSVOwR=False)
    base[objective] = True
    base.update(kw)
    return Flags(**base)


def oracle_flags(FLAGS, objective):
    return dict(Dx=FLAGS.Dx, Dy=FLAGS.Dy, n_particles=FLAGS.n_particles,
                n_particles_for_BSim_proposal=FLAGS.n_particles_for_BSim_proposal,
                use_bootstrap=FLAGS.use_bootstrap, use_2_q=FLAGS.use_2_q, objective=objective,
                use_stack_rnn=FLAGS.use_stack_rnn, BSim_use_single_RNN=FLAGS.BSim_use_single_RNN,
                poisson_emission=FLAGS.poisson_emission)


def noise_to_hip(noise, device):
    out = {}
    f32 = lambda t: t.to(torch.float32).contiguous().to(device)
    if "eps_f" in noise:
        out["eps_f"] = f32(noise["eps_f"].permute(0, 2, 3, 1))
    if "u_f" in noise:
        out["u_f"] = f32(noise["u_f"].permute(0, 2, 1))
    if noise.get("idx_f") is not None:
        out["idx_f"] = noise["idx_f"].permute(0, 2, 1).to(torch.int32).contiguous().to(device)
    if "eps_b" in noise:
        out["eps_b"] = f32(noise["eps_b"].permute(0, 3, 4, 2, 1))
    if "u_b" in noise:
        out["u_b"] = f32(noise["u_b"].permute(0, 2, 1))
    if noise.get("idx_b") is not None:
        out["sel_b"] = noise["idx_b"].permute(0, 2, 1).to(torch.int32).contiguous().to(device)
    if "u_r" in noise:
        out["u_r"] = f32(noise["u_r"].permute(0, 2, 1))
    if noise.get("idx_r") is not None:
        out["anc_r"] = noise["idx_r"].permute(0, 2, 1).to(torch.int32).contiguous().to(device)
    return out


def part_to_ref(t):
    """(T, B, Dx, N) -> (T, N, B, Dx)"""
    return t.permute(0, 3, 1, 2).double().cpu()


def w_to_ref(t):
    """(T, B, N) -> (T, N, B)"""
    return t.permute(0, 2, 1).double().cpu()


def perturb_(model, seed=7, scale=0.3):
    """Give biases and sigmas non-trivial values so that every term of the arithmetic is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("bias") or "biases" in name:
                p.copy_(torch.randn(p.shape, generator=g) * scale)
            if name.endswith("sigma_bias"):     # covariance head (output_cov): keep it around its Constant(1.0) initialiser
                p.add_(1.0)
            if name.endswith("sigma_kernel"):   # ... and its he_normal kernel small: scale = con + 0.1 exp(h W + b) feeds the
                p.mul_(0.25)                    # next state, and without a q2 term a random model runs away to inf in 2 steps
            if name.endswith("sigma_con"):
                p.copy_(0.5 + 1.5 * torch.rand(p.shape, generator=g))
    return model


def run_oracle(model, FLAGS, objective, obs, noise, teacher=None):
    P = model.export_reference_layout(torch.float64)
    o = O.OBJECTIVES[objective](P, oracle_flags(FLAGS, objective))
    nz = dict(noise)
    if teacher:
        nz.update(teacher)
    with torch.no_grad():
        z, log = o.get_log_ZSMC(obs.double().cpu(), nz)
    return z, log


def fill_from_npz(prefix, x, npz):
    """Replace every tensor of the nested structure `x` by the array stored under its path in `npz`
    (the inverse of tests/golden/make_golden.flatten)."""
    if torch.is_tensor(x):
        return torch.as_tensor(npz[prefix])
    if isinstance(x, dict):
        return {k: fill_from_npz(prefix + "." + str(k), v, npz) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(fill_from_npz(prefix + "." + str(i), v, npz) for i, v in enumerate(x))
    return x


def hip_indices(log, objective):
    """the indices the HIP run drew, in the oracle's (T, N, B) layout, keyed as the oracle's teacher-forcing inputs"""
    out = {}
    if log["filter"].get("idx") is not None:
        out["idx_f"] = log["filter"]["idx"].permute(0, 2, 1).cpu().long()
    if objective in ("PSVO", "PSVOwR"):
        out["idx_b"] = log["bsim"]["sel"].permute(0, 2, 1).cpu().long()
    if objective == "PSVOwR":
        out["idx_r"] = log["bsim"]["anc"].permute(0, 2, 1).cpu().long()
    return out


def replay_with_hip_indices(model, FLAGS, objective, obs, noise, log):
    """Run the fp64 oracle with the HIP run's own indices teacher-forced (uniforms still supplied) and measure, for every
    categorical draw of the run, how far the index the kernel took is from the oracle's inverse-CDF draw on the oracle's
    logits (oracle.draw_distance: 0 = the oracle's own draw; a few ulp = the uniform sits on a CDF edge).
    Returns (z_ref, ref_log, stats) with stats = dict(draws, off, worst): number of draws, number with a nonzero
    distance (the genuinely flipped indices), the largest distance (fraction of the total weight)."""
    P = model.export_reference_layout(torch.float64)
    o = O.OBJECTIVES[objective](P, oracle_flags(FLAGS, objective))
    o.draw_log = []
    nz = dict(noise)
    nz.update(hip_indices(log, objective))
    with torch.no_grad():
        z, ref = o.get_log_ZSMC(obs.double().cpu(), nz)
    draws = off = 0
    worst = 0.0
    for log_W, u, idx in o.draw_log:
        d = O.draw_distance(log_W, u, idx)
        draws += d.numel()
        off += int((d > 0).sum())
        worst = max(worst, float(d.max()))
    return z, ref, {"draws": draws, "off": off, "worst": worst}
