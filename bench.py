#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the PSVO hot path on MI355X.

A "step" is ONE TRAINING STEP of the objective on one batch of synthetic Fitzhugh-Nagumo sequences
already resident in HBM: observation encoder, hoisted proposal means, random draws, forward
particle filter, backward simulation (the N x N term), ELBO, the full hand-written reverse pass,
the flat-gradient all-reduce (RCCL, N > 1) and the fused Adam update -- i.e. one
`sess.run(train_op)` of the reference (src/trainer.py:147-151).  The forward-only rate (one
`sess.run(log_ZSMC)`) is reported beside it.

Workload = BASELINE.json's target configuration "C*": PSVO (the reference's backward-simulation
objective, BASELINE "SVO"), batch = 32 sequences per GPU, T = 200, N = 128, Dx = 2, M = 16,
H = 32, Dh = 32.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one rank per GPU with torch.distributed.run; the batch of sequences
is sharded (32 per rank: weak scaling) and the only collective is the gradient all-reduce.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: objective, B per GPU, T, N, Dx, Dy, M, H, Dh
    "C*": ("PSVO", 32, 200, 128, 2, 1, 16, 32, 32),
    "C2": ("AESMC", 16, 200, 64, 2, 1, 16, 32, 32),
    "C3": ("PSVO", 32, 400, 128, 3, 1, 16, 32, 32),
    "C4": ("PSVO", 32, 200, 256, 2, 1, 16, 32, 32),
    "C5": ("PSVO", 8, 1000, 512, 4, 1, 16, 32, 32),
    "C*wR": ("PSVOwR", 32, 200, 128, 2, 1, 16, 32, 32),     # C* sizes under the PSVOwR objective (not a headline line)
    "tiny": ("PSVO", 4, 12, 32, 2, 1, 8, 32, 8),            # launcher / multi-rank rehearsals in the tests (not a bench line)
    # C* sizes with TWO hidden layers per particle MLP (q1 / f / g / q1_inv: `*_layers="H,H"`, the reference's flag-file example
    # is "64,64", src/runner_flag.py:50-57); optional 10th entry = hidden layers.  Not headline lines.
    "C*-2x32": ("PSVO", 32, 200, 128, 2, 1, 16, 32, 32, 2),
    "C*-2x64": ("PSVO", 32, 200, 128, 2, 1, 16, 64, 32, 2),
    # C* sizes with state-dependent diagonal scales (FLAGS.output_cov and FLAGS.diag_cov, src/runner_flag.py:67-70): every MLP
    # with its sigma_layer head, psvo_*_cov kernels; optional 11th entry = covariance heads.  Not a headline line.
    "C*-cov": ("PSVO", 32, 200, 128, 2, 1, 16, 32, 32, 1, True),
    "C2-cov": ("AESMC", 16, 200, 64, 2, 1, 16, 32, 32, 1, True),
    "C*wR-cov": ("PSVOwR", 32, 200, 128, 2, 1, 16, 32, 32, 1, True),
    "C5-cov": ("PSVO", 8, 1000, 512, 4, 1, 16, 32, 32, 1, True),
}
FP32_PEAK_TFLOPS = 157.3     # MI355X f32 vector peak == f32-input MFMA dense peak (MI355X_MICROARCH.md)
EXP_PEAK = 9.8e12            # transcendental quarter rate, exp/s


def flop_model(Dx, Dy, N, M, H, E, layers=1, cov=False):
    """Algorithmic flop per particle-step of each native kernel (forward figures: SURVEY.md section 8(d);
    backward figures: DESIGN.md section 5).  FMA = 2 flop, exp/log = 1.  `layers` = 2: an H x H layer more per MLP."""
    mlp = lambda i, o: 2 * H * (i + (2 if cov else 1) * o) + (layers - 1) * 2 * H * H      # (cov: two output heads)
    f_filt = mlp(Dx, Dx) + mlp(Dx, Dy) + mlp(E, Dx) / N + 20 * Dx + 6 * Dy + 10
    f_bsim = 2 * mlp(Dx, Dx) + M * (mlp(Dx, Dx) + mlp(Dx, Dy)) + M * N * (3 * Dx + 4) + M * (14 * Dx + 6 * Dy + 12)
    # reverse passes: MLP forward recompute + input-gradient pass (2x), second pair pass (5 Dx + 6 per pair)
    f_filt_b = 2 * (mlp(Dx, Dx) + mlp(Dx, Dy)) + 40 * Dx + 12 * Dy + 20
    f_bsim_b = 2 * mlp(Dx, Dx) + 2 * M * (mlp(Dx, Dx) + mlp(Dx, Dy)) + M * N * (5 * Dx + 6) + M * (20 * Dx + 8 * Dy + 20)
    # the PSVOwR kernels do the same per-item arithmetic as the PSVO ones (plus an O(N) cross-chain draw per step)
    if cov:     # per-particle scales: one fma per pair and dimension more forward, 2 Dx + 1 per-j sums instead of Dx + 1 in reverse
        f_bsim += M * N * 2 * Dx
        f_bsim_b += M * N * 4 * Dx
        return {"psvo_filter_forward_cov": f_filt, "psvo_bsim_forward_cov": f_bsim,
                "psvo_filter_backward_cov": f_filt_b, "psvo_bsim_backward_cov": f_bsim_b,
                "psvo_bsimwr_forward_cov": f_bsim, "psvo_bsimwr_backward_cov": f_bsim_b}, M * N
    return {"psvo_filter_forward": f_filt, "psvo_bsim_forward": f_bsim,
            "psvo_filter_backward": f_filt_b, "psvo_bsim_backward": f_bsim_b,
            "psvo_bsimwr_forward": f_bsim, "psvo_bsimwr_backward": f_bsim_b}, M * N


def fhn_batch(B, T, seed, device):
    """Synthetic FHN observations: RK4 restatement of the reference generator
    (src/transformation/fhn.py:26-35, src/utils/data_generator.py:38-45): (a,b,c,I,dt) =
    (1.0, 0.95, 0.05, 1.0, 0.15), x0 ~ U(-2.5, 2.5)^2, y = N(x_1, 0.01)."""
    g = torch.Generator().manual_seed(seed)
    a, b, c, I, dt = 1.0, 0.95, 0.05, 1.0, 0.15
    x = torch.rand(B, 2, generator=g, dtype=torch.float64) * 5.0 - 2.5

    def rhs(x):
        V, w = x[:, 0], x[:, 1]
        return torch.stack([V - V ** 3 / 3 - w + I, a * (b * V - c * w)], 1)
    xs, h = [x], dt / 4
    for _ in range(T - 1):
        for _ in range(4):
            k1 = rhs(x); k2 = rhs(x + 0.5 * h * k1); k3 = rhs(x + 0.5 * h * k2); k4 = rhs(x + h * k3)
            x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        xs.append(x)
    hidden = torch.stack(xs, 1)
    obs = hidden[:, :, :1] + 0.1 * torch.randn(B, T, 1, generator=g, dtype=torch.float64)
    return hidden.float().to(device), obs.float().to(device)


def build_objective(wl, device, seed=0):
    from psvo_amd.flags import Flags
    from psvo_amd.model import SSM
    from psvo_amd.SMC.AESMC import AESMC
    from psvo_amd.SMC.IWAE import IWAE
    from psvo_amd.SMC.PSVO import PSVO
    from psvo_amd.SMC.SVO import SVO
    obj, B, T, N, Dx, Dy, M, H, Dh = wl[:9]
    from psvo_amd.SMC.PSVOwR import PSVOwR
    flags = dict(PSVO=False, SVO=False, AESMC=False, IWAE=False, PSVOwR=False)
    flags[obj] = True
    hs = str(H)
    hp = ",".join([hs] * (wl[9] if len(wl) > 9 else 1))        # per-particle MLPs (hoisted q0 / q2 keep one layer)
    cov = bool(wl[10]) if len(wl) > 10 else False
    FLAGS = Flags(Dx=Dx, Dy=Dy, n_particles=N, n_particles_for_BSim_proposal=M, batch_size=B, time=T,
                  q0_layers=hs, q1_layers=hp, q2_layers=hs, f_layers=hp, g_layers=hp,
                  y_smoother_Dhs=str(Dh), X0_smoother_Dhs=str(Dh), output_cov=cov, diag_cov=cov, **flags)
    torch.manual_seed(seed)
    model = SSM(FLAGS).to(device)
    smc = {"PSVO": PSVO, "SVO": SVO, "AESMC": AESMC, "IWAE": IWAE, "PSVOwR": PSVOwR}[obj](model, FLAGS)
    return FLAGS, model, smc


def cpu_baseline(wl, P, obs_cpu, sample_T, threads, train):
    """Time the CPU oracle (op-for-op restatement of the reference's TF graph, including the
    materialised (M, N, N, B) tile) on the host cores, on the first `sample_T` time steps."""
    from oracle import psvo_oracle as O
    obj, B, T, N, Dx, Dy, M, H, Dh = wl[:9]
    torch.set_num_threads(threads)
    P = O.params_to(P, torch.float32)
    fl = dict(Dx=Dx, Dy=Dy, n_particles=N, n_particles_for_BSim_proposal=M, use_bootstrap=True, use_2_q=True,
              objective=obj)
    obs_s = obs_cpu[:, :sample_T].contiguous()
    noise = O.make_noise(fl, B, sample_T, seed=99, dtype=torch.float32)
    if train:
        def req(x):
            if torch.is_tensor(x):
                x.requires_grad_(True)
            elif isinstance(x, dict):
                [req(v) for v in x.values()]
            elif isinstance(x, (list, tuple)):
                [req(v) for v in x]
        req(P)
    o = O.OBJECTIVES[obj](P, fl)
    t0 = time.perf_counter()
    if train:
        z, _ = o.get_log_ZSMC(obs_s, noise)
        z.backward()
    else:
        with torch.no_grad():
            o.get_log_ZSMC(obs_s, noise)
    dt = time.perf_counter() - t0
    what = "training step (forward + torch autograd, no optimizer)" if train else "forward evaluation (no_grad)"
    return {"value": B * sample_T * N / dt, "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": "%s in fp32 of the first %d of %d time steps of the same workload, PyTorch-CPU oracle "
                      "(materialises the (M,N,N,B) tile like the reference's TF graph), %.1f s" % (what, sample_T, T, dt)}


def elbo_vs_oracle(wl, P, obs_cpu, sample_T, device, threads):
    """The "ELBO vs ref" half of the metric: the HIP path against the fp64 CPU oracle on identical inputs -- the same
    parameters (the snapshot taken before training), observations and injected noise, first `sample_T` time steps.

    Two comparisons, because a free-running particle system stops being comparable draw by draw after the first resampling
    index that differs (an fp32 CDF against an fp64 one flips an index whenever a uniform lands within rounding of an
    edge; from then on the two runs are different, equally valid, particle systems and their trajectories differ by O(1)):
      * free-running: the kernels draw their own indices; the oracle is then re-run with THOSE indices teacher-forced, every
        value must agree, and every index is checked to be the oracle's own inverse-CDF draw or to sit within
        `worst_edge_distance` (fraction of the total weight) of the CDF edge that separates the two.  `flipped_draws` is
        the number of indices that differ from the oracle's own draw on the same logits.
      * the ELBO of the oracle's own free run is reported beside it (`elbo_oracle_free_run`): it agrees to rounding when no
        index flipped and differs like two independent draws otherwise."""
    from oracle import psvo_oracle as O
    obj, B, T, N, Dx, Dy, M, H, Dh = wl[:9]
    torch.set_num_threads(threads)
    P64 = O.params_to(P, torch.float64)
    fl = dict(Dx=Dx, Dy=Dy, n_particles=N, n_particles_for_BSim_proposal=M, use_bootstrap=True, use_2_q=True,
              objective=obj)
    obs_s = obs_cpu[:, :sample_T].contiguous().double()
    noise = O.make_noise(fl, B, sample_T, seed=7, dtype=torch.float64)
    with torch.no_grad():
        z_free, _ = O.OBJECTIVES[obj](P64, fl).get_log_ZSMC(obs_s, noise)
    FLAGS, model, smc = build_objective(wl, device, seed=0)
    model.load_reference_layout(P)
    perm = {"eps_f": (0, 2, 3, 1), "u_f": (0, 2, 1), "eps_b": (0, 3, 4, 2, 1), "u_b": (0, 2, 1), "u_r": (0, 2, 1)}
    nz = {k: noise[k].permute(*perm[k]).float().contiguous().to(device) for k in perm if k in noise}
    with torch.no_grad():
        z, log = smc.get_log_ZSMC(obs_s.float().to(device), None, noise=nz)
    torch.cuda.synchronize()
    # the oracle with the kernels' own indices teacher-forced, every draw logged
    teach = {}
    if log["filter"].get("idx") is not None:
        teach["idx_f"] = log["filter"]["idx"].permute(0, 2, 1).cpu().long()
    if obj in ("PSVO", "PSVOwR"):
        teach["idx_b"] = log["bsim"]["sel"].permute(0, 2, 1).cpu().long()
    if obj == "PSVOwR":
        teach["idx_r"] = log["bsim"]["anc"].permute(0, 2, 1).cpu().long()
    o = O.OBJECTIVES[obj](P64, fl)
    o.draw_log = []
    with torch.no_grad():
        z_ref, log_ref = o.get_log_ZSMC(obs_s, {**noise, **teach})
    draws = flipped = 0
    worst = 0.0
    for log_W, u, idx in o.draw_log:
        d = O.draw_distance(log_W, u, idx)
        draws += d.numel()
        flipped += int((d > 0).sum())
        worst = max(worst, float(d.max()))
    return {"elbo_hip": float(z), "elbo_oracle": float(z_ref),
            "rel_err": abs(float(z) - float(z_ref)) / abs(float(z_ref)),
            "max_abs_trajectory_err": float((log["Xs"].double().cpu() - log_ref["Xs"]).abs().max()),
            "draws": draws, "flipped_draws": flipped, "worst_edge_distance": worst,
            "elbo_oracle_free_run": float(z_free),
            "sample": "first %d of %d time steps of the workload, identical parameters / observations / injected "
                      "noise, fp64 PyTorch-CPU oracle vs fp32 HIP path; in-kernel multinomial draws, the oracle re-run "
                      "with the kernels' indices (every draw verified against the oracle's CDF)" % (sample_T, T)}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` given as it stands (no torchrun around it): start the N ranks ourselves, one per GPU,
    exactly as the driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py ...`), BEFORE this process has made any GPU call -- it stays a plain launcher: the children
    inherit stdout, so rank 0's JSON line is the only line on it, and their exit code is ours."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def plumbing_only(args):
    """--plumbing-only: the multi-rank skeleton of the bench without the hot path (which needs an MI355X and has no CPU
    fallback): rendezvous, barrier, the flat-gradient all-reduce on a buffer of the model's size, max-over-ranks timing and
    rank 0's single JSON line.  Used by the CPU tests (gloo) to cover launching and relaying; never a bench result."""
    from psvo_amd import dp
    rank, world = dp.init(backend=os.environ.get("PSVO_DIST_BACKEND", "gloo"))
    flat = torch.full((22426,), float(rank + 1))
    dist = torch.distributed if world > 1 else None
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g = flat.clone()
        dp.all_reduce_sum_(g)
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)
    ok = bool((g == world * (world + 1) / 2).all())
    if rank == 0:
        print(json.dumps({"metric": "plumbing-only (no hot path, not a bench result)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / max(1, args.steps) * 1e3,
                          "allreduce_ok": ok}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node; default: WORLD_SIZE when a launcher has set it, else 1")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C*", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="train", choices=["train", "forward"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the step as one captured hipGraph: the launching Python thread needs ~2.7 ms per C* "
                         "step and more than the GPU time of the smaller workloads, and its speed varies with the host.  "
                         "Default (neither flag): both ways are calibrated on a few untimed steps and the faster one "
                         "runs the timed region; eager issue is also the fallback if capture fails")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="issue every launch eagerly")
    ap.add_argument("--cpu-sample-T", type=int, default=40)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--plumbing-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--bsim-bwd-variant", type=int, default=-1, choices=[-1, 0, 1, 2, 3, 4],
                    help="A/B switch of the reverse backward-simulation kernel (psvo_set_tuning, include/psvo_hip.h): "
                         "0 = v1 butterflies, 1 = v2 VALU, 2 = v2 with the per-j sums on f32 MFMA, 3 = v2 with the pair exponents on f32 "
                         "MFMA, -1 = library default")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))      # (no GPU call has been made in this process)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or run `python bench.py --gpus N` "
                         "without a launcher: it starts its own ranks)" % (args.gpus, world))
    if args.plumbing_only:
        raise SystemExit(plumbing_only(args))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the PSVO hot path")
    # one rank per GPU; (rehearsals on a one-GPU box: ranks share the card, PSVO_DIST_BACKEND=gloo)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from psvo_amd import _lib, dp, ops
    from psvo_amd.optim import FlatParams, TFAdam
    if args.bsim_bwd_variant >= 0:
        _lib.check(_lib.load().psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, args.bsim_bwd_variant), "psvo_set_tuning")
    dp.init(backend=os.environ.get("PSVO_DIST_BACKEND", "nccl"), device=device)
    # (PSVO_FORCE_PG=1 creates the group for one rank too: RCCL's init, barrier and all-reduce then run on a one-GPU box)
    dist = torch.distributed if torch.distributed.is_initialized() else None

    wl = WORKLOADS[args.workload]
    obj, B, T, N, Dx, Dy, M, H, Dh = wl[:9]
    layers = wl[9] if len(wl) > 9 else 1
    FLAGS, model, smc = build_objective(wl, device, seed=0)
    P_ref = model.export_reference_layout(torch.float32)     # snapshot for the CPU baseline
    smc.generator = torch.Generator(device=device).manual_seed(1234 + rank)
    hidden, obs = fhn_batch(B, T, seed=100 + rank, device=device)
    flat = FlatParams(model)
    dp.broadcast_(flat.flat)
    opt = TFAdam(flat)
    lr = 3e-3

    # HIP events around every native launch, on the stream they are launched on (torch's current stream)
    events = {}
    marks = []                     # (name, phase, event) in issue order; ("step", 0, e) opens a step
    rec = {"on": False}

    def hook(name, phase, stream):
        if rec["on"]:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            events.setdefault(name, []).append(e)
            marks.append((name, phase, e))

    def mark_step():
        if rec["on"]:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            marks.append(("step", 0, e))

    from psvo_amd import autograd as _ag

    def train_step():
        mark_step()
        flat.zero_grad()
        z, _ = smc.get_log_ZSMC(obs, hidden)
        with _ag.deferred_join():
            z.backward()
        dp.all_reduce_sum_(flat.grad)
        opt.step(lr, world_size=world)
        return z

    def fwd_step():
        mark_step()
        with torch.no_grad():
            z, _ = smc.get_log_ZSMC(obs, hidden)
        return z

    # The step is captured once into a hipGraph and replayed (same kernels, same buffers, fresh random
    # draws per replay).  With more than one rank only the local compute is captured; the gradient
    # all-reduce and the Adam launch stay eager behind it.
    from psvo_amd.graph import GraphedStep

    def local_step():
        flat.zero_grad()
        z, _ = smc.get_log_ZSMC(obs, hidden)
        with _ag.deferred_join():
            z.backward()
        return z.detach()

    def update():
        dp.all_reduce_sum_(flat.grad)
        opt.step(lr, world_size=world)

    use_graph, graph_note = (args.graph is not False), None
    eager_step = train_step if args.mode == "train" else fwd_step
    step = eager_step
    if use_graph:
        try:
            if args.mode == "train":
                # as in psvo_amd.trainer.train_step: the local part of the step is replayed, the gradient all-reduce
                # and Adam (step count, learning rate) stay eager behind it
                g_local = GraphedStep(local_step, generators=[smc.generator])

                def step():
                    z = g_local()
                    update()
                    return z
            else:
                step = GraphedStep(lambda: fwd_step().detach(), generators=[smc.generator])
        except Exception as exc:     # e.g. a kernel that cannot be captured (cooperative launches): issue eagerly
            use_graph, step = False, eager_step
            graph_note = "graph capture failed (%s: %s)" % (type(exc).__name__, str(exc)[:120])
            torch.cuda.synchronize()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, record, ticks=None):
        """`steps` calls of fn between two barrier + device syncs, wall clock.  `ticks` (a list) also receives one HIP event
        per step boundary, recorded on the compute stream without any synchronisation: the spread of the step time."""
        sync()
        rec["on"] = record
        stream = torch.cuda.current_stream()
        t0 = time.perf_counter()
        for _ in range(steps):
            if ticks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                ticks.append(e)
            z = fn()
        if ticks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            ticks.append(e)
        rec["on"] = False
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, z

    if use_graph and args.graph is None:
        # Neither --graph nor --no-graph: replay takes the launching thread out of the loop (decisive on a slow host and
        # for the small workloads), but the hipGraph executor's own scheduling of the step's three branches can cost a few
        # per cent against eager issue from a fast host.  Both are timed on a few untimed steps (max over ranks, so every
        # rank takes the same decision) and the faster one runs the timed region.
        def quick(fn, n=10):
            for _ in range(5):
                fn()
            return timed(fn, n, False)[0] / n
        t_graph, t_eager = quick(step), quick(eager_step)
        if t_eager < 0.985 * t_graph:
            use_graph, step = False, eager_step
            graph_note = "chosen by calibration (%.3f ms against %.3f ms replayed)" % (1e3 * t_eager, 1e3 * t_graph)
    for _ in range(args.warmup):
        z = step()
    # the timed region carries no instrumentation; the per-kernel HIP events (two per native launch) are
    # recorded afterwards on the same step issued eagerly -- they cost ~1 ms per step of host time
    ticks = []
    elapsed, z = timed(step, args.steps, False, ticks)
    step_ms = sorted(ticks[i].elapsed_time(ticks[i + 1]) for i in range(len(ticks) - 1))
    elbo = float(z.detach())
    eager = train_step if args.mode == "train" else fwd_step
    for _ in range(3):      # (after graph replay the eager path first has to populate its own allocator pool)
        eager()
    ev_steps = max(3, min(args.steps, 10))
    ops.set_timing_hook(hook)
    timed(eager, ev_steps, True)
    ops.set_timing_hook(None)
    other = None
    if args.mode == "train":     # forward-only rate beside it (not `value`)
        for _ in range(2):
            fwd_step()
        nf = max(5, args.steps // 2)
        el_f, _ = timed(fwd_step, nf, False)
        other = world * B * T * N * nf / el_f

    if rank == 0:
        units = B * T * N                                    # particle-steps one launch processes
        kms = {}
        for name, evs in events.items():
            d = [evs[i].elapsed_time(evs[i + 1]) for i in range(0, len(evs) - 1, 2)]
            cps = max(1, len(d) // ev_steps)                         # calls per step
            per_step = sorted(sum(d[i * cps:(i + 1) * cps]) for i in range(ev_steps))
            kms[name] = (per_step[len(per_step) // 2], float(cps))   # median ms per step (all calls), calls per step
        # where in the step each native kernel runs: [first start, last end] in ms after the step's first launch
        tl_steps, t0e, per = {}, None, {}
        for name, phase, e in marks + [("step", 0, None)]:
            if name == "step":
                if t0e is not None:
                    for k, (a0, a1) in per.items():
                        tl_steps.setdefault(k, []).append((t0e.elapsed_time(a0), t0e.elapsed_time(a1)))
                t0e, per = e, {}
            elif phase == 0:
                per.setdefault(name, [e, e])
            else:
                per[name][1] = e
        med = lambda v: sorted(v)[len(v) // 2]
        timeline = {k: [round(med([x[0] for x in v]), 3), round(med([x[1] for x in v]), 3)] for k, v in tl_steps.items()}
        timeline = dict(sorted(timeline.items(), key=lambda kv: kv[1][0]))      # (medians over the instrumented steps)
        if os.environ.get("PSVO_BENCH_CALLS"):      # every native call of the last instrumented step, in issue order
            last = max(i for i, m in enumerate(marks) if m[0] == "step")
            t0c, calls, open_ = marks[last][2], [], {}
            for name, phase, e in marks[last + 1:]:
                if phase == 0:
                    open_[name] = e
                else:
                    calls.append((name, round(t0c.elapsed_time(open_[name]), 3), round(t0c.elapsed_time(e), 3)))
            print("calls:", calls, file=sys.stderr)
        flops, x_bsim = flop_model(Dx, Dy, N, M, H, Dy, layers, cov=len(wl) > 10 and bool(wl[10]))
        cand = {k: v for k, v in kms.items() if k in flops}
        dominant = max(cand, key=lambda k: cand[k][0])
        k_avg = cand[dominant][0] / max(1.0, cand[dominant][1])
        f_dom = flops[dominant]
        achieved = units * f_dom / (k_avg * 1e-3) / 1e12
        value = world * units * args.steps / elapsed
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the
        # figure is the committed, calibrated FETCH_SIZE / WRITE_SIZE measurement of the same workload
        # (tools/profile_round.sh: traffic_probe.py + traffic_report.py -> profiles/r03_hbm_traffic_Cstar.json), else null
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "r03_hbm_traffic_Cstar.json")
        if args.workload == "C*" and os.path.exists(tp):
            key = {"psvo_bsim_backward": "bsim_bwd", "psvo_bsim_forward": "bsim_fwd_kernel",
                   "psvo_filter_backward": "filter_bwd_kernel", "psvo_filter_forward": "filter_fwd"}.get(dominant, dominant)
            for k, v in json.load(open(tp))["kernels"].items():
                if key in k and "finalize" not in k and v["hbm_MB_per_launch"] * 1e6 > (traffic or 0.0):
                    traffic, traffic_src = v["hbm_MB_per_launch"] * 1e6, "profiles/r03_hbm_traffic_Cstar.json"
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # spread of the timed steps on rank 0 (HIP events on the compute stream at the step boundaries of the SAME
            # timed region; `ms_per_step` / `value` stay wall clock over all steps, max over ranks)
            "step_ms": {"median": step_ms[len(step_ms) // 2], "min": step_ms[0], "max": step_ms[-1], "n": len(step_ms)},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s on Fitzhugh-Nagumo, batch=%d/GPU, T=%d, N=%d, Dx=%d, M=%d, H=%d, Dh=%d"
                                   % (args.workload, obj, B, T, N, Dx, M, H, Dh),
                       "mode": ("training step: forward + reverse pass + gradient all-reduce + Adam"
                                if args.mode == "train" else "objective evaluation (ELBO + smoothed trajectories)"),
                       "global_batch": B * world, "parallelism": "dp%d (batch of sequences sharded)" % world,
                       "elbo": elbo, "forward_only_particle_steps_per_s": other,
                       "launch": ("hipGraph replay" + ("" if args.graph else " (calibrated against eager issue)")) if use_graph
                                 else ("eager" + ("; " + graph_note if graph_note else "")),
                       "bsim_bwd_variant": args.bsim_bwd_variant,
                       "native_ms_per_step": {k: round(v[0], 4) for k, v in sorted(kms.items())},
                       "native_timeline_ms": timeline},
            "roofline": {"bound": "mfma", "pipe": "valu", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch (HBM, rocprofv3 PMC)",
                         "traffic_source": traffic_src,
                         "kernel": dominant, "kernel_ms_avg": k_avg, "flop_per_particle_step": f_dom,
                         "exp_frac": (units * x_bsim / (k_avg * 1e-3) / EXP_PEAK) if "bsim" in dominant else None,
                         "note": "compute-bound, priced against the dense f32 peak of 157.3 TFLOP/s, which on gfx950 is the "
                                 "f32-input MFMA peak AND the packed f32 vector peak (`bound` keeps the contract's label for "
                                 "the compute roofline; `pipe` says which pipe executes it: the dominant kernel's default "
                                 "build issues its flops on the vector ALU -- the f32 MFMA variants were measured beside it "
                                 "and do not co-execute with the VALU, profiles/r02_bsim_bwd_ab.md; it moves ~0.5 TB/s; a stream of "
                                 "nothing but independent f32 FMAs sustains 95 (scalar) -- 112 (packed) TFLOP/s at the two waves "
                                 "per SIMD these kernels run: tools/micro/valu_rate.hip, profiles/r03_valu_rate.txt)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, P_ref, obs.cpu(), min(args.cpu_sample_T, T),
                                               min(args.cpu_threads, os.cpu_count() or 1), args.mode == "train")
            out["elbo_vs_oracle"] = elbo_vs_oracle(wl, P_ref, obs.cpu(), min(20, T), device,
                                                   min(args.cpu_threads, os.cpu_count() or 1))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
