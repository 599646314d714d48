"""Flag registry with the reference's names, types and defaults (src/runner_flag.py:22-284).

The reference uses tf.app.flags (absl); absl is not part of this stack, so this is a small
argparse-free shim accepting the same command-line forms: --name=value, --name value,
--boolname, --noboolname, --boolname=true|false.
"""
import math
import os

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, type, default, help) -- order and defaults follow src/runner_flag.py:166-284
DEFINITIONS = [
    # training hyperparameters (runner_flag.py:22-28)
    ("Dx", int, 2, "dimension of hidden states"),
    ("Dy", int, 1, "dimension of observations"),
    ("n_particles", int, 16, "number of particles"),
    ("batch_size", int, 1, "batch size"),
    ("lr", float, 3e-3, "learning rate"),
    ("epoch", int, 200, "number of epoch"),
    ("seed", int, 2, "random seed"),
    # data (runner_flag.py:33-48)
    ("generateTrainingData", bool, False, "True: generate data set from simulation; False: read from file"),
    ("datadir", str, os.path.join(_REPO, "data/fhn/[1,0]_obs_cov_0.01/"), "path of the data set directory"),
    ("datadict", str, "datadict", "name of the data set file"),
    ("isPython2", bool, False, "was the data pickled in python 2?"),
    ("time", int, 5, "number of timesteps for simulated data"),
    ("n_train", int, 2, "number of trajectories for training set"),
    ("n_test", int, 2, "number of trajectories for testing set"),
    # networks (runner_flag.py:53-83)
    ("q0_layers", str, "32", "architecture for q0 network, int separated by comma"),
    ("q1_layers", str, "32", "architecture for q1 network"),
    ("q2_layers", str, "32", "architecture for q2 network"),
    ("f_layers", str, "32", "architecture for f network"),
    ("g_layers", str, "32", "architecture for g network"),
    ("q0_sigma_init", float, 5, "initial value of q0_sigma"),
    ("q1_sigma_init", float, 5, "initial value of q1_sigma"),
    ("q2_sigma_init", float, 5, "initial value of q2_sigma"),
    ("f_sigma_init", float, 5, "initial value of f_sigma"),
    ("g_sigma_init", float, 5, "initial value of g_sigma"),
    ("q0_sigma_min", float, 1, "minimal value of q0_sigma"),
    ("q1_sigma_min", float, 1, "minimal value of q1_sigma"),
    ("q2_sigma_min", float, 1, "minimal value of q2_sigma"),
    ("f_sigma_min", float, 1, "minimal value of f_sigma"),
    ("g_sigma_min", float, 1, "minimal value of g_sigma"),
    ("output_cov", bool, False, "whether q, f and g networks also output covariance (sigma)"),
    ("diag_cov", bool, False, "whether the networks only output diagonal value of cov matrix"),
    ("y_smoother_Dhs", str, "32", "number of units for y_smoother bidirectional RNNs"),
    ("X0_smoother_Dhs", str, "32", "number of units for X0_smoother bidirectional RNNs"),
    ("X0_use_separate_RNN", bool, True, "whether use a separate RNN for getting X0"),
    ("use_stack_rnn", bool, True, "stack_bidirectional_dynamic_rnn vs bidirectional_dynamic_rnn"),
    # state space model (runner_flag.py:88-98)
    ("use_bootstrap", bool, True, "whether q1 and f share the same network"),
    ("q_uses_true_X", bool, False, "whether q1 uses true hidden states to sample"),
    ("use_2_q", bool, True, "whether q uses two networks q1(x_t|x_t-1) and q2(x_t|y_t)"),
    ("poisson_emission", bool, False, "whether emission uses Poisson distribution"),
    # inference schemes (runner_flag.py:103-113)
    ("PSVO", bool, True, "Particle Smoothing Variational Objective (FFBSim)"),
    ("PSVOwR", bool, False, "Particle Smoothing Variational Objective with Resampling"),
    ("SVO", bool, False, "Smoothing Variational Objective (proposal based on bRNN)"),
    ("AESMC", bool, False, "Auto-Encoding Sequential Monte Carlo"),
    ("IWAE", bool, False, "Importance Weighted Auto-Encoder"),
    ("n_particles_for_BSim_proposal", int, 16, "sub-particles per trajectory in backward simulation proposal"),
    ("BSim_use_single_RNN", bool, False, "backward simulation proposal uses unidirectional RNN"),
    # training (runner_flag.py:118-127)
    ("early_stop_patience", int, 200, "stop training early if validation set does not improve"),
    ("lr_reduce_patience", int, 30, "reduce learning rate when testing loss doesn't improve"),
    ("lr_reduce_factor", float, 1 / math.sqrt(2), "new_lr = old_lr * lr_reduce_factor"),
    ("min_lr", float, 3e-3 / 10, "minimum learning rate"),
    # printing and data saving (runner_flag.py:131-153)
    ("print_freq", int, 1, "frequency to evaluate testing loss & other metrics and save results"),
    ("save_trajectory", bool, True, "whether to save hidden trajectories during training"),
    ("save_y_hat", bool, True, "whether to save k-step y-hat during training"),
    ("rslt_dir_name", str, "test_FFBSim", "dir to save all results"),
    ("MSE_steps", int, 30, "number of steps to predict y-hat and calculate R_square"),
    ("saving_num", int, 30, "number of testing data used to save hidden trajectories, y-hat, ..."),
    ("save_tensorboard", bool, False, "whether to save tensorboard"),
    ("save_model", bool, False, "whether to save model"),
]


class Flags(object):
    """Attribute bag like tf.app.flags.FLAGS; attributes may be overwritten (runner.py:55-59)."""

    def __init__(self, **overrides):
        for name, _, default, _ in DEFINITIONS:
            setattr(self, name, default)
        for k, v in overrides.items():
            if k not in _TYPES:
                raise ValueError("unknown flag --%s" % k)
            setattr(self, k, v)

    def as_dict(self):
        return {name: getattr(self, name) for name, _, _, _ in DEFINITIONS}


_TYPES = {name: typ for name, typ, _, _ in DEFINITIONS}


def _parse_bool(s):
    s = s.strip().lower()
    if s in ("1", "true", "t", "yes", "y"):
        return True
    if s in ("0", "false", "f", "no", "n"):
        return False
    raise ValueError("not a boolean: %r" % s)


def parse_flags(argv):
    """Parse absl-style arguments into a Flags object; unknown flags raise ValueError."""
    FLAGS = Flags()
    i = 0
    argv = list(argv)
    while i < len(argv):
        arg = argv[i]
        i += 1
        if not arg.startswith("-"):
            raise ValueError("unexpected positional argument %r" % arg)
        body = arg.lstrip("-")
        if "=" in body:
            name, val = body.split("=", 1)
        else:
            name, val = body, None
        if name not in _TYPES and name.startswith("no") and _TYPES.get(name[2:]) is bool and val is None:
            setattr(FLAGS, name[2:], False)
            continue
        if name not in _TYPES:
            raise ValueError("unknown flag --%s" % name)
        typ = _TYPES[name]
        if typ is bool:
            setattr(FLAGS, name, True if val is None else _parse_bool(val))
            continue
        if val is None:
            if i >= len(argv):
                raise ValueError("flag --%s needs a value" % name)
            val = argv[i]
            i += 1
        setattr(FLAGS, name, typ(val))
    return FLAGS
