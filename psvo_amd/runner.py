"""runner.main(FLAGS) -- one experiment from flags to result files: the entry the reference's CLI calls
(src/runner_flag.py:287 -> src/runner.py:31-133), for the MI355X path.

What a caller of the reference relies on, and gets here:
  * flag side effects (src/runner.py:38-39,55-59): `q_uses_true_X` is switched off under `use_2_q`; when data is read from a
    file, `n_train / n_test / time` are overwritten by the data's shape; `MSE_steps <= time - 1`; `saving_num <= n_train, n_test`;
  * seeds: FLAGS.seed seeds numpy (shuffling, data generation) and torch (parameter init) -- src/runner.py:41-42;
  * exactly one objective out of PSVO / PSVOwR / SVO / AESMC / IWAE (PSVO first, like the reference's if-chain; none: ValueError);
  * files: rslts/<rslt_dir_name>/D<stamp>_np_.._t_.._bs_.._lr_.._epoch_.._seed_../{param.json, history.json, data.p} with the
    reference's keys (history: the four metric lists; data.p: testing_data_dict{hidden_test, obs_test},
    learned_model_dict{Xs_val, y_hat_val}); the reference's figures are not produced;
  * return value: the history dict (the reference returns nothing; tf.app.run discards it).
New relative to the reference: one process per GPU (LOCAL_RANK), the batch of sequences sharded over the ranks (psvo_amd.dp),
rank 0 writes the files.
"""
import json
import os
import pickle

import numpy as np
import torch

from . import dp
from .model import SSM
from .rslts_saving.rslts_saving import NumpyEncoder, create_RLT_DIR, save_experiment_param
from .SMC.AESMC import AESMC
from .SMC.IWAE import IWAE
from .SMC.PSVO import PSVO
from .SMC.PSVOwR import PSVOwR
from .SMC.SVO import SVO
from .trainer import trainer
from .utils.data_generator import generate_dataset
from .utils.data_loader import load_data

# flag name -> objective class, in the reference's order of precedence (src/runner.py:70-81)
OBJECTIVES = (("PSVO", PSVO), ("PSVOwR", PSVOwR), ("SVO", SVO), ("AESMC", AESMC), ("IWAE", IWAE))


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("the PSVO hot path runs on MI355X only: no GPU is visible and there is no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    return torch.device("cuda", local_rank)


def _dataset(FLAGS):
    """(hidden_train, hidden_test, obs_train, obs_test); updates the flags that describe the data"""
    if FLAGS.generateTrainingData:
        data = generate_dataset(FLAGS.n_train, FLAGS.n_test, FLAGS.time, model="fhn", Dy=FLAGS.Dy, lb=-2.5, ub=2.5)
    else:
        data = load_data(FLAGS.datadir + FLAGS.datadict, FLAGS.Dx, FLAGS.isPython2, FLAGS.q_uses_true_X)
        obs_train, obs_test = data[2], data[3]
        FLAGS.n_train, FLAGS.n_test, FLAGS.time = obs_train.shape[0], obs_test.shape[0], obs_test.shape[1]
    FLAGS.MSE_steps = min(FLAGS.MSE_steps, FLAGS.time - 1)
    FLAGS.saving_num = min(FLAGS.saving_num, FLAGS.n_train, FLAGS.n_test)
    return data


def _objective(model, FLAGS):
    chosen = [name for name, _ in OBJECTIVES if getattr(FLAGS, name)]
    # (src/runner.py:67 asserts this over four of the five flags; PSVOwR is missing from the reference's assert)
    assert len(chosen) < 2, "at most one objective flag may be set: %s" % chosen
    if not chosen:
        raise ValueError("Choose one of objectives among: PSVO, SVO, AESMC, IWAE")
    return dict(OBJECTIVES)[chosen[0]](model, FLAGS)


def main(FLAGS):
    if FLAGS.use_2_q:
        FLAGS.q_uses_true_X = False
    device = _device()
    rank, _ = dp.init(device=device)
    torch.manual_seed(FLAGS.seed)
    np.random.seed(FLAGS.seed)

    hidden_train, hidden_test, obs_train, obs_test = _dataset(FLAGS)
    if rank == 0:
        print("finished preparing dataset")

    model = SSM(FLAGS).to(device)
    smc = _objective(model, FLAGS)
    smc.generator = torch.Generator(device=device).manual_seed(FLAGS.seed + 1000 * rank)    # this rank's noise stream

    run_dir = create_RLT_DIR({"np": FLAGS.n_particles, "t": FLAGS.time, "bs": FLAGS.batch_size, "lr": FLAGS.lr,
                              "epoch": FLAGS.epoch, "seed": FLAGS.seed, "rslt_dir_name": FLAGS.rslt_dir_name})
    if rank == 0:
        save_experiment_param(run_dir, FLAGS)
        print("RLT_DIR:", run_dir)

    fit = trainer(model, smc, FLAGS)
    fit.init_data_saving(run_dir)
    history, log = fit.train(obs_train, obs_test, hidden_train, hidden_test, FLAGS.print_freq)

    # final evaluation of the first saving_num held-out sequences (collective: every rank takes part), then rank 0 writes
    keep = FLAGS.saving_num
    feed = getattr(fit, "saving_feed_dict", None) or {fit.obs: obs_test[:keep], fit.hidden: hidden_test[:keep]}
    Xs_val = fit.evaluate(log["Xs"], feed)
    y_hat_val = fit.evaluate(log["y_hat"], feed)
    if rank == 0:
        print("finish evaluating training results")
        with open(run_dir + "history.json", "w") as fh:
            json.dump(history, fh, indent=4, cls=NumpyEncoder)
        with open(run_dir + "data.p", "wb") as fh:
            pickle.dump({"testing_data_dict": {"hidden_test": hidden_test[:keep], "obs_test": obs_test[:keep]},
                         "learned_model_dict": {"Xs_val": Xs_val, "y_hat_val": y_hat_val}}, fh)
    fit.close_session()
    return history
