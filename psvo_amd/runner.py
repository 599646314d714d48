"""main -- mirror of reference src/runner.py:31-133: seeds, data, SSM + objective switch, result
dir, training, final dumps (history.json, data.p).  Plots are not produced."""
import json
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


def main(FLAGS):
    Dx = FLAGS.Dx
    print_freq = FLAGS.print_freq

    if FLAGS.use_2_q:
        FLAGS.q_uses_true_X = False

    if not torch.cuda.is_available():
        raise RuntimeError("the PSVO hot path runs on MI355X only: no GPU is visible and there is no CPU fallback")
    import os
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, world = dp.init(device=device)

    torch.manual_seed(FLAGS.seed)
    np.random.seed(FLAGS.seed)

    # ============================================= dataset part ============================================= #
    if FLAGS.generateTrainingData:
        hidden_train, hidden_test, obs_train, obs_test = \
            generate_dataset(FLAGS.n_train, FLAGS.n_test, FLAGS.time, model="fhn", Dy=FLAGS.Dy, lb=-2.5, ub=2.5)
    else:
        hidden_train, hidden_test, obs_train, obs_test = \
            load_data(FLAGS.datadir + FLAGS.datadict, Dx, FLAGS.isPython2, FLAGS.q_uses_true_X)
        FLAGS.n_train, FLAGS.n_test, FLAGS.time = obs_train.shape[0], obs_test.shape[0], obs_test.shape[1]

    # clip saving_num to avoid it > n_train or n_test
    FLAGS.MSE_steps = min(FLAGS.MSE_steps, FLAGS.time - 1)
    FLAGS.saving_num = saving_num = min(FLAGS.saving_num, FLAGS.n_train, FLAGS.n_test)
    if rank == 0:
        print("finished preparing dataset")

    # ============================================== model part ============================================== #
    SSM_model = SSM(FLAGS).to(device)

    # at most one of them can be set to True (runner.py:67 -- PSVOwR is missing from the reference's assert)
    assert FLAGS.PSVO + FLAGS.PSVOwR + FLAGS.SVO + FLAGS.AESMC + FLAGS.IWAE < 2

    if FLAGS.PSVO:
        SMC_train = PSVO(SSM_model, FLAGS)
    elif FLAGS.PSVOwR:
        SMC_train = PSVOwR(SSM_model, FLAGS)
    elif FLAGS.SVO:
        SMC_train = SVO(SSM_model, FLAGS)
    elif FLAGS.AESMC:
        SMC_train = AESMC(SSM_model, FLAGS)
    elif FLAGS.IWAE:
        SMC_train = IWAE(SSM_model, FLAGS)
    else:
        raise ValueError("Choose one of objectives among: PSVO, SVO, AESMC, IWAE")
    SMC_train.generator = torch.Generator(device=device).manual_seed(FLAGS.seed + 1000 * rank)

    # =========================================== data saving part =========================================== #
    Experiment_params = {"np": FLAGS.n_particles, "t": FLAGS.time, "bs": FLAGS.batch_size, "lr": FLAGS.lr,
                         "epoch": FLAGS.epoch, "seed": FLAGS.seed, "rslt_dir_name": FLAGS.rslt_dir_name}
    RLT_DIR = create_RLT_DIR(Experiment_params)
    if rank == 0:
        save_experiment_param(RLT_DIR, FLAGS)
        print("RLT_DIR:", RLT_DIR)

    # ============================================= training part ============================================ #
    mytrainer = trainer(SSM_model, SMC_train, FLAGS)
    mytrainer.init_data_saving(RLT_DIR)

    history, log = mytrainer.train(obs_train, obs_test, hidden_train, hidden_test, print_freq)

    # ======================================== final data saving part ======================================== #
    if rank == 0:
        with open(RLT_DIR + "history.json", "w") as f:
            json.dump(history, f, indent=4, cls=NumpyEncoder)

        Xs, y_hat = log["Xs"], log["y_hat"]
        feed = getattr(mytrainer, "saving_feed_dict", {mytrainer.obs: obs_test[0:saving_num],
                                                       mytrainer.hidden: hidden_test[0:saving_num]})
        Xs_val = mytrainer.evaluate(Xs, feed)
        y_hat_val = mytrainer.evaluate(y_hat, feed)
        print("finish evaluating training results")

        testing_data_dict = {"hidden_test": hidden_test[0:saving_num], "obs_test": obs_test[0:saving_num]}
        learned_model_dict = {"Xs_val": Xs_val, "y_hat_val": y_hat_val}
        data_dict = {"testing_data_dict": testing_data_dict, "learned_model_dict": learned_model_dict}
        with open(RLT_DIR + "data.p", "wb") as f:
            pickle.dump(data_dict, f)
    return history
