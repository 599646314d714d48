"""trainer -- mirror of reference src/trainer.py:21-395 (public surface: constructor,
init_data_saving, train, evaluate, evaluate_R_square, adjust_lr, saving_feed_dict).

TensorFlow graph/session plumbing has no equivalent: `fetches` are names ("log_ZSMC", "Xs",
"y_hat", "y") instead of graph tensors and feed dicts are keyed by model.obs / model.hidden
("obs" / "hidden").  Matplotlib output (quiver, R-square plots) is presentation only and not
built; the pickled per-epoch artefacts keep the reference's keys.

Data parallelism (new relative to the reference): with W ranks every
mini-batch of FLAGS.batch_size sequences is sharded FLAGS.batch_size / W per rank; one flat
all-reduce of the gradient per step, identical Adam update on every rank.
"""
import math
import os
import pickle
import time

import numpy as np
import torch

from . import dp
from .optim import FlatParams, TFAdam


class StopTraining(Exception):
    pass


def _shuffle(*arrays):
    """sklearn.utils.shuffle(obs, hidden) of the reference (trainer.py:146): one shared permutation
    drawn from numpy's global RNG (seeded by FLAGS.seed, runner.py:42)."""
    perm = np.random.permutation(len(arrays[0]))
    return tuple(a[perm] for a in arrays)


class trainer:
    def __init__(self, model, SMC, FLAGS):
        self.model = model
        self.SMC = SMC
        self.FLAGS = FLAGS

        self.Dx = self.FLAGS.Dx
        self.Dy = self.FLAGS.Dy
        self.time = self.FLAGS.time
        self.n_particles = self.FLAGS.n_particles

        self.MSE_steps = self.FLAGS.MSE_steps

        self.save_res = False
        self.draw_quiver_during_training = False

        self.init_placeholder()
        self.init_training_param()

        self.device = next(model.parameters()).device
        self.flat = None
        self.optimizer = None

    def init_placeholder(self):
        self.obs = self.model.obs
        self.hidden = self.model.hidden

    def init_training_param(self):
        self.batch_size = self.FLAGS.batch_size
        self.lr = self.FLAGS.lr
        self.epoch = self.FLAGS.epoch

        # early stopping
        self.early_stop_patience = self.FLAGS.early_stop_patience
        self.bestCost = 0
        self.early_stop_count = 0

        # lr auto decreasing
        self.lr_reduce_factor = self.FLAGS.lr_reduce_factor
        self.lr_reduce_patience = self.FLAGS.lr_reduce_patience
        self.min_lr = self.FLAGS.min_lr
        self.lr_reduce_count = 0

    def init_data_saving(self, RLT_DIR):
        self.save_res = True
        self.RLT_DIR = RLT_DIR
        self.save_trajectory = self.FLAGS.save_trajectory
        self.save_y_hat = self.FLAGS.save_y_hat
        self.saving_num = self.FLAGS.saving_num

        # metrics
        self.log_ZSMC_trains = []
        self.log_ZSMC_tests = []
        self.R_square_trains = []
        self.R_square_tests = []

        # epoch data (trajectory, y_hat and quiver lattice)
        epoch_data_DIR = self.RLT_DIR.split("/")
        epoch_data_DIR.insert(epoch_data_DIR.index("rslts") + 1, "epoch_data")
        self.epoch_data_DIR = "/".join(epoch_data_DIR)

        self.save_tensorboard = self.FLAGS.save_tensorboard
        self.save_model = self.FLAGS.save_model

    # ------------------------------------------------------------------------------------------
    def _to_dev(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.device)

    def _local(self, a):
        """this rank's shard of one mini-batch (contiguous, equal shards)"""
        lo, hi = dp.shard(len(a))
        return a[lo:hi]

    def train_step(self, obs_batch, hidden_batch, lr):
        """one sess.run(train_op) of the reference (trainer.py:147-151).

        On the GPU the local part of the step (zero the gradient buffer, objective, reverse pass: ~90 launches on three
        streams) is captured once per batch shape into a hipGraph and replayed with the batch copied into static
        buffers; the gradient all-reduce and Adam (whose step count and learning rate change) stay outside the graph.
        PSVO_HIPGRAPH=0, or a failed capture, issues everything eagerly."""
        obs = self._to_dev(self._local(obs_batch))
        hidden = self._to_dev(self._local(hidden_batch))
        log_ZSMC = self._graphed_local_step(obs, hidden)
        if log_ZSMC is None:
            self.flat.zero_grad()
            log_ZSMC, _ = self.SMC.get_log_ZSMC(obs, hidden)
            log_ZSMC.backward()
        dp.all_reduce_sum_(self.flat.grad)
        self.optimizer.step(lr, world_size=dp.world_size())
        return log_ZSMC.detach()

    def _graphed_local_step(self, obs, hidden):
        """replay (capturing on first use) the hipGraph of the local step for this batch shape; None = run eagerly"""
        import os
        if (not obs.is_cuda or os.environ.get("PSVO_HIPGRAPH", "1") == "0"
                or getattr(self.SMC, "generator", None) is None):
            return None      # (a CPU generator cannot be registered with the graph)
        graphs = self.__dict__.setdefault("_graphs", {})
        key = (tuple(obs.shape), tuple(hidden.shape))
        g = graphs.get(key)
        if g is None:
            try:
                from .graph import GraphedStep
                s_obs, s_hidden = obs.clone(), hidden.clone()

                def local_step():
                    self.flat.zero_grad()
                    z, _ = self.SMC.get_log_ZSMC(s_obs, s_hidden)
                    z.backward()
                    return z.detach()
                g = (GraphedStep(local_step, generators=[self.SMC.generator]), s_obs, s_hidden)
            except Exception as exc:     # fall back to eager issue for this shape from now on
                if dp.rank() == 0:
                    print("hipGraph capture of the training step failed (%s: %s); issuing eagerly"
                          % (type(exc).__name__, str(exc)[:100]))
                torch.cuda.synchronize()
                g = False
            graphs[key] = g
        if g is False:
            return None
        step, s_obs, s_hidden = g
        s_obs.copy_(obs)
        s_hidden.copy_(hidden)
        return step()

    def train(self, obs_train, obs_test, hidden_train, hidden_test, print_freq):
        self.obs_train, self.obs_test = obs_train, obs_test
        self.hidden_train, self.hidden_test = hidden_train, hidden_test

        assert self.batch_size % dp.world_size() == 0, "batch_size must split evenly over the ranks"
        self.flat = FlatParams(self.model)
        dp.broadcast_(self.flat.flat)
        self.optimizer = TFAdam(self.flat)                       # tf.train.AdamOptimizer(lr), trainer.py:117
        log = {"Xs": "Xs", "y_hat": "y_hat"}
        verbose = dp.rank() == 0

        for i in range(self.epoch):
            start = time.time()

            if i == 0:
                self.evaluate_and_save_metrics(i)

            # training
            obs_train, hidden_train = _shuffle(obs_train, hidden_train)
            for j in range(0, len(obs_train), self.batch_size):
                self.train_step(obs_train[j:j + self.batch_size], hidden_train[j:j + self.batch_size], self.lr)

            if (i + 1) % print_freq == 0:
                check = getattr(self.SMC, "check_exchange", None)      # (PSVOwR: the kernels' bounded polls)
                if check is not None:
                    check()
                try:
                    self.evaluate_and_save_metrics(i)
                    self.adjust_lr(i, print_freq)
                except StopTraining:
                    break

                if self.save_res:
                    self.saving_feed_dict = {self.obs: obs_test[0:self.saving_num],
                                             self.hidden: hidden_test[0:self.saving_num]}
                    if verbose and (self.save_trajectory or self.save_y_hat):
                        Xs_val = self.evaluate("Xs", self.saving_feed_dict, average=False)
                        if self.save_trajectory:
                            with open(self.epoch_data_DIR + "trajectory_{}.p".format(i + 1), "wb") as f:
                                pickle.dump({"Xs": Xs_val}, f)
                        if self.save_y_hat:
                            y_hat_val = self.evaluate("y_hat", self.saving_feed_dict, average=False)
                            with open(self.epoch_data_DIR + "y_hat_{}.p".format(i + 1), "wb") as f:
                                pickle.dump({"y_hat": y_hat_val}, f)

            end = time.time()
            if verbose:
                print("epoch {:<4} took {:.3f} seconds".format(i + 1, end - start))

        if verbose:
            print("finished training...")

        metrics = {"log_ZSMC_trains": self.log_ZSMC_trains,
                   "log_ZSMC_tests": self.log_ZSMC_tests,
                   "R_square_trains": self.R_square_trains,
                   "R_square_tests": self.R_square_tests} if self.save_res else {}
        return metrics, log

    def close_session(self):
        pass

    def evaluate_and_save_metrics(self, iter_num, y_hat_N_BxTxDy=None, y_N_BxTxDy=None):
        log_ZSMC_train, y_hat_train, y_train = \
            self.evaluate(["log_ZSMC", "y_hat", "y"], {self.obs: self.obs_train, self.hidden: self.hidden_train})
        log_ZSMC_test, y_hat_test, y_test = \
            self.evaluate(["log_ZSMC", "y_hat", "y"], {self.obs: self.obs_test, self.hidden: self.hidden_test})

        log_ZSMC_train, log_ZSMC_test = np.mean(log_ZSMC_train), np.mean(log_ZSMC_test)
        R_square_train = self.evaluate_R_square(y_hat_train, y_train)
        R_square_test = self.evaluate_R_square(y_hat_test, y_test)

        # every rank must take the same early-stop / lr decisions: rank 0's numbers are authoritative
        if dp.world_size() > 1:
            import torch.distributed as dist
            box = [(log_ZSMC_train, log_ZSMC_test, R_square_train, R_square_test)]
            dist.broadcast_object_list(box, src=0)
            log_ZSMC_train, log_ZSMC_test, R_square_train, R_square_test = box[0]

        if dp.rank() == 0:
            print()
            print("iter", iter_num + 1)
            print("Train log_ZSMC: {:>7.3f}, valid log_ZSMC: {:>7.3f}".format(log_ZSMC_train, log_ZSMC_test))
            print("Train, Valid k-step Rsq:\n", R_square_train, "\n", R_square_test)

        if not math.isfinite(log_ZSMC_train):
            print("Nan in log_ZSMC, stop training")
            raise StopTraining()

        if self.save_res:
            self.log_ZSMC_trains.append(log_ZSMC_train)
            self.log_ZSMC_tests.append(log_ZSMC_test)
            self.R_square_trains.append(R_square_train)
            self.R_square_tests.append(R_square_test)

            if dp.rank() == 0:
                if not os.path.exists(self.epoch_data_DIR):
                    os.makedirs(self.epoch_data_DIR)
                metric_dict = {"log_ZSMC_train": log_ZSMC_train,
                               "log_ZSMC_test": log_ZSMC_test,
                               "R_square_train": R_square_train,
                               "R_square_test": R_square_test}
                with open(self.epoch_data_DIR + "metric_{}.p".format(iter_num + 1), "wb") as f:
                    pickle.dump(metric_dict, f)

        return log_ZSMC_train, log_ZSMC_test, R_square_train, R_square_test

    def adjust_lr(self, iter_num, print_freq):
        # determine whether should decrease lr or even stop training
        if self.bestCost != np.argmax(self.log_ZSMC_tests):
            self.early_stop_count = 0
            self.lr_reduce_count = 0
            self.bestCost = np.argmax(self.log_ZSMC_tests)

        if dp.rank() == 0:
            print("best valid cost on iter: {}\n".format(self.bestCost * print_freq))

        if self.bestCost != len(self.log_ZSMC_tests) - 1:
            self.early_stop_count += 1
            if self.early_stop_count * print_freq == self.early_stop_patience:
                print("valid cost not improving. stopping training...")
                raise StopTraining()

            self.lr_reduce_count += 1
            if self.lr_reduce_count * print_freq == self.lr_reduce_patience:
                self.lr_reduce_count = 0
                self.lr = max(self.lr * self.lr_reduce_factor, self.min_lr)
                print("valid cost not improving. reduce learning rate to {}".format(self.lr))

        if self.save_model and dp.rank() == 0:
            if not os.path.exists(self.RLT_DIR + "model/"):
                os.makedirs(self.RLT_DIR + "model/")
            if self.bestCost == len(self.log_ZSMC_tests) - 1:
                print("Test log_ZSMC improves to {}, save model".format(self.log_ZSMC_tests[-1]))
                torch.save(self.model.state_dict(), self.RLT_DIR + "model/model_epoch_{}.pt".format(iter_num + 1))

    # ------------------------------------------------------------------------------------------
    def _run(self, names, obs, hidden):
        """one forward evaluation of a batch (sess.run(fetches) of the reference)"""
        with torch.no_grad():
            log_ZSMC, log = self.SMC.get_log_ZSMC(self._to_dev(obs), self._to_dev(hidden))
            out = {"log_ZSMC": float(log_ZSMC)}
            if any(n in names for n in ("Xs", "y_hat", "y")):
                Xs = log["Xs"]
                out["Xs"] = Xs.contiguous().cpu().numpy()
                if "y_hat" in names or "y" in names:
                    y_hat, y = self.SMC.n_step_prediction(self.MSE_steps, Xs, self._to_dev(obs))
                    out["y_hat"] = [v.cpu().numpy() for v in y_hat]
                    out["y"] = [v.cpu().numpy() for v in y]
        return [out[n] for n in names]

    def evaluate(self, fetches, feed_dict_w_batches={}, average=False, keepdims=False):
        """trainer.py:272-320: evaluate `fetches` across the batches of feed_dict_w_batches."""
        single = not isinstance(fetches, list)
        names = [fetches] if single else list(fetches)
        obs_all = feed_dict_w_batches[self.obs]
        hid_all = feed_dict_w_batches[self.hidden]
        n_batches = len(obs_all)
        assert n_batches >= self.batch_size

        fetches_list = []
        for i in range(0, n_batches, self.batch_size):
            fetches_list.append(self._run(names, obs_all[i:i + self.batch_size], hid_all[i:i + self.batch_size]))

        res = []
        for i in range(len(names)):
            if isinstance(fetches_list[0][i], np.ndarray):
                tmp = np.stack([x[i] for x in fetches_list]) if keepdims else np.concatenate([x[i] for x in fetches_list])
            elif isinstance(fetches_list[0][i], list):
                tmp = [np.concatenate([x[i][j] for x in fetches_list]) for j in range(len(fetches_list[0][i]))]
            else:
                tmp = np.array([x[i] for x in fetches_list])
            res.append(tmp)
        if average:
            res = [[np.mean(y, axis=0) for y in x] if isinstance(x, list) else np.mean(x, axis=0) for x in res]
        return res[0] if single else res

    def evaluate_R_square(self, y_hat, y):
        """trainer.py:322-335"""
        n_steps = len(y_hat) - 1

        def get_R_square(y_hat_i, y_i):
            MSE = np.sum((y_hat_i - y_i) ** 2)
            y_i_mean = np.mean(y_i, axis=0, keepdims=True)
            y_i_var = np.sum((y_i - y_i_mean) ** 2)
            return 1 - MSE / y_i_var

        R_square = np.zeros(n_steps + 1)
        for i, (y_hat_i, y_i) in enumerate(zip(y_hat, y)):
            R_square[i] = get_R_square(y_hat_i, y_i)
        return R_square
