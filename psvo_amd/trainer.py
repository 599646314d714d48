"""trainer -- the reference's training driver (src/trainer.py:21-395) for the MI355X path: same public surface
(constructor, init_data_saving, train, evaluate, evaluate_R_square, adjust_lr, saving_feed_dict, the metric /
trajectory pickles and their keys) and the same schedule semantics, written for this stack.

TensorFlow graph/session plumbing has no equivalent: `fetches` are names ("log_ZSMC", "Xs", "y_hat", "y") instead of
graph tensors and feed dicts are keyed by model.obs / model.hidden ("obs" / "hidden").  Matplotlib output (quiver,
R-square plots) is presentation only and not built.

Data parallelism (new relative to the reference): with W ranks every mini-batch of
FLAGS.batch_size sequences is sharded FLAGS.batch_size / W per rank; one flat all-reduce of the gradient per step,
identical Adam update on every rank (checked bit for bit every print_freq epochs); evaluation batches are dealt
round-robin over the ranks and gathered.
"""
import math
import os
import pickle
import time

import numpy as np
import torch

from . import autograd, dp
from .optim import FlatParams, TFAdam


class StopTraining(Exception):
    pass


def _shuffle(*arrays):
    """sklearn.utils.shuffle(obs, hidden) of the reference (trainer.py:146): one shared permutation
    drawn from numpy's global RNG (seeded by FLAGS.seed, runner.py:42)."""
    perm = np.random.permutation(len(arrays[0]))
    return tuple(a[perm] for a in arrays)


def _r_square(y_hat, y):
    """1 - SS_res / SS_tot with the total sum of squares taken around the mean over the evaluation set (axis 0), per
    time step and feature -- the reference's definition (trainer.py:325-329)."""
    resid = y_hat - y
    centred = y - y.mean(axis=0, keepdims=True)
    return 1.0 - float(np.square(resid).sum()) / float(np.square(centred).sum())


def _collate(per_batch, keepdims):
    """values of ONE fetch over the evaluation batches -> one array (or list of arrays for a list-valued fetch)"""
    first = per_batch[0]
    if isinstance(first, np.ndarray):
        return np.stack(per_batch) if keepdims else np.concatenate(per_batch)
    if isinstance(first, list):                       # k-step predictions: one array per horizon
        return [np.concatenate(col) for col in zip(*per_batch)]
    return np.asarray(per_batch)                      # scalars (log_ZSMC of each batch)


class trainer:
    # schedule state, reference names (trainer.py:63-80): bestCost = index of the best validation ELBO so far,
    # *_count = evaluations since it improved
    def __init__(self, model, SMC, FLAGS):
        self.model, self.SMC, self.FLAGS = model, SMC, FLAGS
        for name in ("Dx", "Dy", "time", "n_particles", "MSE_steps"):
            setattr(self, name, getattr(FLAGS, name))
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
        F = self.FLAGS
        self.batch_size, self.lr, self.epoch = F.batch_size, F.lr, F.epoch
        self.early_stop_patience = F.early_stop_patience
        self.lr_reduce_factor, self.lr_reduce_patience, self.min_lr = F.lr_reduce_factor, F.lr_reduce_patience, F.min_lr
        self.bestCost = self.early_stop_count = self.lr_reduce_count = 0

    def init_data_saving(self, RLT_DIR):
        F = self.FLAGS
        self.save_res, self.RLT_DIR = True, RLT_DIR
        self.save_trajectory, self.save_y_hat, self.saving_num = F.save_trajectory, F.save_y_hat, F.saving_num
        self.save_tensorboard, self.save_model = F.save_tensorboard, F.save_model
        self.log_ZSMC_trains, self.log_ZSMC_tests, self.R_square_trains, self.R_square_tests = [], [], [], []
        # per-epoch artefacts live in a sibling tree: .../rslts/<...> -> .../rslts/epoch_data/<...> (trainer.py:93-95)
        parts = RLT_DIR.split("/")
        at = parts.index("rslts") + 1
        self.epoch_data_DIR = "/".join(parts[:at] + ["epoch_data"] + parts[at:])

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
            with autograd.deferred_join():
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
                    with autograd.deferred_join():
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
        check_exchange = getattr(self.SMC, "check_exchange", None)   # (PSVOwR: sticky flag of the kernels' bounded polls)

        for i in range(self.epoch):
            start = time.time()
            if i == 0:
                self.evaluate_and_save_metrics(i)

            obs_train, hidden_train = _shuffle(obs_train, hidden_train)
            for j in range(0, len(obs_train), self.batch_size):
                self.train_step(obs_train[j:j + self.batch_size], hidden_train[j:j + self.batch_size], self.lr)
            if check_exchange is not None:
                check_exchange()                                 # every epoch: a timed-out exchange means garbage gradients

            if (i + 1) % print_freq == 0:
                if not dp.replicas_in_sync(self.flat.flat):
                    raise RuntimeError("data-parallel replicas diverged (parameters differ between ranks)")
                try:
                    self.evaluate_and_save_metrics(i)
                    self.adjust_lr(i, print_freq)
                except StopTraining:
                    break
                if self.save_res:
                    self._save_epoch_samples(i + 1, obs_test, hidden_test, verbose)
            if verbose:
                print("epoch {:<4} took {:.3f} seconds".format(i + 1, time.time() - start))

        if verbose:
            print("finished training...")
        metrics = {}
        if self.save_res:
            metrics = {k: getattr(self, k) for k in ("log_ZSMC_trains", "log_ZSMC_tests", "R_square_trains", "R_square_tests")}
        return metrics, log

    def _save_epoch_samples(self, epoch, obs_test, hidden_test, write):
        """trajectory_<epoch>.p / y_hat_<epoch>.p for the first saving_num held-out sequences (trainer.py:170-186) and, for
        two-dimensional latents, lattice_val_<epoch>.p -- the data behind the reference's quiver plot (trainer.py:337-361),
        which its notebook reads back (notebooks/PSVO.ipynb: `lattice_dict["X_trajs"], ["X"], ["nextX"]`)."""
        self.saving_feed_dict = {self.obs: obs_test[:self.saving_num], self.hidden: hidden_test[:self.saving_num]}
        Xs_val = None
        for flag, fetch, stem in ((self.save_trajectory, "Xs", "trajectory"), (self.save_y_hat, "y_hat", "y_hat")):
            if not flag:
                continue
            val = self.evaluate(fetch, self.saving_feed_dict, average=False)      # (collective: every rank takes part)
            if fetch == "Xs":
                Xs_val = val
            if write:
                with open(self.epoch_data_DIR + "{}_{}.p".format(stem, epoch), "wb") as f:
                    pickle.dump({fetch: val}, f)
        if self.Dx == 2 and Xs_val is not None:      # (the reference draws from the trajectories it has just saved)
            lattice = self.quiver_lattice(Xs_val)
            if write:
                with open(self.epoch_data_DIR + "lattice_val_{}.p".format(epoch), "wb") as f:
                    pickle.dump(lattice, f)

    def quiver_lattice(self, Xs_val, shape=(25, 25), margin=0.05):
        """{"X_trajs", "X", "nextX"} of trainer.draw_2D_quiver_plot (trainer.py:337-361) without the figure: the particle-mean
        trajectories (saving_num, T, 2), a 25 x 25 lattice over their bounding box and f.mean on it (SMC.get_nextX).  The
        reference takes the box from the axes of its plot of those trajectories, i.e. the data range widened by matplotlib's
        default 5 % margin on either side; that is restated here."""
        X_trajs = np.mean(np.asarray(Xs_val), axis=2)[:self.saving_num]
        lo, hi = X_trajs.reshape(-1, 2).min(axis=0), X_trajs.reshape(-1, 2).max(axis=0)
        pad = margin * (hi - lo)
        x1 = np.linspace(lo[0] - pad[0], hi[0] + pad[0], num=shape[0])
        x2 = np.linspace(lo[1] - pad[1], hi[1] + pad[1], num=shape[1])
        X = np.stack(np.meshgrid(x1, x2), axis=-1)                         # (25, 25, 2), trainer.define2Dlattice
        with torch.no_grad():
            nextX = self.SMC.get_nextX(self._to_dev(X.reshape(-1, 2))).reshape(shape[1], shape[0], 2).cpu().numpy()
        return {"X_trajs": X_trajs, "X": X, "nextX": nextX}

    def close_session(self):
        """the reference closes its tf.Session here (trainer.py:197-198).  Here: release the captured hipGraphs of the training
        step at a quiescent point -- after a device synchronisation, by the caller -- instead of whenever the cycle collector
        reaches this object, which may be in the middle of another trainer's replays (runner.main calls it on its way out)."""
        graphs = self.__dict__.pop("_graphs", None)
        if graphs:
            torch.cuda.synchronize()
            graphs.clear()

    def evaluate_and_save_metrics(self, iter_num, y_hat_N_BxTxDy=None, y_N_BxTxDy=None):
        """ELBO and k-step R-square on the training and held-out sets; appended to the histories and pickled as
        metric_<iter>.p with the reference's keys (trainer.py:201-241).  A non-finite training ELBO stops training."""
        fetches = ["log_ZSMC", "y_hat", "y"]
        m = {}
        for split, obs, hid in (("train", self.obs_train, self.hidden_train), ("test", self.obs_test, self.hidden_test)):
            elbos, y_hat, y = self.evaluate(fetches, {self.obs: obs, self.hidden: hid})
            m["log_ZSMC_" + split] = np.mean(elbos)
            m["R_square_" + split] = self.evaluate_R_square(y_hat, y)

        if dp.rank() == 0:
            print()
            print("iter", iter_num + 1)
            print("Train log_ZSMC: {:>7.3f}, valid log_ZSMC: {:>7.3f}".format(m["log_ZSMC_train"], m["log_ZSMC_test"]))
            print("Train, Valid k-step Rsq:\n", m["R_square_train"], "\n", m["R_square_test"])

        if not math.isfinite(m["log_ZSMC_train"]):
            print("Nan in log_ZSMC, stop training")
            raise StopTraining()

        if self.save_res:
            for key, val in m.items():
                getattr(self, key + "s").append(val)
            if dp.rank() == 0:
                os.makedirs(self.epoch_data_DIR, exist_ok=True)
                with open(self.epoch_data_DIR + "metric_{}.p".format(iter_num + 1), "wb") as f:
                    pickle.dump(dict(m), f)
        return m["log_ZSMC_train"], m["log_ZSMC_test"], m["R_square_train"], m["R_square_test"]

    def adjust_lr(self, iter_num, print_freq):
        """Plateau schedule on the held-out ELBO history (trainer.py:244-270): whenever the best evaluation moves, both
        patience counters restart; while the latest evaluation is not the best one, each counter advances, training stops
        when early_stop_count * print_freq hits early_stop_patience exactly, and the learning rate is multiplied by
        lr_reduce_factor (floored at min_lr) each time lr_reduce_count * print_freq hits lr_reduce_patience."""
        history = self.log_ZSMC_tests
        best, latest = int(np.argmax(history)), len(history) - 1
        if best != self.bestCost:
            self.bestCost, self.early_stop_count, self.lr_reduce_count = best, 0, 0
        talk = dp.rank() == 0
        if talk:
            print("best valid cost on iter: {}\n".format(self.bestCost * print_freq))

        if best != latest:
            self.early_stop_count += 1
            self.lr_reduce_count += 1
            if self.early_stop_count * print_freq == self.early_stop_patience:
                if talk:
                    print("valid cost not improving. stopping training...")
                raise StopTraining()
            if self.lr_reduce_count * print_freq == self.lr_reduce_patience:
                self.lr_reduce_count = 0
                self.lr = max(self.lr * self.lr_reduce_factor, self.min_lr)
                if talk:
                    print("valid cost not improving. reduce learning rate to {}".format(self.lr))
        elif self.save_model and talk:
            os.makedirs(self.RLT_DIR + "model/", exist_ok=True)
            print("Test log_ZSMC improves to {}, save model".format(history[-1]))
            torch.save(self.model.state_dict(), self.RLT_DIR + "model/model_epoch_{}.pt".format(iter_num + 1))

    # ------------------------------------------------------------------------------------------
    def _run(self, names, obs, hidden):
        """one forward evaluation of a batch (sess.run(fetches) of the reference)"""
        with torch.no_grad():
            obs_d = self._to_dev(obs)
            log_ZSMC, log = self.SMC.get_log_ZSMC(obs_d, self._to_dev(hidden))
            out = {"log_ZSMC": float(log_ZSMC)}
            if any(n in names for n in ("Xs", "y_hat", "y")):
                Xs = log["Xs"]
                out["Xs"] = Xs.contiguous().cpu().numpy()
                if "y_hat" in names or "y" in names:
                    y_hat, y = self.SMC.n_step_prediction(self.MSE_steps, Xs, obs_d)
                    out["y_hat"] = [v.cpu().numpy() for v in y_hat]
                    out["y"] = [v.cpu().numpy() for v in y]
        return [out[n] for n in names]

    def evaluate(self, fetches, feed_dict_w_batches={}, average=False, keepdims=False):
        """Evaluate `fetches` (a name or a list of names) over feed_dict_w_batches in mini-batches of batch_size (a short
        last batch is evaluated as it is) and join the results: arrays are concatenated over the batch axis (stacked with
        keepdims), list-valued fetches element-wise, scalars become a vector with one entry per batch; `average` then takes
        the mean over the leading axis.  Semantics of the reference's trainer.evaluate (trainer.py:272-320).
        With W ranks the batches are dealt round-robin (batch k to rank k mod W) and gathered, so every rank returns the
        full result."""
        single = not isinstance(fetches, list)
        names = [fetches] if single else list(fetches)
        obs_all, hid_all = feed_dict_w_batches[self.obs], feed_dict_w_batches[self.hidden]
        n = len(obs_all)
        assert n >= self.batch_size
        starts = list(range(0, n, self.batch_size))
        W, r = dp.world_size(), dp.rank()
        mine = {k: self._run(names, obs_all[s:s + self.batch_size], hid_all[s:s + self.batch_size])
                for k, s in enumerate(starts) if k % W == r}
        if W > 1:
            import torch.distributed as dist
            parts = [None] * W
            dist.all_gather_object(parts, mine)
            mine = {k: v for part in parts for k, v in part.items()}
        rows = [mine[k] for k in range(len(starts))]

        res = [_collate([row[i] for row in rows], keepdims) for i in range(len(names))]
        if average:
            res = [[np.mean(a, axis=0) for a in x] if isinstance(x, list) else np.mean(x, axis=0) for x in res]
        return res[0] if single else res

    def evaluate_R_square(self, y_hat, y):
        """R-square of the k-step-ahead predictions, k = 0 .. MSE_steps (trainer.py:322-335): one number per horizon"""
        return np.array([_r_square(np.asarray(p), np.asarray(t)) for p, t in zip(y_hat, y)])
