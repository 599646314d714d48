"""load_data -- same contract as the reference's loader (src/utils/data_loader.py:5-45): a numpy pickle with
observation splits under Ytrain / Yvalid / Ytest and, optionally, latents under Xtrue or Xtrain / Xtest, is turned
into (hidden_train, hidden_test, obs_train, obs_test).

Rules kept from the reference (cited lines are its file):
  * the held-out observations are the LARGER of Ytest / Yvalid when both exist (:12-14), else whichever exists,
    else ValueError (:15-21);
  * 2-D observation arrays (n, T) get a trailing feature axis (:23-25);
  * latents: Xtrue is split at n_train (:31-33); else Xtrain / Xtest (:34-36); else zeros of shape (n, T, Dx) --
    unless the proposal is centred on the true latents (q_uses_true_X), which then is an error (:37-42).
"""
import pickle

import numpy as np

_HELD_OUT_KEYS = ("Ytest", "Yvalid")


def _read_pickle(path, python2):
    kw = {"encoding": "latin1"} if python2 else {}
    with open(path, "rb") as fh:
        return pickle.load(fh, **kw)


def _held_out_observations(data):
    found = [data[k] for k in _HELD_OUT_KEYS if k in data]
    if not found:
        raise ValueError("obs test set is not found")
    # Ytest wins only when strictly larger than Yvalid
    return found[0] if len(found) == 1 or found[0].shape[0] > found[1].shape[0] else found[1]


def _with_feature_axis(a):
    return a[:, :, None] if a.ndim == 2 else a


def _latents(data, n_train, n_test, T, Dx, required):
    if "Xtrue" in data:
        X = data["Xtrue"]
        return X[:n_train], X[n_train:]
    if "Xtrain" in data and "Xtest" in data:
        return data["Xtrain"], data["Xtest"]
    if required:
        raise ValueError("hidden train and hidden test is not found")
    return np.zeros((n_train, T, Dx)), np.zeros((n_test, T, Dx))


def load_data(path, Dx, isPython2, q_uses_true_X):
    data = _read_pickle(path, isPython2)
    obs_train = data["Ytrain"]
    obs_test = _held_out_observations(data)
    if obs_train.ndim == 2:                      # (the reference keys the reshape of BOTH splits on the training array)
        obs_train, obs_test = _with_feature_axis(obs_train), _with_feature_axis(obs_test)
    n_train, T = obs_train.shape[:2]
    hidden_train, hidden_test = _latents(data, n_train, obs_test.shape[0], T, Dx, required=q_uses_true_X)
    return hidden_train, hidden_test, obs_train, obs_test
