"""Simulated data sets for the runner's `generateTrainingData` switch (reference src/utils/data_generator.py:28-85).

generate_dataset(n_train, n_test, time, model="fhn" | "lorenz", Dy, f=None, g=None, x_0_in=None, lb, ub)
    -> hidden_train (n_train, time, Dx), hidden_test (n_test, time, Dx), obs_train (n_train, time, Dy), obs_test (n_test, time, Dy)

Contract kept from the reference (cited lines: its files)
  * the latent path is deterministic: one step = the ODE integrated by scipy's odeint over [0, dt] from the previous state
    (src/transformation/fhn.py:26-35, lorenz.py) -- FHN (a, b, c, I, dt) = (1, 0.95, 0.05, 1, 0.15), Lorenz
    (sigma, rho, beta, dt) = (10, 28, 8/3, 0.01);
  * the emission is y_t = G x_t + chol(cov) z with z ~ N(0, I_Dy) drawn from numpy's GLOBAL generator
    (src/distribution/mvn.py:10-19): G = [[1, 0]], cov = 0.01 I (FHN); G = [[1, 0, 0]], cov = 0.4 I (Lorenz);
  * order of the random draws (it fixes the data for a given np.random.seed): per sequence -- training sequences first -- the
    initial state x_0 ~ U(lb, ub)^Dx, then one emission draw per time step, t = 0 first;
  * `f` / `g` may be supplied: anything with a `.sample(x)` method (the reference's distribution objects) or a plain callable.
"""
import numpy as np
from scipy.integrate import odeint

FHN_PARAMS = (1.0, 0.95, 0.05, 1.0, 0.15)              # a, b, c, I, dt
LORENZ_PARAMS = (10.0, 28.0, 8.0 / 3.0, 0.01)          # sigma, rho, beta, dt


def _advance(rhs, state, dt):
    """the state one step of length dt later (odeint over the two-point grid [0, dt], as the reference does)"""
    return odeint(rhs, state, np.array([0.0, dt]))[1]


def _fhn_step(X_prev, params):
    a, b, c, I, dt = params
    return _advance(lambda X, t: (X[0] - X[0] ** 3 / 3.0 - X[1] + I, a * (b * X[0] - c * X[1])), X_prev, dt)


def _lorenz_step(X_prev, params):
    sigma, rho, beta, dt = params
    return _advance(lambda X, t: (sigma * (X[1] - X[0]), X[0] * (rho - X[2]) - X[1], X[0] * X[1] - beta * X[2]), X_prev, dt)


class _Deterministic:
    """x -> step(x): the reference's dirac_delta around an ODE transformation"""

    def __init__(self, step, params):
        self._step, self._params = step, params

    def sample(self, x):
        return self._step(x, self._params)


class _LinearGaussian:
    """x -> G x + chol(cov) z: the reference's numpy mvn around a linear transformation"""

    def __init__(self, G, cov):
        self._G, self._chol = np.asarray(G, dtype=float), np.linalg.cholesky(cov)

    def sample(self, x):
        mean = self._G @ x
        return mean + self._chol @ np.random.randn(mean.shape[0])


_MODELS = {   # name: (Dx, transition step, its parameters, emission matrix, emission variance)
    "fhn": (2, _fhn_step, FHN_PARAMS, [[1.0, 0.0]], 0.01),
    "lorenz": (3, _lorenz_step, LORENZ_PARAMS, [[1.0, 0.0, 0.0]], 0.4),
}


def _sampler(obj):
    return obj.sample if hasattr(obj, "sample") else obj


def generate_hidden_obs(time, Dx, Dy, x_0, f, g):
    """one sequence: X[0] = x_0, X[t] = f(X[t-1]); Y[t] = g(X[t]) (src/utils/data_generator.py:12-26)"""
    step, emit = _sampler(f), _sampler(g)
    X, Y = np.empty((time, Dx)), np.empty((time, Dy))
    state = np.asarray(x_0, dtype=float)
    for t in range(time):
        if t:
            state = step(state)
        X[t], Y[t] = state, emit(state)
    return X, Y


def generate_dataset(n_train, n_test, time, model="lorenz", Dy=1, Di=1, f=None, g=None, x_0_in=None, lb=-2.5, ub=2.5):
    if model not in _MODELS:
        raise ValueError("Unknown model {}".format(model))
    Dx, step, params, G, var = _MODELS[model]
    f = f if f is not None else _Deterministic(step, params)
    g = g if g is not None else _LinearGaussian(G, var * np.eye(Dy))
    if x_0_in is None and (lb is None or ub is None):
        raise AssertionError("must specify x_0 or (lb and ub)")

    hidden = np.zeros((n_train + n_test, time, Dx))
    obs = np.zeros((n_train + n_test, time, Dy))
    for i in range(n_train + n_test):
        x_0 = np.random.uniform(low=lb, high=ub, size=Dx) if x_0_in is None else x_0_in
        hidden[i], obs[i] = generate_hidden_obs(time, Dx, Dy, x_0, f, g)
    return hidden[:n_train], hidden[n_train:], obs[:n_train], obs[n_train:]
