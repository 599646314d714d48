"""generate_dataset -- mirror of reference src/utils/data_generator.py:12-85 for the two simulators
the runner can reach ("fhn", "lorenz"): latent ODE integrated with scipy.integrate.odeint over one
step of length dt (reference src/transformation/fhn.py:26-35, lorenz.py), Gaussian emission
y = N(g_mat x, g_cov) (reference src/distribution/mvn.py:10-19)."""
import numpy as np
from scipy.integrate import odeint


def _fhn_step(X_prev, params):
    a, b, c, I, dt = params

    def rhs(X, t):
        V, w = X
        return [V - V ** 3 / 3 - w + I, a * (b * V - c * w)]
    return odeint(rhs, X_prev, np.arange(0, 2 * dt, dt))[1, :]


def _lorenz_step(X_prev, params):
    sigma, rho, beta, dt = params

    def rhs(X, t):
        x, y, z = X
        return [sigma * (y - x), x * (rho - z) - y, x * y - beta * z]
    return odeint(rhs, X_prev, np.arange(0, 2 * dt, dt))[1, :]


def generate_hidden_obs(time, Dx, Dy, x_0, f, g):
    """x_t = f(x_{t-1}), y_t = g(x_t) (data_generator.py:12-26)"""
    X = np.zeros((time, Dx))
    Y = np.zeros((time, Dy))
    X[0] = x_0
    Y[0] = g(x_0)
    for t in range(1, time):
        X[t] = f(X[t - 1])
        Y[t] = g(X[t])
    return X, Y


def generate_dataset(n_train, n_test, time, model="lorenz", Dy=1, Di=1, f=None, g=None, x_0_in=None,
                     lb=-2.5, ub=2.5):
    if model == "fhn":
        Dx = 2
        if f is None:
            f_params = (1.0, 0.95, 0.05, 1.0, 0.15)
            f = lambda x: _fhn_step(x, f_params)
        if g is None:
            g_params = np.array([[1.0, 0.0]])
            g_cov = 0.01 * np.eye(Dy)
    elif model == "lorenz":
        Dx = 3
        if f is None:
            f_params = (10.0, 28.0, 8.0 / 3.0, 0.01)
            f = lambda x: _lorenz_step(x, f_params)
        if g is None:
            g_params = np.array([[1.0, 0.0, 0.0]])
            g_cov = 0.4 * np.eye(Dy)
    else:
        raise ValueError("Unknown model {}".format(model))

    if g is None:
        chol = np.linalg.cholesky(g_cov)
        g = lambda x: np.dot(g_params, x) + np.dot(chol, np.random.randn(Dy))

    hidden_train, obs_train = np.zeros((n_train, time, Dx)), np.zeros((n_train, time, Dy))
    hidden_test, obs_test = np.zeros((n_test, time, Dx)), np.zeros((n_test, time, Dy))

    if x_0_in is None and (lb and ub) is None:
        assert False, "must specify x_0 or (lb and ub)"

    for i in range(n_train + n_test):
        x_0 = np.random.uniform(low=lb, high=ub, size=Dx) if x_0_in is None else x_0_in
        hidden, obs = generate_hidden_obs(time, Dx, Dy, x_0, f, g)
        if i < n_train:
            hidden_train[i], obs_train[i] = hidden, obs
        else:
            hidden_test[i - n_train], obs_test[i - n_train] = hidden, obs

    return hidden_train, hidden_test, obs_train, obs_test
