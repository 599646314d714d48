"""Tensor-level wrappers over the C ABI: argument checking, output allocation, stream plumbing.

All tensors use the library's particle-minor HBM layout (include/psvo_hip.h):
particles (T, B, Dx, N), log-weights (T, B, N), features (T, B, D), bsim noise (T, B, Dx, N, M).
PyTorch is only the allocator / stream provider here.
"""
import ctypes

import torch

from . import _lib


# Optional side stream for the native launches of one wrapper call.  Buffers the wrapper allocates come from
# the SIDE stream's pool: a block of the current stream's pool may have been freed by work that is still
# queued there (the caching allocator recycles within a stream without waiting), and the side stream is not
# ordered after that work.  Tensors that arrive from the current stream are record_stream()'ed when their
# pointer is taken; whoever consumes the outputs on the current stream must order the streams (event /
# wait_stream) and record_stream() them there (autograd.FilterFunction does).
_LAUNCH_STREAM = None


def _empty(*shape, **kw):
    if _LAUNCH_STREAM is not None:
        with torch.cuda.stream(_LAUNCH_STREAM):
            return torch.empty(*shape, **kw)
    return torch.empty(*shape, **kw)


class launch_on(object):
    """with ops.launch_on(stream): native kernels of the enclosed wrapper calls go to `stream`."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        global _LAUNCH_STREAM
        self.prev, _LAUNCH_STREAM = _LAUNCH_STREAM, self.stream
        return self.stream

    def __exit__(self, *exc):
        global _LAUNCH_STREAM
        _LAUNCH_STREAM = self.prev
        return False


class _wgrad_on(object):
    """launch_on(stream), ordered after everything issued so far on the current stream; a no-op for None"""

    def __init__(self, stream):
        self.stream, self.ctx = stream, None

    def __enter__(self):
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            self.ctx = launch_on(self.stream)
            self.ctx.__enter__()
        return self.stream

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def _ptr(t):
    if t is None:
        return None
    if _LAUNCH_STREAM is not None:
        t.record_stream(_LAUNCH_STREAM)
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, shape, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError("%s must live in HBM (cuda tensor); got %s" % (name, t.device))
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def _mlp_struct(p, Din, H, Dout, name):
    """p = (W1 (Din,H), b1 (H), W2 (H,Dout), b2 (Dout)) keras-layout tensors; with two hidden layers
    (W1, b1, W2, b2, Wh (H,H), bh (H)): hidden_0, mu_layer, hidden_1."""
    W1, b1, W2, b2 = p[:4]
    _chk(W1, (Din, H), name + ".W1")
    _chk(b1, (H,), name + ".b1")
    _chk(W2, (H, Dout), name + ".W2")
    _chk(b2, (Dout,), name + ".b2")
    s = _lib.psvo_mlp()
    if len(p) == 6:
        _chk(p[4], (H, H), name + ".Wh")
        _chk(p[5], (H,), name + ".bh")
        s.Wh, s.bh = p[4].data_ptr(), p[5].data_ptr()
    elif len(p) != 4:
        raise ValueError("%s: an MLP is 4 tensors (one hidden layer) or 6 (two), got %d" % (name, len(p)))
    if _LAUNCH_STREAM is not None:
        for t in p:
            t.record_stream(_LAUNCH_STREAM)
    s.W1, s.b1, s.W2, s.b2 = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr()
    return s


def mlp_grad_size(Din, H, Dout, layers=1):
    return Din * H + H + H * Dout + Dout + (H * H + H if layers == 2 else 0)


def make_desc(B, T, N, M, Dx, Dy, H, resample=True, two_q=True, bootstrap=True, emission=0, layers=1):
    d = _lib.psvo_desc()
    d.B, d.T, d.N, d.M, d.Dx, d.Dy, d.H = B, T, N, M, Dx, Dy, H
    d.layers = int(layers)            # hidden layers of every per-particle MLP of the call (1 or 2)
    d.resample, d.two_q, d.bootstrap = int(resample), int(two_q), int(bootstrap)
    d.emission = int(emission)        # 1: tf_poisson emission (unit-scale normal, softplus mean)
    return d


def _cur_stream():
    return _LAUNCH_STREAM if _LAUNCH_STREAM is not None else torch.cuda.current_stream()


def _stream():
    return ctypes.c_void_p(_cur_stream().cuda_stream)


# Optional instrumentation: bench.py installs a hook that records HIP events (on the stream the
# kernels are launched on) right before and after each native launch.  None = no overhead.
_HOOK = None


def set_timing_hook(hook):
    global _HOOK
    _HOOK = hook


def _mark(name, phase):
    if _HOOK is not None:
        _HOOK(name, phase, _cur_stream())


def filter_forward(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0,
                   obs, eps, u=None, idx_in=None):
    """psvo_filter_forward.  Returns dict(X, Xanc, Fm, logW, idx, lse)."""
    lib = _lib.load()
    B, T, N, Dx, Dy, H = desc.B, desc.T, desc.N, desc.Dx, desc.Dy, desc.H
    dev = eps.device
    q1s = _mlp_struct(q1, Dx, H, Dx, "q1")
    fs = None if desc.bootstrap else _mlp_struct(f, Dx, H, Dx, "f")
    gs = _mlp_struct(g, Dx, H, Dy, "g")
    _chk(sig_q1, (Dx,), "sig_q1"); _chk(sig_g, (Dy,), "sig_g")
    if desc.two_q:
        _chk(sig_q2, (Dx,), "sig_q2"); _chk(mu2, (T, B, Dx), "mu2")
    if not desc.bootstrap:
        _chk(sig_f, (Dx,), "sig_f")
    _chk(m0, (B, Dx), "m0"); _chk(sig0, (Dx,), "sig0")
    _chk(fm0, (B, Dx), "fm0"); _chk(fsig0, (Dx,), "fsig0")
    _chk(obs, (T, B, Dy), "obs"); _chk(eps, (T, B, Dx, N), "eps")
    _chk(u, (T, B, N), "u"); _chk(idx_in, (T, B, N), "idx_in", torch.int32)
    if desc.resample and u is None and idx_in is None:
        raise ValueError("resampling needs uniforms `u` or teacher-forced `idx_in`")
    out = {
        "X": _empty(T, B, Dx, N, device=dev), "Xanc": _empty(T, B, Dx, N, device=dev),
        "Fm": _empty(T, B, Dx, N, device=dev), "logW": _empty(T, B, N, device=dev),
        "idx": _empty(T, B, N, device=dev, dtype=torch.int32), "lse": _empty(T, B, device=dev),
        "P1": None if desc.bootstrap else _empty(T, B, Dx, N, device=dev),
    }
    _mark("psvo_filter_forward", 0)
    st = lib.psvo_filter_forward(
        ctypes.byref(desc), ctypes.byref(q1s), ctypes.byref(fs) if fs is not None else None, ctypes.byref(gs),
        _ptr(sig_q1), _ptr(sig_q2), _ptr(sig_f), _ptr(sig_g), _ptr(mu2), _ptr(m0), _ptr(sig0), _ptr(fm0),
        _ptr(fsig0), _ptr(obs), _ptr(eps), _ptr(u), _ptr(idx_in),
        _ptr(out["X"]), _ptr(out["Xanc"]), _ptr(out["Fm"]), _ptr(out["P1"]), _ptr(out["logW"]), _ptr(out["idx"]),
        _ptr(out["lse"]), _stream())
    _mark("psvo_filter_forward", 1)
    _lib.check(st, "psvo_filter_forward")
    return out


def bsim_forward(desc, filt, f, g, q1_inv, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init,
                 imean, isig, obs, eps_b, u_b=None, sel_in=None, save=False):
    """psvo_bsim_forward.  `filt` is the dict returned by filter_forward.
    Returns dict(bwX, flp, glp, Omega, sel, score)."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs = _mlp_struct(f, Dx, H, Dx, "f")
    gs = _mlp_struct(g, Dx, H, Dy, "g")
    qs = _mlp_struct(q1_inv, Dx, H, Dx, "q1_inv")
    for k in ("X", "Fm"):
        _chk(filt[k], (T, B, Dx, N), k)
    _chk(filt["logW"], (T, B, N), "logW"); _chk(filt["lse"], (T, B), "lse")
    _chk(sig_f, (Dx,), "sig_f"); _chk(sig_g, (Dy,), "sig_g")
    _chk(sig_q1inv, (Dx,), "sig_q1inv"); _chk(sig_bq2, (Dx,), "sig_bq2")
    _chk(bmu2, (T, B, Dx), "bmu2"); _chk(minit, (B, Dx), "minit"); _chk(sig_init, (Dx,), "sig_init")
    _chk(imean, (B, Dx), "imean"); _chk(isig, (Dx,), "isig")
    _chk(obs, (T, B, Dy), "obs"); _chk(eps_b, (T, B, Dx, N, M), "eps_b")
    _chk(u_b, (T, B, N), "u_b"); _chk(sel_in, (T, B, N), "sel_in", torch.int32)
    if u_b is None and sel_in is None:
        raise ValueError("backward simulation needs uniforms `u_b` or teacher-forced `sel_in`")
    out = {
        "bwX": _empty(T, B, Dx, N, device=dev), "flp": _empty(T, B, N, device=dev),
        "glp": _empty(T, B, N, device=dev), "Omega": _empty(T, B, N, device=dev),
        "sel": _empty(T, B, N, device=dev, dtype=torch.int32), "score": _empty(B, N, device=dev),
        "lam2": _empty(T, B, N, M, device=dev) if save else None,
        "om": _empty(T, B, N, M, device=dev) if save else None,
        # (row t = T-1 of mu1 is written as zeros by the kernel: there is no predecessor step)
        "mu1": _empty(T, B, Dx, N, device=dev) if save else None,
    }
    _mark("psvo_bsim_forward", 0)
    st = lib.psvo_bsim_forward(
        ctypes.byref(desc), _ptr(filt["X"]), _ptr(filt["Fm"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs),
        _ptr(sig_f), _ptr(sig_g), _ptr(sig_q1inv), _ptr(sig_bq2), _ptr(bmu2), _ptr(minit), _ptr(sig_init),
        _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b), _ptr(u_b), _ptr(sel_in),
        _ptr(out["bwX"]), _ptr(out["flp"]), _ptr(out["glp"]), _ptr(out["Omega"]), _ptr(out["sel"]),
        _ptr(out["score"]), _ptr(out["lam2"]), _ptr(out["om"]), _ptr(out["mu1"]), _stream())
    _mark("psvo_bsim_forward", 1)
    _lib.check(st, "psvo_bsim_forward")
    return out


def elbo_filter(desc, lse):
    lib = _lib.load()
    _chk(lse, (desc.T, desc.B), "lse")
    out = _empty(desc.B, device=lse.device)
    _lib.check(lib.psvo_elbo_filter(ctypes.byref(desc), _ptr(lse), _ptr(out), _stream()), "psvo_elbo_filter")
    return out


def elbo_bsim(desc, score):
    lib = _lib.load()
    _chk(score, (desc.B, desc.N), "score")
    out = _empty(desc.B, device=score.device)
    _lib.check(lib.psvo_elbo_bsim(ctypes.byref(desc), _ptr(score), _ptr(out), _stream()), "psvo_elbo_bsim")
    return out


def elbo_bsim_mean(desc, score):
    """psvo_elbo_bsim_mean: scalar mean_b [logsumexp_n score - log N] (0-d tensor)."""
    lib = _lib.load()
    _chk(score, (desc.B, desc.N), "score")
    out = _empty(1, device=score.device)
    _lib.check(lib.psvo_elbo_bsim_mean(ctypes.byref(desc), _ptr(score), _ptr(out), _stream()), "psvo_elbo_bsim_mean")
    return out.reshape(())


def elbo_bsim_mean_backward(desc, score, dz):
    lib = _lib.load()
    _chk(score, (desc.B, desc.N), "score")
    dz = dz.detach().float().contiguous().reshape(1)
    dscore = _empty(desc.B, desc.N, device=score.device)
    _lib.check(lib.psvo_elbo_bsim_mean_backward(ctypes.byref(desc), _ptr(score), _ptr(dz), _ptr(dscore), _stream()),
               "psvo_elbo_bsim_mean_backward")
    return dscore


def bilstm_forward(x, W_fw, b_fw, W_bw, b_bw, save=False):
    """psvo_bilstm_forward: x (B,T,Din) -> out (B,T,2Dh) [, cs (2,B,T,Dh), gates (2,B,T,4Dh)]."""
    lib = _lib.load()
    B, T, Din = x.shape
    Dh = W_fw.shape[1] // 4
    _chk(x, (B, T, Din), "x")
    for nm, W, b in (("fw", W_fw, b_fw), ("bw", W_bw, b_bw)):
        _chk(W, (Din + Dh, 4 * Dh), "W_" + nm)
        _chk(b, (4 * Dh,), "b_" + nm)
    out = _empty(B, T, 2 * Dh, device=x.device)
    cs = _empty(2, B, T, Dh, device=x.device) if save else None
    gates = _empty(2, B, T, 4 * Dh, device=x.device) if save else None
    _mark("psvo_bilstm_forward", 0)
    st = lib.psvo_bilstm_forward(B, T, Din, Dh, _ptr(x), _ptr(W_fw), _ptr(b_fw), _ptr(W_bw), _ptr(b_bw),
                                 _ptr(out), _ptr(cs), _ptr(gates), _stream())
    _mark("psvo_bilstm_forward", 1)
    _lib.check(st, "psvo_bilstm_forward")
    return (out, cs, gates) if save else out


def mlp_wgrad(X, dOut, w, Din, H, Dout, grad=None, axis=2):
    """psvo_mlp_wgrad / psvo_mlp2_wgrad: rows X [S][Din][L], dOut [S][Dout][L] -> flat grad [dW1|db1|dW2|db2]
    (two hidden layers, len(w) == 6: [dW1|db1|dWh|dbh|dW2|db2], the H x H products on the f32 matrix instruction).
    X / dOut are contiguous tensors whose feature axis is `axis` (dims before it flatten to the
    segments S, dims after it to the rows L of a segment), e.g. (T,B,Din,N) or (T,B,Din,N,M)."""
    lib = _lib.load()
    dev = X.device
    ax = axis
    if (not X.is_contiguous() or not dOut.is_contiguous() or X.shape[ax] != Din or dOut.shape[ax] != Dout
            or X.shape[:ax] != dOut.shape[:ax] or X.shape[ax + 1:] != dOut.shape[ax + 1:]):
        raise ValueError("mlp_wgrad: bad row tensors %s / %s for Din=%d, Dout=%d, axis=%d"
                         % (tuple(X.shape), tuple(dOut.shape), Din, Dout, ax))
    S = 1
    for v in X.shape[:ax]:
        S *= v
    L = 1
    for v in X.shape[ax + 1:]:
        L *= v
    ws = _mlp_struct(w, Din, H, Dout, "w")
    two = len(w) == 6
    name = "psvo_mlp2_wgrad" if two else "psvo_mlp_wgrad"
    NP = mlp_grad_size(Din, H, Dout, 2 if two else 1)
    nblk = (lib.psvo_mlp2_wgrad_blocks if two else lib.psvo_mlp_wgrad_blocks)(S * L)
    partial = _empty(nblk, NP, device=dev)
    acc = grad is not None
    if grad is None:
        grad = _empty(NP, device=dev)
    elif grad.numel() != NP:
        raise ValueError("mlp_wgrad: gradient slice of %d floats for an MLP of %d parameters" % (grad.numel(), NP))
    _mark(name, 0)
    st = getattr(lib, name)(S, L, Din, H, Dout, _ptr(X), _ptr(dOut), ctypes.byref(ws), _ptr(partial), _ptr(grad),
                            int(acc), _stream())
    _mark(name, 1)
    _lib.check(st, name)
    return grad


def rows_mlp_forward(X, w):
    """psvo_rows_mlp_forward: X (R, Din) -> (R, Dout); w = (W1 (Din,H), b1, W2 (H,Dout), b2)."""
    lib = _lib.load()
    R, Din = X.shape
    H, Dout = w[2].shape
    _chk(X, (R, Din), "X")
    ws = _mlp_struct(w, Din, H, Dout, "w")
    out = _empty(R, Dout, device=X.device)
    _mark("psvo_rows_mlp_forward", 0)
    st = lib.psvo_rows_mlp_forward(R, Din, H, Dout, _ptr(X), ctypes.byref(ws), _ptr(out), _stream())
    _mark("psvo_rows_mlp_forward", 1)
    _lib.check(st, "psvo_rows_mlp_forward")
    return out


def rows_mlp_backward(X, dOut, w, need_dX=True, grad=None):
    """psvo_rows_mlp_backward -> (dX (R,Din) or None, flat grad [dW1|db1|dW2|db2]); `grad` given: accumulate."""
    lib = _lib.load()
    R, Din = X.shape
    H, Dout = w[2].shape
    _chk(X, (R, Din), "X"); _chk(dOut, (R, Dout), "dOut")
    ws = _mlp_struct(w, Din, H, Dout, "w")
    NP = Din * H + H + H * Dout + Dout
    dX = _empty(R, Din, device=X.device) if need_dX else None
    partial = _empty(lib.psvo_rows_mlp_blocks(R), NP, device=X.device)
    acc = grad is not None
    if grad is None:
        grad = _empty(NP, device=X.device)
    _mark("psvo_rows_mlp_backward", 0)
    st = lib.psvo_rows_mlp_backward(R, Din, H, Dout, _ptr(X), _ptr(dOut), ctypes.byref(ws), _ptr(dX), _ptr(partial),
                                    _ptr(grad), int(acc), _stream())
    _mark("psvo_rows_mlp_backward", 1)
    _lib.check(st, "psvo_rows_mlp_backward")
    return dX, grad


def dense_forward(X, W, b, relu):
    """psvo_dense_forward: X (R, Din) -> act(X W + b) (R, Dout) on the f32 matrix instruction."""
    lib = _lib.load()
    R, Din = X.shape
    Dout = W.shape[1]
    _chk(X, (R, Din), "X"); _chk(W, (Din, Dout), "W"); _chk(b, (Dout,), "b")
    Y = _empty(R, Dout, device=X.device)
    _mark("psvo_dense_forward", 0)
    st = lib.psvo_dense_forward(R, Din, Dout, _ptr(X), _ptr(W), _ptr(b), int(relu), _ptr(Y), _stream())
    _mark("psvo_dense_forward", 1)
    _lib.check(st, "psvo_dense_forward")
    return Y


def dense_backward(X, Y, dY, W, relu, need_dX=True):
    """psvo_dense_backward -> (dX (R, Din) or None, dW (Din, Dout), db (Dout))."""
    lib = _lib.load()
    R, Din = X.shape
    Dout = W.shape[1]
    _chk(X, (R, Din), "X"); _chk(dY, (R, Dout), "dY"); _chk(W, (Din, Dout), "W")
    if relu:
        _chk(Y, (R, Dout), "Y")
    dX = _empty(R, Din, device=X.device) if need_dX else None
    partial = _empty(lib.psvo_dense_wgrad_slices(R), (Din + 1) * Dout, device=X.device)
    grad = _empty((Din + 1) * Dout, device=X.device)
    _mark("psvo_dense_backward", 0)
    st = lib.psvo_dense_backward(R, Din, Dout, _ptr(X), _ptr(Y) if relu else None, _ptr(dY), _ptr(W), int(relu), _ptr(dX),
                                 _ptr(partial), _ptr(grad), 0, _stream())
    _mark("psvo_dense_backward", 1)
    _lib.check(st, "psvo_dense_backward")
    return dX, grad[:Din * Dout].view(Din, Dout), grad[Din * Dout:]


def split_mlp_grad(g, Din, H, Dout, layers=1):
    """flat [dW1|db1|dW2|db2] -> (dW1 (Din,H), db1 (H), dW2 (H,Dout), db2 (Dout)) views; two hidden layers: flat
    [dW1|db1|dWh|dbh|dW2|db2] -> (dW1, db1, dW2, db2, dWh (H,H), dbh (H)), the order of the 6-tensor MLP tuples."""
    a = Din * H
    if layers == 2:
        b = a + H + H * H + H
        return (g[:a].view(Din, H), g[a:a + H], g[b:b + H * Dout].view(H, Dout), g[b + H * Dout:],
                g[a + H:a + H + H * H].view(H, H), g[a + H + H * H:b])
    return (g[:a].view(Din, H), g[a:a + H], g[a + H:a + H + H * Dout].view(H, Dout), g[a + H + H * Dout:])


def filter_backward(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0, obs, eps, filt,
                    dlse=None, dFm=None, dlogW=None, gbufs=None, before_wgrad=None, defer_wgrad=False):
    """psvo_filter_backward + psvo_mlp_wgrad.  `filt` = forward outputs.  dFm (T,B,Dx,N) / dlogW (T,B,N)
    are upstream gradients (or None).  Returns a dict of gradients."""
    lib = _lib.load()
    B, T, N, Dx, Dy, H = desc.B, desc.T, desc.N, desc.Dx, desc.Dy, desc.H
    dev = eps.device
    q1s = _mlp_struct(q1, Dx, H, Dx, "q1")
    fs = None if desc.bootstrap else _mlp_struct(f, Dx, H, Dx, "f")
    gs = _mlp_struct(g, Dx, H, Dy, "g")
    _chk(dlse, (T, B), "dlse"); _chk(dFm, (T, B, Dx, N), "dFm"); _chk(dlogW, (T, B, N), "dlogW")
    z = lambda *s: _empty(*s, device=dev)
    out = {"dP": z(T, B, Dx, N), "dF": None if desc.bootstrap else z(T, B, Dx, N), "dG": z(T, B, Dy, N),
           "dmu2": z(T, B, Dx) if desc.two_q else None, "dm0": z(B, Dx), "dfm0": z(B, Dx),
           "dsig_q1": z(Dx), "dsig_q2": z(Dx), "dsig_f": z(Dx), "dsig_g": z(Dy), "dsig0": z(Dx), "dfsig0": z(Dx)}
    sacc = z(lib.psvo_filter_ws_floats(B, T, N, Dx, Dy))
    nparts = 1 if (dFm is not None or dlogW is not None) else 0
    _mark("psvo_filter_backward", 0)
    st = lib.psvo_filter_backward(
        ctypes.byref(desc), ctypes.byref(q1s), ctypes.byref(fs) if fs is not None else None, ctypes.byref(gs),
        _ptr(sig_q1), _ptr(sig_q2), _ptr(sig_f), _ptr(sig_g), _ptr(mu2), _ptr(m0), _ptr(sig0), _ptr(fm0), _ptr(fsig0),
        _ptr(obs), _ptr(eps), _ptr(filt["X"]), _ptr(filt["Fm"]), _ptr(filt["P1"]), _ptr(filt["logW"]),
        _ptr(filt["lse"]), _ptr(filt["idx"]), _ptr(dlse), nparts, _ptr(dFm), _ptr(dlogW),
        _ptr(out["dP"]), _ptr(out["dF"]), _ptr(out["dG"]), _ptr(out["dmu2"]), _ptr(out["dm0"]), _ptr(out["dfm0"]),
        _ptr(out["dsig_q1"]), _ptr(out["dsig_q2"]), _ptr(out["dsig_f"]), _ptr(out["dsig_g"]), _ptr(out["dsig0"]),
        _ptr(out["dfsig0"]), _ptr(sacc), _stream())
    _mark("psvo_filter_backward", 1)
    _lib.check(st, "psvo_filter_backward")
    # weight gradients from rows; gbufs = (q1, f, g) slices of the flat gradient buffer to accumulate into
    # directly (then no gradient tensor is returned for that MLP), or None
    gb = gbufs or (None, None, None)

    def wgrad(g_stream=None, after=None):
        """the two or three weight-gradient launches; g_stream: issue MLP_g's on that stream (ordered after the event
        `after`, i.e. after this kernel's rows exist) beside MLP_q1's on the current one -- the caller joins both"""
        out["gq1"] = mlp_wgrad(filt["X"], out["dP"], q1, Dx, H, Dx, grad=gb[0])
        out["gf"] = None if desc.bootstrap else mlp_wgrad(filt["X"], out["dF"], f, Dx, H, Dx, grad=gb[1])
        if g_stream is not None and gb[2] is not None:
            g_stream.wait_event(after)
            for t in (filt["X"], out["dG"]):
                t.record_stream(g_stream)
            with launch_on(g_stream):
                out["gg"] = mlp_wgrad(filt["X"], out["dG"], g, Dx, H, Dy, grad=gb[2])
        else:
            out["gg"] = mlp_wgrad(filt["X"], out["dG"], g, Dx, H, Dy, grad=gb[2])
    if defer_wgrad:          # the caller issues them later (out["_wgrad"]()), e.g. after other writers of the same slices
        out["_wgrad"] = wgrad
        return out
    if before_wgrad is not None:
        before_wgrad()
    wgrad()
    return out


# ---------------------------------------------------------------------------------------------
# state-dependent scales (FLAGS.output_cov and FLAGS.diag_cov): psvo_filter_forward_cov / psvo_filter_backward_cov.
# A per-particle MLP is the 6-tuple (W1, b1, W_mu, b_mu, W_sigma, b_sigma); the kernels take the two heads as one output
# layer, W2 = [W_mu | W_sigma] (H, 2 Dout).
# ---------------------------------------------------------------------------------------------
def _cov_struct(p, Din, H, Dout, name):
    """(psvo_mlp with the concatenated output layer, the tensors that must stay alive until the launch has been issued)"""
    W1, b1, Wm, bm, Ws, bs = p
    _chk(Wm, (H, Dout), name + ".W_mu"); _chk(bm, (Dout,), name + ".b_mu")
    _chk(Ws, (H, Dout), name + ".W_sigma"); _chk(bs, (Dout,), name + ".b_sigma")
    if _LAUNCH_STREAM is not None:
        with torch.cuda.stream(_LAUNCH_STREAM):
            W2, b2 = torch.cat([Wm, Ws], dim=1).contiguous(), torch.cat([bm, bs]).contiguous()
    else:
        W2, b2 = torch.cat([Wm, Ws], dim=1).contiguous(), torch.cat([bm, bs]).contiguous()
    return _mlp_struct((W1, b1, W2, b2), Din, H, 2 * Dout, name), (W2, b2)


def filter_forward_cov(desc, q1, f, g, sigc_q1, sigc_f, sigc_g, mu2, sig2, m0, sig0, fm0, fsig0, obs, eps,
                       u=None, idx_in=None):
    """psvo_filter_forward_cov.  Returns dict(X, Xanc, Fm, Fs, P1, P1s, logW, idx, lse)."""
    lib = _lib.load()
    B, T, N, Dx, Dy, H = desc.B, desc.T, desc.N, desc.Dx, desc.Dy, desc.H
    dev = eps.device
    q1s, k1 = _cov_struct(q1, Dx, H, Dx, "q1")
    fs, k2 = (None, None) if desc.bootstrap else _cov_struct(f, Dx, H, Dx, "f")
    gs, k3 = _cov_struct(g, Dx, H, Dy, "g")
    _chk(sigc_q1, (Dx,), "sigc_q1"); _chk(sigc_g, (Dy,), "sigc_g")
    if desc.two_q:
        _chk(mu2, (T, B, Dx), "mu2"); _chk(sig2, (T, B, Dx), "sig2")
    if not desc.bootstrap:
        _chk(sigc_f, (Dx,), "sigc_f")
    for t, nm in ((m0, "m0"), (sig0, "sig0"), (fm0, "fm0"), (fsig0, "fsig0")):
        _chk(t, (B, Dx), nm)
    _chk(obs, (T, B, Dy), "obs"); _chk(eps, (T, B, Dx, N), "eps")
    _chk(u, (T, B, N), "u"); _chk(idx_in, (T, B, N), "idx_in", torch.int32)
    if desc.resample and u is None and idx_in is None:
        raise ValueError("resampling needs uniforms `u` or teacher-forced `idx_in`")
    z = lambda *s: _empty(*s, device=dev)
    out = {"X": z(T, B, Dx, N), "Xanc": z(T, B, Dx, N), "Fm": z(T, B, Dx, N), "Fs": z(T, B, Dx, N), "logW": z(T, B, N),
           "idx": _empty(T, B, N, device=dev, dtype=torch.int32), "lse": z(T, B),
           "P1": None if desc.bootstrap else z(T, B, Dx, N), "P1s": None if desc.bootstrap else z(T, B, Dx, N)}
    _mark("psvo_filter_forward_cov", 0)
    st = lib.psvo_filter_forward_cov(
        ctypes.byref(desc), ctypes.byref(q1s), ctypes.byref(fs) if fs is not None else None, ctypes.byref(gs),
        _ptr(sigc_q1), _ptr(sigc_f), _ptr(sigc_g), _ptr(mu2), _ptr(sig2), _ptr(m0), _ptr(sig0), _ptr(fm0), _ptr(fsig0),
        _ptr(obs), _ptr(eps), _ptr(u), _ptr(idx_in),
        _ptr(out["X"]), _ptr(out["Xanc"]), _ptr(out["Fm"]), _ptr(out["Fs"]), _ptr(out["P1"]), _ptr(out["P1s"]),
        _ptr(out["logW"]), _ptr(out["idx"]), _ptr(out["lse"]), _stream())
    _mark("psvo_filter_forward_cov", 1)
    _lib.check(st, "psvo_filter_forward_cov")
    del k1, k2, k3
    return out


def filter_backward_cov(desc, q1, f, g, sigc_q1, sigc_f, sigc_g, mu2, sig2, m0, sig0, fm0, fsig0, obs, eps, filt,
                        dlse=None, dFm=None, dFs=None, dlogW=None):
    """psvo_filter_backward_cov + two psvo_mlp_wgrad launches per MLP (one per head; both contribute to dW1 / db1).
    Returns a dict of gradients; the MLP entries gq1 / gf / gg are 6-tuples in the order of the MLP tuples."""
    lib = _lib.load()
    B, T, N, Dx, Dy, H = desc.B, desc.T, desc.N, desc.Dx, desc.Dy, desc.H
    dev = eps.device
    boot = bool(desc.bootstrap)
    q1s, k1 = _cov_struct(q1, Dx, H, Dx, "q1")
    fs, k2 = (None, None) if boot else _cov_struct(f, Dx, H, Dx, "f")
    gs, k3 = _cov_struct(g, Dx, H, Dy, "g")
    _chk(dlse, (T, B), "dlse"); _chk(dFm, (T, B, Dx, N), "dFm"); _chk(dFs, (T, B, Dx, N), "dFs")
    _chk(dlogW, (T, B, N), "dlogW")
    z = lambda *s: _empty(*s, device=dev)
    rows = lambda D: (z(T, B, D, N), z(T, B, D, N))
    out = {"dP": rows(Dx), "dF": (None, None) if boot else rows(Dx), "dG": rows(Dy),
           "dmu2": z(T, B, Dx) if desc.two_q else None, "dsig2": z(T, B, Dx) if desc.two_q else None,
           "dm0": z(B, Dx), "dsig0": z(B, Dx), "dfm0": z(B, Dx), "dfsig0": z(B, Dx),
           "dsigc_q1": z(Dx), "dsigc_f": None if boot else z(Dx), "dsigc_g": z(Dy)}
    ws = z(lib.psvo_filter_cov_ws_floats(B, T, N, Dx, Dy))
    _mark("psvo_filter_backward_cov", 0)
    st = lib.psvo_filter_backward_cov(
        ctypes.byref(desc), ctypes.byref(q1s), ctypes.byref(fs) if fs is not None else None, ctypes.byref(gs),
        _ptr(sigc_q1), _ptr(sigc_f), _ptr(sigc_g), _ptr(mu2), _ptr(sig2), _ptr(m0), _ptr(sig0), _ptr(fm0), _ptr(fsig0),
        _ptr(obs), _ptr(eps), _ptr(filt["X"]), _ptr(filt["Fm"]), _ptr(filt["Fs"]), _ptr(filt["P1"]), _ptr(filt["P1s"]),
        _ptr(filt["logW"]), _ptr(filt["lse"]), _ptr(filt["idx"]), _ptr(dlse), _ptr(dFm), _ptr(dFs), _ptr(dlogW),
        _ptr(out["dP"][0]), _ptr(out["dP"][1]), _ptr(out["dF"][0]), _ptr(out["dF"][1]), _ptr(out["dG"][0]),
        _ptr(out["dG"][1]), _ptr(out["dmu2"]), _ptr(out["dsig2"]), _ptr(out["dm0"]), _ptr(out["dsig0"]), _ptr(out["dfm0"]),
        _ptr(out["dfsig0"]), _ptr(out["dsigc_q1"]), _ptr(out["dsigc_f"]), _ptr(out["dsigc_g"]), _ptr(ws), _stream())
    _mark("psvo_filter_backward_cov", 1)
    _lib.check(st, "psvo_filter_backward_cov")
    del k1, k2, k3

    def head_grads(p, rows_pair, Dout):
        """[W1 | b1 | W_head | b_head] of each head from its rows -> (dW1, db1, dW_mu, db_mu, dW_sigma, db_sigma)"""
        W1, b1, Wm, bm, Ws, bs = p
        gm = split_mlp_grad(mlp_wgrad(filt["X"], rows_pair[0], (W1, b1, Wm, bm), Dx, H, Dout), Dx, H, Dout)
        gs_ = split_mlp_grad(mlp_wgrad(filt["X"], rows_pair[1], (W1, b1, Ws, bs), Dx, H, Dout), Dx, H, Dout)
        return (gm[0] + gs_[0], gm[1] + gs_[1], gm[2], gm[3], gs_[2], gs_[3])
    out["gq1"] = head_grads(q1, out["dP"], Dx)
    out["gf"] = None if boot else head_grads(f, out["dF"], Dx)
    out["gg"] = head_grads(g, out["dG"], Dy)
    return out


def bsim_forward_cov(desc, filt, f, g, q1_inv, sigc_f, sigc_g, sigc_q1inv, bmu2, bsig2, minit, sinit, imean, isig,
                     obs, eps_b, u_b=None, sel_in=None, save=False):
    """psvo_bsim_forward_cov.  Returns dict(bwX, flp, glp, Omega, sel, score[, lam, om, mu1, s1])."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs, k1 = _cov_struct(f, Dx, H, Dx, "f")
    gs, k2 = _cov_struct(g, Dx, H, Dy, "g")
    qs, k3 = _cov_struct(q1_inv, Dx, H, Dx, "q1_inv")
    _chk(filt["Fm"], (T, B, Dx, N), "Fm"); _chk(filt["Fs"], (T, B, Dx, N), "Fs")
    _chk(filt["logW"], (T, B, N), "logW"); _chk(filt["lse"], (T, B), "lse")
    _chk(sigc_f, (Dx,), "sigc_f"); _chk(sigc_g, (Dy,), "sigc_g"); _chk(sigc_q1inv, (Dx,), "sigc_q1inv")
    _chk(bmu2, (T, B, Dx), "bmu2"); _chk(bsig2, (T, B, Dx), "bsig2")
    for t, nm in ((minit, "minit"), (sinit, "sinit"), (imean, "imean"), (isig, "isig")):
        _chk(t, (B, Dx), nm)
    _chk(obs, (T, B, Dy), "obs"); _chk(eps_b, (T, B, Dx, N, M), "eps_b")
    _chk(u_b, (T, B, N), "u_b"); _chk(sel_in, (T, B, N), "sel_in", torch.int32)
    if u_b is None and sel_in is None:
        raise ValueError("the backward simulation needs uniforms `u_b` or teacher-forced `sel_in`")
    z = lambda *s: _empty(*s, device=dev)
    out = {"bwX": z(T, B, Dx, N), "flp": z(T, B, N), "glp": z(T, B, N), "Omega": z(T, B, N),
           "sel": _empty(T, B, N, device=dev, dtype=torch.int32), "score": z(B, N),
           "lam": z(T, B, N, M) if save else None, "om": z(T, B, N, M) if save else None,
           "mu1": z(T, B, Dx, N) if save else None, "s1": z(T, B, Dx, N) if save else None}
    _mark("psvo_bsim_forward_cov", 0)
    st = lib.psvo_bsim_forward_cov(
        ctypes.byref(desc), _ptr(filt["Fm"]), _ptr(filt["Fs"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs), _ptr(sigc_f), _ptr(sigc_g), _ptr(sigc_q1inv),
        _ptr(bmu2), _ptr(bsig2), _ptr(minit), _ptr(sinit), _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b), _ptr(u_b),
        _ptr(sel_in), _ptr(out["bwX"]), _ptr(out["flp"]), _ptr(out["glp"]), _ptr(out["Omega"]), _ptr(out["sel"]),
        _ptr(out["score"]), _ptr(out["lam"]), _ptr(out["om"]), _ptr(out["mu1"]), _ptr(out["s1"]), _stream())
    _mark("psvo_bsim_forward_cov", 1)
    _lib.check(st, "psvo_bsim_forward_cov")
    del k1, k2, k3
    return out


def bsim_backward_cov(desc, filt, f, g, q1_inv, sigc_f, sigc_g, sigc_q1inv, bmu2, bsig2, minit, sinit, imean, isig,
                      obs, eps_b, bs, dscore):
    """psvo_bsim_backward_cov + two psvo_mlp_wgrad launches per MLP (one per head).  `bs` = the forward call's outputs with
    its saves.  Returns a dict of gradients; gf / gg / gq1inv are 6-tuples in the order of the MLP tuples."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs, k1 = _cov_struct(f, Dx, H, Dx, "f")
    gs, k2 = _cov_struct(g, Dx, H, Dy, "g")
    qs, k3 = _cov_struct(q1_inv, Dx, H, Dx, "q1_inv")
    _chk(dscore, (B, N), "dscore")
    z = lambda *s: _empty(*s, device=dev)

    def zeros(*s):
        t = _empty(*s, device=dev)
        if _LAUNCH_STREAM is not None:
            with torch.cuda.stream(_LAUNCH_STREAM):
                return t.zero_()
        return t.zero_()
    out = {"xt": z(T, B, Dx, N, M), "dFt": z(T, B, Dx, N, M), "dFts": z(T, B, Dx, N, M), "dGt": z(T, B, Dy, N, M),
           "dGts": z(T, B, Dy, N, M), "dmu1": z(T, B, Dx, N), "dmu1s": z(T, B, Dx, N),
           # accumulated with float atomics by the kernel
           "dFm": zeros(T, B, Dx, N), "dFs": zeros(T, B, Dx, N), "dlogW": zeros(T, B, N), "dlse": zeros(T, B),
           "dbmu2": zeros(T, B, Dx), "dbsig2": zeros(T, B, Dx), "dminit": zeros(B, Dx), "dsinit": zeros(B, Dx),
           "dimean": zeros(B, Dx), "disig": zeros(B, Dx), "dsigc_f": zeros(Dx), "dsigc_g": zeros(Dy),
           "dsigc_q1inv": zeros(Dx)}
    _mark("psvo_bsim_backward_cov", 0)
    st = lib.psvo_bsim_backward_cov(
        ctypes.byref(desc), _ptr(filt["Fm"]), _ptr(filt["Fs"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs), _ptr(sigc_f), _ptr(sigc_g), _ptr(sigc_q1inv),
        _ptr(bmu2), _ptr(bsig2), _ptr(minit), _ptr(sinit), _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b),
        _ptr(bs["bwX"]), _ptr(bs["sel"]), _ptr(bs["lam"]), _ptr(bs["om"]), _ptr(bs["mu1"]), _ptr(bs["s1"]), _ptr(dscore),
        _ptr(out["xt"]), _ptr(out["dFt"]), _ptr(out["dFts"]), _ptr(out["dGt"]), _ptr(out["dGts"]), _ptr(out["dmu1"]),
        _ptr(out["dmu1s"]), _ptr(out["dFm"]), _ptr(out["dFs"]), _ptr(out["dlogW"]), _ptr(out["dlse"]), _ptr(out["dbmu2"]),
        _ptr(out["dbsig2"]), _ptr(out["dminit"]), _ptr(out["dsinit"]), _ptr(out["dimean"]), _ptr(out["disig"]),
        _ptr(out["dsigc_f"]), _ptr(out["dsigc_g"]), _ptr(out["dsigc_q1inv"]), _stream())
    _mark("psvo_bsim_backward_cov", 1)
    _lib.check(st, "psvo_bsim_backward_cov")
    del k1, k2, k3

    def head_grads(p, X, rows_mu, rows_sig, Dout, axis=2):
        W1, b1, Wm, bm, Ws, bs_ = p
        gm = split_mlp_grad(mlp_wgrad(X, rows_mu, (W1, b1, Wm, bm), Dx, H, Dout, axis=axis), Dx, H, Dout)
        gs_ = split_mlp_grad(mlp_wgrad(X, rows_sig, (W1, b1, Ws, bs_), Dx, H, Dout, axis=axis), Dx, H, Dout)
        return (gm[0] + gs_[0], gm[1] + gs_[1], gm[2], gm[3], gs_[2], gs_[3])
    out["gf"] = head_grads(f, out["xt"][:T - 1], out["dFt"][:T - 1], out["dFts"][:T - 1], Dx)
    out["gg"] = head_grads(g, out["xt"], out["dGt"], out["dGts"], Dy)
    out["gq1inv"] = head_grads(q1_inv, bs["bwX"][1:], out["dmu1"][:T - 1], out["dmu1s"][:T - 1], Dx)
    return out


def bsimwr_forward_cov(desc, filt, f, g, q1_inv, sigc_f, sigc_g, sigc_q1inv, bmu2, bsig2, minit, sinit, imean, isig,
                       obs, eps_b, u_b=None, u_r=None, sel_in=None, anc_in=None, save=False):
    """psvo_bsimwr_forward_cov (PSVOwR, state-dependent scales).  Returns dict(bwX, bwXanc, bwW, lseW, sel, anc, omsel
    [, lam, om, mu1, s1])."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs, k1 = _cov_struct(f, Dx, H, Dx, "f")
    gs, k2 = _cov_struct(g, Dx, H, Dy, "g")
    qs, k3 = _cov_struct(q1_inv, Dx, H, Dx, "q1_inv")
    _chk(filt["Fm"], (T, B, Dx, N), "Fm"); _chk(filt["Fs"], (T, B, Dx, N), "Fs")
    _chk(filt["logW"], (T, B, N), "logW"); _chk(filt["lse"], (T, B), "lse")
    _chk(sigc_f, (Dx,), "sigc_f"); _chk(sigc_g, (Dy,), "sigc_g"); _chk(sigc_q1inv, (Dx,), "sigc_q1inv")
    _chk(bmu2, (T, B, Dx), "bmu2"); _chk(bsig2, (T, B, Dx), "bsig2")
    for t, nm in ((minit, "minit"), (sinit, "sinit"), (imean, "imean"), (isig, "isig")):
        _chk(t, (B, Dx), nm)
    _chk(obs, (T, B, Dy), "obs"); _chk(eps_b, (T, B, Dx, N, M), "eps_b")
    _chk(u_b, (T, B, N), "u_b"); _chk(sel_in, (T, B, N), "sel_in", torch.int32)
    _chk(u_r, (T, B, N), "u_r"); _chk(anc_in, (T, B, N), "anc_in", torch.int32)
    if (u_b is None and sel_in is None) or (u_r is None and anc_in is None):
        raise ValueError("PSVOwR needs uniforms (`u_b`, `u_r`) or teacher-forced indices (`sel_in`, `anc_in`)")
    z = lambda *s: _empty(*s, device=dev)
    zi = lambda *s: _empty(*s, device=dev, dtype=torch.int32)
    out = {"bwX": z(T, B, Dx, N), "bwXanc": z(T, B, Dx, N), "bwW": z(T, B, N), "lseW": z(T, B), "sel": zi(T, B, N),
           "anc": zi(T, B, N), "omsel": z(T, B, N),
           "lam": z(T, B, N, M) if save else None, "om": z(T, B, N, M) if save else None,
           "mu1": z(T, B, Dx, N) if save else None, "s1": z(T, B, Dx, N) if save else None}
    _mark("psvo_bsimwr_forward_cov", 0)
    st = lib.psvo_bsimwr_forward_cov(
        ctypes.byref(desc), _ptr(filt["Fm"]), _ptr(filt["Fs"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs), _ptr(sigc_f), _ptr(sigc_g), _ptr(sigc_q1inv),
        _ptr(bmu2), _ptr(bsig2), _ptr(minit), _ptr(sinit), _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b), _ptr(u_b),
        _ptr(u_r), _ptr(sel_in), _ptr(anc_in), _ptr(out["bwX"]), _ptr(out["bwXanc"]), _ptr(out["bwW"]), _ptr(out["lseW"]),
        _ptr(out["sel"]), _ptr(out["anc"]), _ptr(out["omsel"]), _ptr(out["lam"]), _ptr(out["om"]), _ptr(out["mu1"]),
        _ptr(out["s1"]), _stream())
    _mark("psvo_bsimwr_forward_cov", 1)
    _lib.check(st, "psvo_bsimwr_forward_cov")
    del k1, k2, k3
    return out


def bsimwr_backward_cov(desc, filt, f, g, q1_inv, sigc_f, sigc_g, sigc_q1inv, bmu2, bsig2, minit, sinit, imean, isig,
                        obs, eps_b, bs, dlseW):
    """psvo_bsimwr_backward_cov + two psvo_mlp_wgrad launches per MLP.  `bs` = the forward call's outputs with its saves."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs, k1 = _cov_struct(f, Dx, H, Dx, "f")
    gs, k2 = _cov_struct(g, Dx, H, Dy, "g")
    qs, k3 = _cov_struct(q1_inv, Dx, H, Dx, "q1_inv")
    _chk(dlseW, (T, B), "dlseW")
    z = lambda *s: _empty(*s, device=dev)

    def zeros(*s):
        t = _empty(*s, device=dev)
        if _LAUNCH_STREAM is not None:
            with torch.cuda.stream(_LAUNCH_STREAM):
                return t.zero_()
        return t.zero_()
    out = {"xt": z(T, B, Dx, N, M), "dFt": z(T, B, Dx, N, M), "dFts": z(T, B, Dx, N, M), "dGt": z(T, B, Dy, N, M),
           "dGts": z(T, B, Dy, N, M), "dmu1": z(T, B, Dx, N), "dmu1s": z(T, B, Dx, N),
           "dFm": zeros(T, B, Dx, N), "dFs": zeros(T, B, Dx, N), "dlogW": zeros(T, B, N), "dlse": zeros(T, B),
           "dbmu2": zeros(T, B, Dx), "dbsig2": zeros(T, B, Dx), "dminit": zeros(B, Dx), "dsinit": zeros(B, Dx),
           "dimean": zeros(B, Dx), "disig": zeros(B, Dx), "dsigc_f": zeros(Dx), "dsigc_g": zeros(Dy),
           "dsigc_q1inv": zeros(Dx)}
    dXs = zeros(T, B, Dx, N)
    _mark("psvo_bsimwr_backward_cov", 0)
    st = lib.psvo_bsimwr_backward_cov(
        ctypes.byref(desc), _ptr(filt["Fm"]), _ptr(filt["Fs"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs), _ptr(sigc_f), _ptr(sigc_g), _ptr(sigc_q1inv),
        _ptr(bmu2), _ptr(bsig2), _ptr(minit), _ptr(sinit), _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b),
        _ptr(bs["bwXanc"]), _ptr(bs["bwW"]), _ptr(bs["lseW"]), _ptr(bs["sel"]), _ptr(bs["anc"]), _ptr(bs["lam"]),
        _ptr(bs["om"]), _ptr(bs["mu1"]), _ptr(bs["s1"]), _ptr(dlseW),
        _ptr(out["xt"]), _ptr(out["dFt"]), _ptr(out["dFts"]), _ptr(out["dGt"]), _ptr(out["dGts"]), _ptr(out["dmu1"]),
        _ptr(out["dmu1s"]), _ptr(out["dFm"]), _ptr(out["dFs"]), _ptr(out["dlogW"]), _ptr(out["dlse"]), _ptr(out["dbmu2"]),
        _ptr(out["dbsig2"]), _ptr(out["dminit"]), _ptr(out["dsinit"]), _ptr(out["dimean"]), _ptr(out["disig"]),
        _ptr(out["dsigc_f"]), _ptr(out["dsigc_g"]), _ptr(out["dsigc_q1inv"]), _ptr(dXs), _stream())
    _mark("psvo_bsimwr_backward_cov", 1)
    _lib.check(st, "psvo_bsimwr_backward_cov")
    del k1, k2, k3

    def head_grads(p, X, rows_mu, rows_sig, Dout):
        W1, b1, Wm, bm, Ws, bs_ = p
        gm = split_mlp_grad(mlp_wgrad(X, rows_mu, (W1, b1, Wm, bm), Dx, H, Dout), Dx, H, Dout)
        gs_ = split_mlp_grad(mlp_wgrad(X, rows_sig, (W1, b1, Ws, bs_), Dx, H, Dout), Dx, H, Dout)
        return (gm[0] + gs_[0], gm[1] + gs_[1], gm[2], gm[3], gs_[2], gs_[3])
    out["gf"] = head_grads(f, out["xt"][:T - 1], out["dFt"][:T - 1], out["dFts"][:T - 1], Dx)
    out["gg"] = head_grads(g, out["xt"], out["dGt"], out["dGts"], Dy)
    out["gq1inv"] = head_grads(q1_inv, bs["bwXanc"][1:], out["dmu1"][:T - 1], out["dmu1s"][:T - 1], Dx)
    return out


def _chain_rows(out, z, T, B, Dx, N):
    """per-chain rows of d bmu2 (T,B,Dx,N), d minit and d imean (B,Dx,N) in ONE buffer, so that one reduction over the
    chains serves all three (they sit on the dependent chain in front of the encoder BPTT)"""
    rows = z((T + 2) * B * Dx, N)
    n = T * B * Dx
    out["chain_rows"] = rows
    out["dbmu2_rows"] = rows[:n].view(T, B, Dx, N)
    out["dminit_rows"] = rows[n:n + B * Dx].view(B, Dx, N)
    out["dimean_rows"] = rows[n + B * Dx:].view(B, Dx, N)


def sum_chain_rows(out, T, B, Dx):
    """d bmu2 (T,B,Dx), d minit, d imean (B,Dx) from the chain rows of bsim_backward / bsimwr_backward.  Called by the
    autograd node when it returns, i.e. AFTER the weight-gradient launches were enqueued: issued between the kernel and
    those launches it changes the captured graph's topology, and the hipGraph executor then ran the filter's reverse pass
    and the weight gradients one after the other (+0.45 ms per replayed step at C*)."""
    s = out["chain_rows"].sum(-1)
    n = T * B * Dx
    out["dbmu2"], out["dminit"], out["dimean"] = s[:n].view(T, B, Dx), s[n:n + B * Dx].view(B, Dx), s[n + B * Dx:].view(B, Dx)


def bsim_backward(desc, filt, f, g, q1_inv, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig,
                  obs, eps_b, bs, dscore, gbufs=None, after_kernel=None, wgrad_stream=None, defer_wgrad=False):
    """psvo_bsim_backward + psvo_mlp_wgrad.  `bs` = bsim_forward(..., save=True) outputs."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs = _mlp_struct(f, Dx, H, Dx, "f")
    gs = _mlp_struct(g, Dx, H, Dy, "g")
    qs = _mlp_struct(q1_inv, Dx, H, Dx, "q1_inv")
    _chk(dscore, (B, N), "dscore")
    for k, shp in (("lam2", (T, B, N, M)), ("om", (T, B, N, M)), ("mu1", (T, B, Dx, N)), ("bwX", (T, B, Dx, N))):
        if bs.get(k) is None:
            raise ValueError("bsim_backward needs bsim_forward(save=True) outputs (missing %s)" % k)
        _chk(bs[k], shp, k)
    nblk = lib.psvo_bsim_blocks(ctypes.byref(desc))
    z = lambda *s: _empty(*s, device=dev)
    out = {"xt": z(T, B, Dx, N, M), "dFt": z(T, B, Dx, N, M), "dGt": z(T, B, Dy, N, M), "dmu1": z(T, B, Dx, N),
           "dFm_part": z(T, B, nblk, Dx, N), "dlogW_part": z(T, B, nblk, N), "dFm": z(T, B, Dx, N), "dlogW": z(T, B, N),
           "dlse": z(T, B), "dsig_f": z(Dx), "dsig_g": z(Dy), "dsig_q1inv": z(Dx), "dsig_bq2": z(Dx), "dsig_init": z(Dx), "disig": z(Dx)}
    _chain_rows(out, z, T, B, Dx, N)
    sacc = z(B, nblk, lib.psvo_bsim_acc_size(Dx, Dy))
    _mark("psvo_bsim_backward", 0)
    st = lib.psvo_bsim_backward(
        ctypes.byref(desc), _ptr(filt["Fm"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs),
        _ptr(sig_f), _ptr(sig_g), _ptr(sig_q1inv), _ptr(sig_bq2), _ptr(bmu2), _ptr(minit), _ptr(sig_init),
        _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b), _ptr(bs["bwX"]), _ptr(bs["sel"]),
        _ptr(bs["lam2"]), _ptr(bs["om"]), _ptr(bs["mu1"]), _ptr(dscore),
        _ptr(out["xt"]), _ptr(out["dFt"]), _ptr(out["dGt"]), _ptr(out["dmu1"]),
        _ptr(out["dFm_part"]), _ptr(out["dlogW_part"]), _ptr(out["dbmu2_rows"]), _ptr(out["dminit_rows"]),
        _ptr(out["dimean_rows"]), _ptr(sacc), _stream())
    _mark("psvo_bsim_backward", 1)
    _lib.check(st, "psvo_bsim_backward")
    # one more launch folds the per-workgroup partials into d Fm / d logW (which feed the filter's reverse pass) and the
    # scale gradients; then the caller publishes them
    _mark("psvo_bsim_backward_fold", 0)
    st = lib.psvo_bsim_backward_fold(
        ctypes.byref(desc), _ptr(out["dFm_part"]), _ptr(out["dlogW_part"]), _ptr(sacc), _ptr(sig_q1inv), _ptr(sig_bq2),
        _ptr(out["dFm"]), _ptr(out["dlogW"]), _ptr(out["dsig_f"]), _ptr(out["dsig_g"]), _ptr(out["dsig_q1inv"]),
        _ptr(out["dsig_bq2"]), _ptr(out["dsig_init"]), _ptr(out["disig"]), _ptr(out["dlse"]), _stream())
    _mark("psvo_bsim_backward_fold", 1)
    _lib.check(st, "psvo_bsim_backward_fold")
    if after_kernel is not None:
        after_kernel(out)
    # weight gradients from rows: MLP_f / MLP_g on the sub-particles, MLP_q1inv on bwX[t+1];
    # gbufs = (f, g, q1_inv) slices of the flat gradient buffer to accumulate into directly, or None
    # wgrad_stream (only with gbufs: nothing is returned that the caller would read): issue the weight gradients on
    # that stream, ordered after the kernel above, so that the caller's stream is free for what comes next
    gb = gbufs or (None, None, None)

    def wgrad():
        out["gf"] = mlp_wgrad(out["xt"][:T - 1], out["dFt"][:T - 1], f, Dx, H, Dx, grad=gb[0])
        out["gg"] = mlp_wgrad(out["xt"], out["dGt"], g, Dx, H, Dy, grad=gb[1])
        out["gq1inv"] = mlp_wgrad(bs["bwX"][1:], out["dmu1"][:T - 1], q1_inv, Dx, H, Dx, grad=gb[2])
    if defer_wgrad:      # the caller issues them (out["_wgrad"]() under launch_on(stream), ordered after this kernel)
        out["_wgrad"] = wgrad
        return out
    with _wgrad_on(wgrad_stream if gbufs is not None else None):
        wgrad()
    return out


def bsimwr_forward(desc, filt, f, g, q1_inv, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init,
                   imean, isig, obs, eps_b, u_b=None, u_r=None, sel_in=None, anc_in=None, save=False):
    """psvo_bsimwr_forward (PSVOwR).  Returns dict(bwX, bwXanc, bwW, lseW, sel, anc[, lam2, om, mu1])."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs = _mlp_struct(f, Dx, H, Dx, "f")
    gs = _mlp_struct(g, Dx, H, Dy, "g")
    qs = _mlp_struct(q1_inv, Dx, H, Dx, "q1_inv")
    _chk(filt["Fm"], (T, B, Dx, N), "Fm")
    _chk(filt["logW"], (T, B, N), "logW"); _chk(filt["lse"], (T, B), "lse")
    _chk(sig_f, (Dx,), "sig_f"); _chk(sig_g, (Dy,), "sig_g")
    _chk(sig_q1inv, (Dx,), "sig_q1inv"); _chk(sig_bq2, (Dx,), "sig_bq2")
    _chk(bmu2, (T, B, Dx), "bmu2"); _chk(minit, (B, Dx), "minit"); _chk(sig_init, (Dx,), "sig_init")
    _chk(imean, (B, Dx), "imean"); _chk(isig, (Dx,), "isig")
    _chk(obs, (T, B, Dy), "obs"); _chk(eps_b, (T, B, Dx, N, M), "eps_b")
    _chk(u_b, (T, B, N), "u_b"); _chk(sel_in, (T, B, N), "sel_in", torch.int32)
    _chk(u_r, (T, B, N), "u_r"); _chk(anc_in, (T, B, N), "anc_in", torch.int32)
    if (u_b is None and sel_in is None) or (u_r is None and anc_in is None):
        raise ValueError("PSVOwR needs uniforms (`u_b`, `u_r`) or teacher-forced indices (`sel_in`, `anc_in`)")
    z = lambda *s: _empty(*s, device=dev)
    zi = lambda *s: _empty(*s, device=dev, dtype=torch.int32)
    out = {"bwX": z(T, B, Dx, N), "bwXanc": z(T, B, Dx, N), "bwW": z(T, B, N), "lseW": z(T, B),
           "sel": zi(T, B, N), "anc": zi(T, B, N),
           "lam2": z(T, B, N, M) if save else None, "om": z(T, B, N, M) if save else None,
           "mu1": z(T, B, Dx, N) if save else None,
           # workspace; its last word (viewed as int32) is nonzero iff a cluster barrier timed out
           "ws": z(lib.psvo_bsimwr_ws_floats(B, T, N))}
    _mark("psvo_bsimwr_forward", 0)
    st = lib.psvo_bsimwr_forward(
        ctypes.byref(desc), _ptr(filt["Fm"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs),
        _ptr(sig_f), _ptr(sig_g), _ptr(sig_q1inv), _ptr(sig_bq2), _ptr(bmu2), _ptr(minit), _ptr(sig_init),
        _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b), _ptr(u_b), _ptr(u_r), _ptr(sel_in), _ptr(anc_in),
        _ptr(out["bwX"]), _ptr(out["bwXanc"]), _ptr(out["bwW"]), _ptr(out["lseW"]), _ptr(out["sel"]),
        _ptr(out["anc"]), _ptr(out["lam2"]), _ptr(out["om"]), _ptr(out["mu1"]), _ptr(out["ws"]), _stream())
    _mark("psvo_bsimwr_forward", 1)
    _lib.check(st, "psvo_bsimwr_forward")
    return out


def bsimwr_backward(desc, filt, f, g, q1_inv, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig,
                    obs, eps_b, bs, dlseW, gbufs=None, after_kernel=None, wgrad_stream=None):
    """psvo_bsimwr_backward + psvo_mlp_wgrad.  `bs` = bsimwr_forward(..., save=True) outputs."""
    lib = _lib.load()
    B, T, N, M, Dx, Dy, H = desc.B, desc.T, desc.N, desc.M, desc.Dx, desc.Dy, desc.H
    dev = eps_b.device
    fs = _mlp_struct(f, Dx, H, Dx, "f")
    gs = _mlp_struct(g, Dx, H, Dy, "g")
    qs = _mlp_struct(q1_inv, Dx, H, Dx, "q1_inv")
    _chk(dlseW, (T, B), "dlseW")
    for k, shp in (("lam2", (T, B, N, M)), ("om", (T, B, N, M)), ("mu1", (T, B, Dx, N)), ("bwXanc", (T, B, Dx, N)),
                   ("bwW", (T, B, N)), ("lseW", (T, B))):
        if bs.get(k) is None:
            raise ValueError("bsimwr_backward needs bsimwr_forward(save=True) outputs (missing %s)" % k)
        _chk(bs[k], shp, k)
    z = lambda *s: _empty(*s, device=dev)
    K = lib.psvo_bsimwr_blocks(B, N, M)
    out = {"xt": z(T, B, Dx, N, M), "dFt": z(T, B, Dx, N, M), "dGt": z(T, B, Dy, N, M), "dmu1": z(T, B, Dx, N),
           "dFm_part": z(T, B, K, Dx, N), "dlogW_part": z(T, B, K, N), "dlse_part": z(T, B, K),
           "dsig_f": z(Dx), "dsig_g": z(Dy), "dsig_q1inv": z(Dx), "dsig_bq2": z(Dx), "dsig_init": z(Dx), "disig": z(Dx),
           "ws": z(lib.psvo_bsimwr_bwd_ws_floats(B, T, N, Dx))}
    _chain_rows(out, z, T, B, Dx, N)
    sacc = z(B, K, lib.psvo_bsim_acc_size(Dx, Dy))
    _mark("psvo_bsimwr_backward", 0)
    st = lib.psvo_bsimwr_backward(
        ctypes.byref(desc), _ptr(filt["Fm"]), _ptr(filt["logW"]), _ptr(filt["lse"]),
        ctypes.byref(fs), ctypes.byref(gs), ctypes.byref(qs),
        _ptr(sig_f), _ptr(sig_g), _ptr(sig_q1inv), _ptr(sig_bq2), _ptr(bmu2), _ptr(minit), _ptr(sig_init),
        _ptr(imean), _ptr(isig), _ptr(obs), _ptr(eps_b), _ptr(bs["bwXanc"]), _ptr(bs["bwW"]), _ptr(bs["lseW"]),
        _ptr(bs["sel"]), _ptr(bs["anc"]), _ptr(bs["lam2"]), _ptr(bs["om"]), _ptr(bs["mu1"]), _ptr(dlseW),
        _ptr(out["xt"]), _ptr(out["dFt"]), _ptr(out["dGt"]), _ptr(out["dmu1"]),
        _ptr(out["dFm_part"]), _ptr(out["dlogW_part"]), _ptr(out["dlse_part"]), _ptr(out["dbmu2_rows"]),
        _ptr(out["dminit_rows"]), _ptr(out["dimean_rows"]), _ptr(out["dsig_f"]), _ptr(out["dsig_g"]),
        _ptr(out["dsig_q1inv"]), _ptr(out["dsig_bq2"]), _ptr(out["dsig_init"]), _ptr(out["disig"]), _ptr(sacc),
        _ptr(out["ws"]), _stream())
    _mark("psvo_bsimwr_backward", 1)
    _lib.check(st, "psvo_bsimwr_backward")
    # fold the per-workgroup partials that feed the filter's reverse pass, then let the caller publish them
    out["dFm"] = out["dFm_part"].sum(2)
    out["dlogW"] = out["dlogW_part"].sum(2)
    out["dlse"] = out["dlse_part"].sum(2)
    if after_kernel is not None:
        after_kernel(out)
    gb = gbufs or (None, None, None)
    with _wgrad_on(wgrad_stream if gbufs is not None else None):
        out["gf"] = mlp_wgrad(out["xt"][:T - 1], out["dFt"][:T - 1], f, Dx, H, Dx, grad=gb[0])
        out["gg"] = mlp_wgrad(out["xt"], out["dGt"], g, Dx, H, Dy, grad=gb[1])
        out["gq1inv"] = mlp_wgrad(bs["bwXanc"][1:], out["dmu1"][:T - 1], q1_inv, Dx, H, Dx, grad=gb[2])
    return out


def reduce_rows(part, nrows, stride, n, out, accumulate=False):
    """psvo_reduce_rows: out[p] (+)= sum_r part[r*stride + p]; `part` may be a strided view (its data_ptr is used)."""
    lib = _lib.load()
    st = lib.psvo_reduce_rows(_ptr(part), nrows, stride, n, _ptr(out), int(accumulate), _stream())
    _lib.check(st, "psvo_reduce_rows")
    return out


def sigma_forward(raw, mins):
    lib = _lib.load()
    sig = _empty(raw.shape, device=raw.device, dtype=raw.dtype)
    _lib.check(lib.psvo_sigma_forward(_ptr(raw), _ptr(mins), _ptr(sig), raw.numel(), _stream()), "psvo_sigma_forward")
    return sig


def sigma_backward(raw, mins, dsig, graw, accumulate=True):
    lib = _lib.load()
    _lib.check(lib.psvo_sigma_backward(_ptr(raw), _ptr(mins), _ptr(dsig), _ptr(graw), raw.numel(), int(accumulate),
                                       _stream()), "psvo_sigma_backward")
    return graw


def bilstm_backward(x, W_fw, W_bw, out, cs, gates, dout, gbufs=None, need_dx=True):
    """psvo_bilstm_backward -> (dx (B,T,Din) or None, dW_fw, db_fw, dW_bw, db_bw)."""
    lib = _lib.load()
    B, T, Din = x.shape
    Dh = W_fw.shape[1] // 4
    _chk(dout, (B, T, 2 * Dh), "dout"); _chk(out, (B, T, 2 * Dh), "out")
    _chk(cs, (2, B, T, Dh), "cs"); _chk(gates, (2, B, T, 4 * Dh), "gates")
    dev = x.device
    dx_part = _empty(2, B, T, Din, device=dev)
    dW_part = _empty(B, 2, Din + Dh, 4 * Dh, device=dev)
    db_part = _empty(B, 2, 4 * Dh, device=dev)
    _mark("psvo_bilstm_backward", 0)
    st = lib.psvo_bilstm_backward(B, T, Din, Dh, _ptr(x), _ptr(W_fw), _ptr(W_bw), _ptr(out), _ptr(cs), _ptr(gates),
                                  _ptr(dout), _ptr(dx_part), _ptr(dW_part), _ptr(db_part), _stream())
    _mark("psvo_bilstm_backward", 1)
    _lib.check(st, "psvo_bilstm_backward")
    dx = dx_part.sum(0) if need_dx else None     # (the first layer's input is the observations: no gradient wanted)
    K, G4 = Din + Dh, 4 * Dh
    if gbufs is not None and gbufs[0] is not None and gbufs[1] is not None:
        # fold the per-sequence partials straight into the flat gradient buffer: [kernel (K,4Dh) | bias (4Dh)] per cell
        _lib.check(lib.psvo_bilstm_wgrad_fold(B, Din, Dh, _ptr(dW_part), _ptr(db_part), _ptr(gbufs[0]), _ptr(gbufs[1]), 1,
                                              _stream()), "psvo_bilstm_wgrad_fold")
        return dx, None, None, None, None
    dW = dW_part.sum(0)
    db = db_part.sum(0)
    return dx, dW[0], db[0], dW[1], db[1]
