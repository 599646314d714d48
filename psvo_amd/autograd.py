"""torch.autograd glue around the HIP forward / backward kernels.

The reference differentiates `-log_ZSMC` with TensorFlow autodiff (src/trainer.py:115-118).  Here
the per-particle part of the graph (filter, backward simulation) is two opaque nodes whose
backward passes are hand-written persistent kernels (psvo_filter_backward, psvo_bsim_backward,
psvo_mlp_wgrad); everything upstream of them (hoisted per-(b, t) proposal means, the encoder,
softplus/clamp of the scale parameters) stays in torch autograd.
"""
import os

import torch

from . import ops


# ---------------------------------------------------------------------------------------------
# Stream overlap.  The filter kernels run one workgroup per sequence (32 of 256 CUs at C*), so they
# are issued on a side stream and overlap with work that does not depend on them:
#   forward : filter  ||  bsim noise draws, observation encoder, hoisted backward-proposal means
#   backward: filter reverse pass + its weight gradients (side stream)  ||  weight gradients of the bsim rows
#             (second side stream, when they accumulate straight into the flat gradient buffer)  ||  hoisted-MLP
#             backward, encoder BPTT (main stream)
# Synchronisation is by events only (never a device sync).
# ---------------------------------------------------------------------------------------------
_SIDE = {}
OVERLAP = True
# Opt-in: let the main stream run ahead of the filter's weight-gradient launches at the end of the reverse pass.  Whoever sets
# this MUST call join_deferred() after backward() and before reading or reducing the gradients (trainer.train_step and
# bench.py do); left False, FilterFunction.backward joins the streams itself and gradients are safe to read on return.
DEFER_JOIN = False
_PENDING_JOIN = []
# the backward-simulation node launches the filter's reverse kernel itself, right behind its own (launch_reverse_kernel)
PRELAUNCH = os.environ.get("PSVO_PRELAUNCH", "1") != "0"
# ... and leaves its own weight-gradient launches (second side stream) to be issued when the engine reaches the filter node,
# i.e. BEHIND the small launches of the encoder's reverse chain, which otherwise run beside them on a full card
DEFER_BSIM_WGRAD = os.environ.get("PSVO_DEFER_BSIM_WGRAD", "1") != "0"
_PENDING_WORK = []
# A/B switch (off: measured, section 1 of DESIGN.md): the filter's MLP_g weight gradient on the second side stream
SPLIT_FILTER_WGRAD = os.environ.get("PSVO_FILTER_WGRAD_SPLIT", "0") == "1"
SPLIT_CROSS_WAIT = False      # tools/capture_probe.py only: reproduce the cross-wait crash of hip::Stream::EndCapture


def join_deferred():
    """order the current stream after the side-stream launches that backward() left running (no-op if none)"""
    while _PENDING_WORK:        # (deferred launches nobody issued: the filter node did not run)
        ov = _PENDING_WORK.pop()
        fn, ov.pending_wgrad = ov.pending_wgrad, None
        if fn is not None:
            fn()
            _PENDING_JOIN.append(ov.side2)
    cur = torch.cuda.current_stream() if _PENDING_JOIN else None
    while _PENDING_JOIN:
        cur.wait_stream(_PENDING_JOIN.pop())


class deferred_join(object):
    """with autograd.deferred_join(): loss.backward()   -- sets DEFER_JOIN for the block and joins on exit"""

    def __enter__(self):
        global DEFER_JOIN
        self.prev, DEFER_JOIN = DEFER_JOIN, True
        return self

    def __exit__(self, *exc):
        global DEFER_JOIN
        DEFER_JOIN = self.prev
        join_deferred()
        return False


class Overlap(object):
    """events shared by the FilterFunction and BsimFunction nodes of ONE objective evaluation"""

    def __init__(self, side, side2=None):
        self.side = side
        self.side2 = side2                 # weight gradients of the backward-simulation rows (None: main stream)
        self.filter_done = None
        self.bsim_grads_ready = None
        self.bsim_wgrad_done = None
        self.pending_wgrad = None          # the backward simulation's weight-gradient launches, when deferred
        self.pre = None                    # its reverse kernel, launched early by the backward-simulation node


def side_stream(device=None, which=0):
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if (dev, which) not in _SIDE:
        # (a high-priority stream for the filter changed nothing in replayed or eagerly issued steps: measured)
        _SIDE[(dev, which)] = torch.cuda.Stream(device=dev)
    return _SIDE[(dev, which)]


def _cf(t):
    if t is None:
        return None
    t = t.detach()
    if t.dtype is not torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _aliases(d):
    """the tensors of a forward's result dict as fresh aliases (same storage, new tensor objects) for keeping on ctx.

    A Function's output tensors get the node as grad_fn; the very same objects stored on ctx close a reference cycle
    (node -> ctx -> tensor -> grad_fn) that only the cycle collector frees.  Until it runs, the finished evaluation's graph
    stays alive and with it the parameters' gradient accumulators, which the next evaluation then REUSES together with the
    stream they were created on: under a hipGraph capture the engine synchronises that stream into the capture, nothing
    joins it back, and hipStreamEndCapture fails (or crashes, for the null stream, on ROCm 7.2)."""
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()}


def _cg(t):
    return None if t is None else t.contiguous()


def _mlps(t, n=3):
    """the n MLPs of a node's positional tensors: t[0:4n] = (W1, b1, W2, b2) each; with two hidden layers the node
    receives (Wh, bh) per MLP as 2n extra tensors behind its regular 4n + 9 ones (SVO._mlp_args)."""
    mlps = [tuple(t[4 * i:4 * i + 4]) for i in range(n)]
    extra = t[4 * n + 9:]
    if extra:
        mlps = [m + tuple(extra[2 * i:2 * i + 2]) if extra[2 * i] is not None else m for i, m in enumerate(mlps)]
    return mlps


def _mlp_grads(r, keys, dims, H, layers, gbufs):
    """gradients of a node's MLP inputs in its positional layout -> (first 4 per MLP, extras (Wh, bh) per MLP).
    keys: result-dict entries (None = MLP absent); gbufs[i] not None = accumulated in place by the kernels."""
    first, extra = (), ()
    for key, (Din, Dout), gb in zip(keys, dims, gbufs):
        if key is None or gb is not None:
            first += (None,) * 4
            extra += (None,) * 2
        else:
            g = ops.split_mlp_grad(r[key], Din, H, Dout, layers)
            first += tuple(g[:4])
            extra += tuple(g[4:6]) if layers == 2 else (None, None)
    return first, (extra if layers == 2 else ())


def _used_on(stream, tensors):
    """tensors allocated from another stream's pool are about to be consumed on `stream`"""
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(stream)


class FilterFunction(torch.autograd.Function):
    """psvo_filter_forward / psvo_filter_backward.

    apply(desc, obs_TB, eps, u, idx_in,
          q1W1, q1b1, q1W2, q1b2, fW1, fb1, fW2, fb2, gW1, gb1, gW2, gb2,
          sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0)
      -> lse (T,B), Fm (T,B,Dx,N), logW (T,B,N)   [differentiable]
         X, Xanc (T,B,Dx,N), idx (T,B,N) int32      [constants]
    f* are None when desc.bootstrap; sig_q2 / mu2 are None when not desc.two_q.
    """

    @staticmethod
    def forward(ctx, desc, obs_TB, eps, u, idx_in, *t):
        # missing upstream gradients stay None: a zero tensor materialised by the engine would be filled on the
        # main stream AFTER the event the side-stream reverse pass waits for
        ctx.set_materialize_grads(False)
        t = [_cf(v) for v in t]
        q1, f, g = _mlps(t)
        sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0 = t[12:21]
        ctx.gbufs = getattr(desc, "_gbufs", None)     # (q1, f, g) flat-gradient slices or None
        if desc.bootstrap:
            f = None
        ov = getattr(desc, "_ov", None)             # shared with the BsimFunction of the same evaluation
        if ov is not None:                          # PSVO: the consumer (BsimFunction) waits on ov.filter_done
            ov.side.wait_stream(torch.cuda.current_stream())
            with ops.launch_on(ov.side):
                filt = ops.filter_forward(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0,
                                          obs_TB, eps, u, idx_in)
            ov.filter_done = torch.cuda.Event()
            ov.filter_done.record(ov.side)
            _used_on(torch.cuda.current_stream(), filt.values())
        else:
            filt = ops.filter_forward(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0,
                                      obs_TB, eps, u, idx_in)
        ctx.desc, ctx.filt = desc, _aliases(filt)
        ctx.saved = (q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0, obs_TB, eps)
        ctx.mark_non_differentiable(filt["X"], filt["Xanc"], filt["idx"])
        return filt["lse"], filt["Fm"], filt["logW"], filt["X"], filt["Xanc"], filt["idx"]

    @staticmethod
    def launch_reverse_kernel(ctx, dlse, dFm, dlogW):
        """overlap wiring: psvo_filter_backward (kernel, row sums, scale gradients) on the side stream, ordered after the
        event that publishes the upstream gradients; its weight-gradient launches are left to backward() (r["_wgrad"]).
        Called by backward(), or EARLY by the backward-simulation node right after its own kernel (PRELAUNCH): the engine
        runs the encoder's reverse pass before this node (it was created earlier in the forward pass), and in a replayed
        hipGraph the reverse filter kernel then started ~75 us after its inputs existed, behind those launches."""
        desc = ctx.desc
        q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0, obs_TB, eps = ctx.saved
        ov = desc._ov
        ov.side.wait_event(ov.bsim_grads_ready)
        with ops.launch_on(ov.side):
            r = ops.filter_backward(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0,
                                    obs_TB, eps, ctx.filt, dlse=_cg(dlse), dFm=_cg(dFm), dlogW=_cg(dlogW),
                                    gbufs=ctx.gbufs, defer_wgrad=True)
        kernel_done = torch.cuda.Event()
        kernel_done.record(ov.side)              # the reverse kernel's own outputs (d mu2, d m0, scale sums) exist
        return {"r": r, "kernel_done": kernel_done, "grads": (dlse, dFm, dlogW)}

    @staticmethod
    def backward(ctx, dlse, dFm, dlogW, *_):
        desc = ctx.desc
        q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0, obs_TB, eps = ctx.saved
        ov = getattr(desc, "_ov", None)
        side = None if ov is None else ov.side
        ready = None if ov is None else ov.bsim_grads_ready
        if side is not None and ready is not None:
            # upstream gradients were produced by the backward-simulation node (event `ready`); everything issued
            # on the main stream since then (bsim weight gradients, hoisted backward, encoder BPTT) overlaps
            main = torch.cuda.current_stream()
            pre, ov.pre = ov.pre, None
            same = lambda a, b: (a is None and b is None) or (a is not None and b is not None
                                                              and a.data_ptr() == b.data_ptr() and a.shape == b.shape)
            if pre is None or not all(same(a, b) for a, b in zip(pre["grads"], (dlse, dFm, dlogW))):
                # not launched early, or the engine added further contributions to the upstream gradients
                pre = FilterFunction.launch_reverse_kernel(ctx, dlse, dFm, dlogW)
            r, kernel_done = pre["r"], pre["kernel_done"]
            # other writers of the flat-gradient slices the filter's weight gradients accumulate into (q1 / f, g):
            # the bsim weight gradients (event below) and, outside the default wiring, the hoisted f.mean(mu_0) on
            # the main stream -- only then wait for everything the main stream issued before this node (with the
            # second side stream in use there is no such writer, and waiting would serialise behind the encoder BPTT)
            if ov.side2 is None:
                side.wait_stream(main)
            if ov.pending_wgrad is not None:         # the backward simulation's weight gradients, deferred until now
                fn, ov.pending_wgrad = ov.pending_wgrad, None
                fn()
            if ov.bsim_wgrad_done is not None:       # (recorded on the second side stream when used)
                side.wait_event(ov.bsim_wgrad_done)
            # PSVO_FILTER_WGRAD_SPLIT=1: MLP_g's weight gradient on the second side stream beside MLP_q1's (they write
            # different slices of the flat gradient buffer; that stream already carries the backward simulation's weight
            # gradients, i.e. it is ordered after their write of g's slice and is joined below with `side`)
            split = (SPLIT_FILTER_WGRAD and ctx.gbufs is not None and DEFER_JOIN and ov.side2 is not None
                     and ov.bsim_wgrad_done is not None)
            after = kernel_done
            if split and not SPLIT_CROSS_WAIT:
                # `side` already waits for an event of `side2` (bsim_wgrad_done); a wait of side2 on an event of `side` would
                # close a cycle in the runtime's fork relation and hip::Stream::EndCapture (ROCm 7.2) recurses over that
                # relation without a visited set -- stack overflow, SIGSEGV (gpurun_out/r3/gdb_gsplit.log).  The kernel's
                # completion is therefore relayed through the main stream.
                main.wait_event(kernel_done)
                after = torch.cuda.Event()
                after.record(main)
            with ops.launch_on(side):
                r.pop("_wgrad")(ov.side2 if split else None, after)
            if ctx.gbufs is not None and DEFER_JOIN:
                # the weight gradients accumulate straight into the flat gradient buffer: nothing downstream on the main
                # stream reads them, so the main stream (hoisted q2 / q0 backward, scale gradients) continues as soon as
                # the reverse KERNEL is done and the weight-gradient launches overlap it; the caller joins the streams once,
                # after backward() and before it touches the gradients (autograd.join_deferred(): trainer, bench).
                # (The abort recorded in round 2 -- "MLP_g's weight gradient on the second side stream takes the process down
                #  inside hipStreamEndCapture" -- is explained: see `split` above and DESIGN.md section 1.)
                main.wait_event(kernel_done)
                _PENDING_JOIN.extend(st for st in (side, ov.side2) if st is not None)
            else:
                main.wait_stream(side)
                if ov.bsim_wgrad_done is not None:
                    main.wait_event(ov.bsim_wgrad_done)
            _used_on(main, r.values())
        else:
            r = ops.filter_backward(desc, q1, f, g, sig_q1, sig_q2, sig_f, sig_g, mu2, m0, sig0, fm0, fsig0,
                                    obs_TB, eps, ctx.filt, dlse=_cg(dlse), dFm=_cg(dFm), dlogW=_cg(dlogW),
                                    gbufs=ctx.gbufs)
        Dx, Dy, H = desc.Dx, desc.Dy, desc.H
        gfirst, gextra = _mlp_grads(r, ("gq1", None if desc.bootstrap else "gf", "gg"), ((Dx, Dx), (Dx, Dx), (Dx, Dy)), H,
                                    2 if desc.layers == 2 else 1, ctx.gbufs or (None, None, None))
        two_q, boot = bool(desc.two_q), bool(desc.bootstrap)
        # (one tensor passed as both m0 and fm0 -- the default wiring -- gets the summed gradient in d m0 from the kernel:
        #  returning the zero d fm0 as well would only make the engine launch an add; likewise sig0 / fsig0)
        same_m = m0.data_ptr() == fm0.data_ptr()
        same_s = sig0.data_ptr() == fsig0.data_ptr()
        return (None, None, None, None, None) + gfirst + (
            r["dsig_q1"], r["dsig_q2"] if two_q else None, None if boot else r["dsig_f"], r["dsig_g"],
            r["dmu2"] if two_q else None, r["dm0"], r["dsig0"], None if same_m else r["dfm0"],
            None if same_s else r["dfsig0"]) + gextra


class FilterCovFunction(torch.autograd.Function):
    """psvo_filter_forward_cov / psvo_filter_backward_cov: the forward filter with state-dependent diagonal scales
    (FLAGS.output_cov and FLAGS.diag_cov).

    apply(desc, obs_TB, eps, u, idx_in,
          q1 (W1, b1, W_mu, b_mu, W_sigma, b_sigma), f (6 tensors, None x 6 when desc.bootstrap), g (6 tensors),
          sigc_q1, sigc_f, sigc_g, mu2, sig2, m0, sig0, fm0, fsig0)
      -> lse (T,B), Fm, Fs (T,B,Dx,N), logW (T,B,N)   [differentiable]
         X, Xanc (T,B,Dx,N), idx (T,B,N) int32          [constants]
    One stream, gradients returned as tensors (no in-place accumulation into the flat buffer): this wiring is off the
    headline path."""

    @staticmethod
    def forward(ctx, desc, obs_TB, eps, u, idx_in, *t):
        ctx.set_materialize_grads(False)
        t = [_cf(v) for v in t]
        q1, f, g = tuple(t[0:6]), tuple(t[6:12]), tuple(t[12:18])
        if desc.bootstrap:
            f = None
        rest = t[18:27]
        filt = ops.filter_forward_cov(desc, q1, f, g, *rest, obs_TB, eps, u, idx_in)
        ctx.desc, ctx.filt = desc, _aliases(filt)
        ctx.saved = (q1, f, g, rest, obs_TB, eps)
        ctx.mark_non_differentiable(filt["X"], filt["Xanc"], filt["idx"])
        return filt["lse"], filt["Fm"], filt["Fs"], filt["logW"], filt["X"], filt["Xanc"], filt["idx"]

    @staticmethod
    def backward(ctx, dlse, dFm, dFs, dlogW, *_):
        desc = ctx.desc
        q1, f, g, rest, obs_TB, eps = ctx.saved
        r = ops.filter_backward_cov(desc, q1, f, g, *rest, obs_TB, eps, ctx.filt, dlse=_cg(dlse), dFm=_cg(dFm),
                                    dFs=_cg(dFs), dlogW=_cg(dlogW))
        two_q, boot = bool(desc.two_q), bool(desc.bootstrap)
        none6 = (None,) * 6
        return (None, None, None, None, None) + r["gq1"] + (none6 if boot else r["gf"]) + r["gg"] + (
            r["dsigc_q1"], None if boot else r["dsigc_f"], r["dsigc_g"],
            r["dmu2"] if two_q else None, r["dsig2"] if two_q else None,
            r["dm0"], r["dsig0"], r["dfm0"], r["dfsig0"])


def _filter_node_of(Fm, ov):
    """the FilterFunction node that produced `Fm` in this evaluation (its ctx), if the overlap wiring is on: the
    backward-simulation node may launch that node's reverse kernel early.  (A downstream reference only: this node
    already holds the filter node through its graph edge, so nothing new is kept alive.)"""
    node = getattr(Fm, "grad_fn", None)
    if ov is None or node is None or getattr(node, "desc", None) is None or getattr(node.desc, "_ov", None) is not ov:
        return None
    return node if hasattr(node, "filt") and hasattr(node, "saved") else None


class BsimFunction(torch.autograd.Function):
    """psvo_bsim_forward / psvo_bsim_backward.

    apply(desc, obs_TB, eps_b, u_b, sel_in, Fm, logW, lse,
          fW1, fb1, fW2, fb2, gW1, gb1, gW2, gb2, qW1, qb1, qW2, qb2,
          sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig)
      -> score (B,N) [differentiable]; bwX (T,B,Dx,N), flp, glp, Omega (T,B,N), sel (T,B,N) [constants]
    """

    @staticmethod
    def forward(ctx, desc, obs_TB, eps_b, u_b, sel_in, Fm, logW, lse, *t):
        t = [_cf(v) for v in t]
        f, g, q = _mlps(t)
        sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig = t[12:21]
        filt = {"X": None, "Fm": _cf(Fm), "logW": _cf(logW), "lse": _cf(lse)}
        ov = getattr(desc, "_ov", None)
        if ov is not None and ov.filter_done is not None:     # the filter ran on the side stream
            torch.cuda.current_stream().wait_event(ov.filter_done)
        ctx.filter_node = _filter_node_of(Fm, ov)
        ctx.gbufs = getattr(desc, "_gbufs", None)             # (f, g, q1_inv) flat-gradient slices or None
        need = any(ctx.needs_input_grad)
        bs = ops.bsim_forward(desc, {**filt, "X": filt["Fm"]}, f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit,
                              sig_init, imean, isig, obs_TB, eps_b, u_b, sel_in, save=need)
        ctx.desc, ctx.filt, ctx.bs = desc, filt, _aliases(bs)
        ctx.saved = (f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig, obs_TB, eps_b)
        ctx.mark_non_differentiable(bs["bwX"], bs["flp"], bs["glp"], bs["Omega"], bs["sel"])
        ctx.set_materialize_grads(False)      # (else the engine zero-fills a gradient for each of the five constants)
        return bs["score"], bs["bwX"], bs["flp"], bs["glp"], bs["Omega"], bs["sel"]

    @staticmethod
    def backward(ctx, dscore, *_):
        desc = ctx.desc
        if dscore is None:                    # (score unused by the loss)
            dscore = torch.zeros(desc.B, desc.N, device=ctx.filt["Fm"].device)
        f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig, obs_TB, eps_b = ctx.saved
        ov = getattr(desc, "_ov", None)

        def after_kernel(out):   # d Fm / d logW exist: the filter's reverse pass (side stream) may start
            if ov is not None:
                ov.bsim_grads_ready = torch.cuda.Event()
                ov.bsim_grads_ready.record()
                if PRELAUNCH and ctx.filter_node is not None and (ctx.needs_input_grad[5] or ctx.needs_input_grad[6]):
                    # (d lse = -sum_n d logW: zero analytically, rounding-sized numerically, passed on as autodiff of the
                    #  reference's graph would -- ops.bsim_backward / bsim_bwd_fold_finalize)
                    ov.pre = FilterFunction.launch_reverse_kernel(ctx.filter_node, out["dlse"], out["dFm"], out["dlogW"])
        ws = None if (ov is None or ctx.gbufs is None) else ov.side2
        defer = (DEFER_BSIM_WGRAD and DEFER_JOIN and PRELAUNCH and ws is not None and ctx.filter_node is not None
                 and (ctx.needs_input_grad[5] or ctx.needs_input_grad[6]))
        r = ops.bsim_backward(desc, ctx.filt, f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init,
                              imean, isig, obs_TB, eps_b, ctx.bs, _cg(dscore), gbufs=ctx.gbufs,
                              after_kernel=after_kernel, wgrad_stream=ws, defer_wgrad=defer)
        if defer:
            ready, issue_wgrad = ov.bsim_grads_ready, r.pop("_wgrad")

            def issue():
                ws.wait_event(ready)                 # (after the reverse kernel and its fold, not after what main issued since)
                with ops.launch_on(ws):
                    issue_wgrad()
                ov.bsim_wgrad_done = torch.cuda.Event()
                ov.bsim_wgrad_done.record(ws)
            ov.pending_wgrad = issue
            _PENDING_WORK.append(ov)                 # (join_deferred() issues it if the filter node never runs)
        elif ov is not None:
            ov.bsim_wgrad_done = torch.cuda.Event()
            ov.bsim_wgrad_done.record(ws if ws is not None else torch.cuda.current_stream())
        Dx, Dy, H = desc.Dx, desc.Dy, desc.H
        gfirst, gextra = _mlp_grads(r, ("gf", "gg", "gq1inv"), ((Dx, Dx), (Dx, Dy), (Dx, Dx)), H,
                                    2 if desc.layers == 2 else 1, ctx.gbufs or (None, None, None))
        dFm, dlogW = r["dFm"], r["dlogW"]
        ops.sum_chain_rows(r, desc.T, desc.B, Dx)       # (one reduction for d bmu2 / d minit / d imean)
        return (None, None, None, None, None, dFm, dlogW, r["dlse"] if ctx.needs_input_grad[7] else None) + gfirst + (
            r["dsig_f"], r["dsig_g"], r["dsig_q1inv"], r["dsig_bq2"], r["dbmu2"],
            r["dminit"], r["dsig_init"], r["dimean"], r["disig"]) + gextra


class BsimCovFunction(torch.autograd.Function):
    """psvo_bsim_forward_cov / psvo_bsim_backward_cov: the backward simulation with state-dependent diagonal scales.

    apply(desc, obs_TB, eps_b, u_b, sel_in, Fm, Fs, logW, lse,
          f (W1, b1, W_mu, b_mu, W_sigma, b_sigma), g (6 tensors), q1_inv (6 tensors),
          sigc_f, sigc_g, sigc_q1inv, bmu2, bsig2, minit, sinit, imean, isig)
      -> score (B,N) [differentiable]; bwX (T,B,Dx,N), flp, glp, Omega (T,B,N), sel (T,B,N) [constants]
    One stream; gradients returned as tensors."""

    @staticmethod
    def forward(ctx, desc, obs_TB, eps_b, u_b, sel_in, Fm, Fs, logW, lse, *t):
        t = [_cf(v) for v in t]
        f, g, q1_inv = tuple(t[0:6]), tuple(t[6:12]), tuple(t[12:18])
        rest = t[18:27]
        filt = {"Fm": _cf(Fm), "Fs": _cf(Fs), "logW": _cf(logW), "lse": _cf(lse)}
        bs = ops.bsim_forward_cov(desc, filt, f, g, q1_inv, *rest, obs_TB, eps_b, u_b, sel_in,
                                  save=any(ctx.needs_input_grad))
        ctx.desc, ctx.filt, ctx.bs = desc, filt, _aliases(bs)
        ctx.saved = (f, g, q1_inv, rest, obs_TB, eps_b)
        ctx.mark_non_differentiable(bs["bwX"], bs["flp"], bs["glp"], bs["Omega"], bs["sel"])
        return bs["score"], bs["bwX"], bs["flp"], bs["glp"], bs["Omega"], bs["sel"]

    @staticmethod
    def backward(ctx, dscore, *_):
        desc = ctx.desc
        f, g, q1_inv, rest, obs_TB, eps_b = ctx.saved
        r = ops.bsim_backward_cov(desc, ctx.filt, f, g, q1_inv, *rest, obs_TB, eps_b, ctx.bs, _cg(dscore).float())
        return (None, None, None, None, None, r["dFm"], r["dFs"], r["dlogW"], r["dlse"]) + r["gf"] + r["gg"] + r["gq1inv"] + (
            r["dsigc_f"], r["dsigc_g"], r["dsigc_q1inv"], r["dbmu2"], r["dbsig2"], r["dminit"], r["dsinit"], r["dimean"],
            r["disig"])


class BsimWRCovFunction(torch.autograd.Function):
    """psvo_bsimwr_forward_cov / psvo_bsimwr_backward_cov: PSVOwR with state-dependent diagonal scales.

    apply(desc, obs_TB, eps_b, u_b, u_r, sel_in, anc_in, Fm, Fs, logW, lse, f (6), g (6), q1_inv (6),
          sigc_f, sigc_g, sigc_q1inv, bmu2, bsig2, minit, sinit, imean, isig)
      -> lseW (T,B) [differentiable]; bwXanc, bwX (T,B,Dx,N), bwW (T,B,N), sel, anc (T,B,N) [constants]"""

    @staticmethod
    def forward(ctx, desc, obs_TB, eps_b, u_b, u_r, sel_in, anc_in, Fm, Fs, logW, lse, *t):
        t = [_cf(v) for v in t]
        f, g, q1_inv = tuple(t[0:6]), tuple(t[6:12]), tuple(t[12:18])
        rest = t[18:27]
        filt = {"Fm": _cf(Fm), "Fs": _cf(Fs), "logW": _cf(logW), "lse": _cf(lse)}
        bs = ops.bsimwr_forward_cov(desc, filt, f, g, q1_inv, *rest, obs_TB, eps_b, u_b=u_b, u_r=u_r, sel_in=sel_in,
                                    anc_in=anc_in, save=any(ctx.needs_input_grad))
        ctx.desc, ctx.filt, ctx.bs = desc, filt, _aliases(bs)
        ctx.saved = (f, g, q1_inv, rest, obs_TB, eps_b)
        ctx.mark_non_differentiable(bs["bwXanc"], bs["bwX"], bs["bwW"], bs["sel"], bs["anc"])
        ctx.set_materialize_grads(False)
        return bs["lseW"], bs["bwXanc"], bs["bwX"], bs["bwW"], bs["sel"], bs["anc"]

    @staticmethod
    def backward(ctx, dlseW, *_):
        desc = ctx.desc
        if dlseW is None:
            dlseW = torch.zeros(desc.T, desc.B, device=ctx.filt["Fm"].device)
        f, g, q1_inv, rest, obs_TB, eps_b = ctx.saved
        r = ops.bsimwr_backward_cov(desc, ctx.filt, f, g, q1_inv, *rest, obs_TB, eps_b, ctx.bs, _cg(dlseW).float())
        return (None,) * 7 + (r["dFm"], r["dFs"], r["dlogW"], r["dlse"]) + r["gf"] + r["gg"] + r["gq1inv"] + (
            r["dsigc_f"], r["dsigc_g"], r["dsigc_q1inv"], r["dbmu2"], r["dbsig2"], r["dminit"], r["dsinit"], r["dimean"],
            r["disig"])


def _note_exchange(desc, ws, bit):
    """OR the launch's exchange-timeout flag (last word of its workspace, cleared by every launch) into the sticky
    device word the objective keeps across launches (PSVOwR.check_exchange reads it): bit 1 = forward, bit 2 = reverse."""
    sticky = getattr(desc, "_sticky", None)
    if sticky is not None:
        sticky.bitwise_or_((ws[-1:].view(torch.int32) != 0).to(torch.int32) * bit)


class BsimWRFunction(torch.autograd.Function):
    """psvo_bsimwr_forward / psvo_bsimwr_backward (PSVOwR: cross-chain resampling, per-step ELBO).

    apply(desc, obs_TB, eps_b, u_b, u_r, sel_in, anc_in, Fm, logW, lse,
          fW1, fb1, fW2, fb2, gW1, gb1, gW2, gb2, qW1, qb1, qW2, qb2,
          sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig)
      -> lseW (T,B) [differentiable]; bwXanc, bwX (T,B,Dx,N), bwW (T,B,N), sel, anc (T,B,N) [constants]
    """

    @staticmethod
    def forward(ctx, desc, obs_TB, eps_b, u_b, u_r, sel_in, anc_in, Fm, logW, lse, *t):
        t = [_cf(v) for v in t]
        f, g, q = _mlps(t)
        sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig = t[12:21]
        filt = {"Fm": _cf(Fm), "logW": _cf(logW), "lse": _cf(lse)}
        ov = getattr(desc, "_ov", None)
        if ov is not None and ov.filter_done is not None:
            torch.cuda.current_stream().wait_event(ov.filter_done)
        ctx.filter_node = _filter_node_of(Fm, ov)
        ctx.gbufs = getattr(desc, "_gbufs", None)
        need = any(ctx.needs_input_grad)
        bs = ops.bsimwr_forward(desc, filt, f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean,
                                isig, obs_TB, eps_b, u_b=u_b, u_r=u_r, sel_in=sel_in, anc_in=anc_in, save=need)
        ctx.desc, ctx.filt, ctx.bs = desc, filt, _aliases(bs)
        _note_exchange(desc, bs["ws"], 1)
        ctx.saved = (f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig, obs_TB, eps_b)
        ctx.mark_non_differentiable(bs["bwXanc"], bs["bwX"], bs["bwW"], bs["sel"], bs["anc"], bs["ws"])
        ctx.set_materialize_grads(False)      # (else the engine zero-fills a gradient for each of the six constants)
        return bs["lseW"], bs["bwXanc"], bs["bwX"], bs["bwW"], bs["sel"], bs["anc"], bs["ws"]

    @staticmethod
    def backward(ctx, dlseW, *_):
        desc = ctx.desc
        if dlseW is None:                     # (lseW unused by the loss)
            dlseW = torch.zeros(desc.T, desc.B, device=ctx.filt["Fm"].device)
        f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init, imean, isig, obs_TB, eps_b = ctx.saved
        ov = getattr(desc, "_ov", None)

        def after_kernel(out):
            if ov is not None:
                ov.bsim_grads_ready = torch.cuda.Event()
                ov.bsim_grads_ready.record()
                if PRELAUNCH and ctx.filter_node is not None and (ctx.needs_input_grad[7] or ctx.needs_input_grad[8]):
                    ov.pre = FilterFunction.launch_reverse_kernel(ctx.filter_node, out["dlse"], out["dFm"], out["dlogW"])
        ws = None if (ov is None or ctx.gbufs is None) else ov.side2
        r = ops.bsimwr_backward(desc, ctx.filt, f, g, q, sig_f, sig_g, sig_q1inv, sig_bq2, bmu2, minit, sig_init,
                                imean, isig, obs_TB, eps_b, ctx.bs, _cg(dlseW), gbufs=ctx.gbufs,
                                after_kernel=after_kernel, wgrad_stream=ws)
        if ov is not None:
            ov.bsim_wgrad_done = torch.cuda.Event()
            ov.bsim_wgrad_done.record(ws if ws is not None else torch.cuda.current_stream())
        Dx, Dy, H = desc.Dx, desc.Dy, desc.H
        gfirst, gextra = _mlp_grads(r, ("gf", "gg", "gq1inv"), ((Dx, Dx), (Dx, Dy), (Dx, Dx)), H,
                                    2 if desc.layers == 2 else 1, ctx.gbufs or (None, None, None))
        ops.sum_chain_rows(r, desc.T, desc.B, Dx)
        _note_exchange(desc, r["ws"], 2)          # (after the weight-gradient launches: see ops.sum_chain_rows)
        return (None,) * 7 + (r["dFm"], r["dlogW"], r["dlse"]) + gfirst + (
            r["dsig_f"], r["dsig_g"], r["dsig_q1inv"], r["dsig_bq2"], r["dbmu2"],
            r["dminit"], r["dsig_init"], r["dimean"], r["disig"]) + gextra


class RowsMLPFunction(torch.autograd.Function):
    """psvo_rows_mlp_forward / psvo_rows_mlp_backward: a hoisted one-hidden-layer MLP over (R, Din) rows.
    apply(gbuf, X, W1, b1, W2, b2) -> (R, Dout); gbuf = slice of the flat gradient buffer to accumulate
    [dW1|db1|dW2|db2] into directly, or None."""

    @staticmethod
    def forward(ctx, gbuf, X, W1, b1, W2, b2):
        ctx.gbuf = gbuf
        X = _cf(X)
        w = tuple(_cf(v) for v in (W1, b1, W2, b2))
        ctx.saved = (X, w)
        ctx.need_dX = ctx.needs_input_grad[1]
        return ops.rows_mlp_forward(X, w)

    @staticmethod
    def backward(ctx, dOut):
        X, w = ctx.saved
        Din, H = w[0].shape
        Dout = w[2].shape[1]
        dX, g = ops.rows_mlp_backward(X, _cg(dOut).float(), w, need_dX=ctx.need_dX, grad=ctx.gbuf)
        if ctx.gbuf is not None:
            return None, dX, None, None, None, None
        return (None, dX) + tuple(ops.split_mlp_grad(g, Din, H, Dout))


class DenseFunction(torch.autograd.Function):
    """psvo_dense_forward / psvo_dense_backward: one Dense layer (keras layout, optional relu) over (R, Din) rows on
    v_mfma_f32_16x16x4_f32.  apply(X, W, b, relu) -> (R, Dout)."""

    @staticmethod
    def forward(ctx, X, W, b, relu):
        X, W, b = _cf(X), _cf(W), _cf(b)
        Y = ops.dense_forward(X, W, b, relu)
        ctx.relu, ctx.need_dX = bool(relu), ctx.needs_input_grad[0]
        ctx.saved = (X, W, Y.detach() if relu else None)     # (an alias: see _aliases)
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W, Y = ctx.saved
        dX, dW, db = ops.dense_backward(X, Y, _cg(dY).float(), W, ctx.relu, need_dX=ctx.need_dX)
        return dX, dW, db, None


class ElboBsimFunction(torch.autograd.Function):
    """psvo_elbo_bsim_mean / psvo_elbo_bsim_mean_backward: mean_b [logsumexp_n score - log N] (PSVO.py:52-67)
    as one launch each way instead of torch's logsumexp/mean chain between the two bsim kernels."""

    @staticmethod
    def forward(ctx, desc, score):
        score = _cf(score)
        ctx.desc, ctx.score = desc, score
        return ops.elbo_bsim_mean(desc, score)

    @staticmethod
    def backward(ctx, dz):
        return None, ops.elbo_bsim_mean_backward(ctx.desc, ctx.score, dz)


class BiLSTMFunction(torch.autograd.Function):
    """psvo_bilstm_forward / psvo_bilstm_backward: one bidirectional LSTMBlockCell layer."""

    @staticmethod
    def forward(ctx, gbufs, x, W_fw, b_fw, W_bw, b_bw):
        ctx.gbufs = gbufs                                      # (fw, bw) flat-gradient slices or None
        x, W_fw, b_fw, W_bw, b_bw = (_cf(v) for v in (x, W_fw, b_fw, W_bw, b_bw))
        if any(ctx.needs_input_grad):
            out, cs, gates = ops.bilstm_forward(x, W_fw, b_fw, W_bw, b_bw, save=True)
            ctx.saved = (x, W_fw, W_bw, out.detach(), cs, gates)     # (an alias of the output: see _aliases)
        else:
            out = ops.bilstm_forward(x, W_fw, b_fw, W_bw, b_bw)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W_fw, W_bw, out, cs, gates = ctx.saved
        dx, dWf, dbf, dWb, dbb = ops.bilstm_backward(x, W_fw, W_bw, out, cs, gates, _cg(dout), gbufs=ctx.gbufs,
                                                     need_dx=ctx.needs_input_grad[1])
        return None, dx, dWf, dbf, dWb, dbb


class SigmaFunction(torch.autograd.Function):
    """sigma of EVERY distribution in one launch (tf_mvn.get_sigma, reference src/distribution/mvn.py:80-90).

    The raw vectors are contiguous in the flat parameter buffer (`blk` from FlatParams._annotate); the
    gradient is accumulated straight into the flat gradient buffer, so the parameters only anchor the graph.
    """

    @staticmethod
    def forward(ctx, blk, *params):
        ctx.blk = blk
        return ops.sigma_forward(blk["raw"], blk["mins"])

    @staticmethod
    def backward(ctx, dsig):
        blk = ctx.blk
        ops.sigma_backward(blk["raw"], blk["mins"], _cg(dsig), blk["grad"], accumulate=True)
        return (None,) + (None,) * len(blk["dists"])
