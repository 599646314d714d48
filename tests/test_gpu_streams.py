"""Allocation discipline of the side-stream launches (the PSVOwR gradient flake of round 1).

The filter kernels are issued on a side stream while the main stream is still busy.  PyTorch's caching allocator hands a
freed block back to the NEXT request of the SAME stream's pool without waiting for the work queued on that stream, which is
correct only as long as the block is reused on that stream.  A wrapper that allocates its outputs from the MAIN stream's pool
while launching on the SIDE stream can therefore be handed a block that queued main-stream work is still reading or writing
(the side stream is not ordered after that work).  The rule the wrappers follow: buffers allocated under ops.launch_on(stream)
come from THAT stream's pool, inputs that arrive from another stream are record_stream()'ed when their pointer is taken, and
whoever consumes the outputs on another stream records them there."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _busy(stream, ms=30.0):
    """queue roughly `ms` of work on `stream` and return the tensors it uses"""
    a = torch.randn(4096, 4096, device="cuda")
    with torch.cuda.stream(stream):
        b = a
        for _ in range(max(1, int(ms / 0.25))):
            b = b @ a
            b = b / b.norm()
    return a, b


def test_side_stream_wrappers_do_not_take_blocks_of_the_main_pool(built_lib):
    from psvo_amd import ops
    from psvo_amd.autograd import side_stream
    main, side = torch.cuda.current_stream(), side_stream("cuda:0")
    torch.cuda.synchronize()
    R, Din, H, Dout = 1 << 16, 64, 32, 2
    X = torch.randn(R, Din, device="cuda")
    w = (torch.randn(Din, H, device="cuda"), torch.zeros(H, device="cuda"), torch.randn(H, Dout, device="cuda"),
         torch.zeros(Dout, device="cuda"))
    # a block of exactly the size the wrapper will ask for, written by queued main-stream work, then freed
    scratch = torch.empty(R, Dout, device="cuda")
    freed_ptr = scratch.data_ptr()
    keep = _busy(main)                        # main stream is busy for tens of milliseconds ...
    scratch.copy_(X[:, :Dout])                # ... and this write to `scratch` is queued behind it
    del scratch                               # back to the MAIN stream's pool while the write is still pending
    assert not main.query(), "the main stream must still be busy for this test to mean anything"
    side.wait_stream(main) if False else None  # (deliberately NOT ordered after the main stream)
    with ops.launch_on(side):
        out = ops.rows_mlp_forward(X, w)       # allocates (R, Dout) under launch_on(side)
    assert out.data_ptr() != freed_ptr, "side-stream wrapper reused a block the main stream has pending work on"
    # the same request on the main stream does get that block back: the scenario was real
    again = torch.empty(R, Dout, device="cuda")
    assert again.data_ptr() == freed_ptr
    torch.cuda.synchronize()
    del keep
    ref = torch.relu(X @ w[0] + w[1]) @ w[2] + w[3]
    assert torch.allclose(out, ref, atol=1e-3, rtol=1e-4)


def test_cross_stream_tensors_are_recorded(built_lib, monkeypatch):
    """every tensor that crosses between the main and a side stream inside one PSVOwR training evaluation is
    record_stream()'ed on the consuming stream: inputs of side-stream launches when their pointer is taken (ops._ptr), outputs
    of side-stream launches before the main stream consumes them (autograd._used_on)"""
    from psvo_amd import autograd, ops
    from tests.test_gpu_parity import _setup
    seen = []
    orig = torch.Tensor.record_stream

    def spy(self, stream):
        seen.append((self.data_ptr(), stream.cuda_stream))
        return orig(self, stream)
    monkeypatch.setattr(torch.Tensor, "record_stream", spy)
    FLAGS, model, smc, obs, noise = _setup("PSVOwR", 2, 6, 16, 4, 2, 1, 16, True, True, seed=1)
    smc.generator = torch.Generator(device="cuda").manual_seed(0)
    side = autograd.side_stream(obs.device if obs.is_cuda else "cuda:0").cuda_stream
    main = torch.cuda.current_stream().cuda_stream
    z, log = smc.get_log_ZSMC(obs.float().cuda(), None)
    z.backward()
    torch.cuda.synchronize()
    on_side = {p for p, s in seen if s == side}
    on_main = {p for p, s in seen if s == main}
    filt = log["filter"]
    # outputs of the side-stream filter are consumed by the main-stream backward simulation
    for k in ("Fm", "logW", "lse", "X"):
        assert filt[k].data_ptr() in on_main, "filter output %s not recorded on the consuming (main) stream" % k
    # inputs of the side-stream filter were produced on the main stream
    for k in ("eps", "u"):
        assert filt[k].data_ptr() in on_side, "filter input %s not recorded on the side stream" % k
    assert len(on_side) >= 8 and len(on_main) >= 4


@pytest.mark.timeout(900)
def test_local_step_captures_without_flat_parameters():
    """hipGraph capture of objective + reverse pass when the gradients return through autograd (no FlatParams) and when
    the per-particle MLPs run on padded copies: after an eager evaluation on the null stream the capture must neither
    leave a stream unjoined nor crash (it did until the autograd Functions stopped storing their own outputs on ctx, a
    reference cycle that kept the previous graph and its gradient accumulators alive).  One child process per arm."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("capture_probe", os.path.join(ROOT, "tools", "capture_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    bad = probe.run_all(["aesmc+noflat", "psvo+overlap+noflat", "psvo+padded+overlap+flat", "psvowr+overlap+noflat"])
    assert not bad, bad


def test_training_step_gradients_are_bit_reproducible(built_lib):
    """No float atomics across waves are left on the N <= 256 path (round 3: the resampling gather's reverse pass scatters
    into one LDS copy per wave and the parent adds the copies in wave order), so the gradient of one training step is the same
    bit pattern from run to run, issued eagerly or replayed from the captured hipGraph."""
    from tests import test_gpu_parity as TP
    # 3 sequences x 128 particles: eight waves per sequence in the filter kernels, children of one parent spread over waves
    _reproducible(TP._setup("PSVO", 3, 12, 128, 8, 2, 1, 32, True, True, seed=11))
    # ... and N = 200 / 256: one lane per particle, four waves per sequence (one scatter copy per wave as well; above 256 the
    # reverse filter keeps its float atomics: the copies cost 3 % of the C5 step)
    _reproducible(TP._setup("PSVO", 2, 8, 200, 4, 2, 1, 32, True, True, seed=12))
    _reproducible(TP._setup("AESMC", 2, 8, 256, 1, 4, 1, 32, False, True, seed=13))


def _reproducible(setup):
    import torch
    from tests import helpers as Hh
    from psvo_amd import autograd
    from psvo_amd.graph import GraphedStep
    from psvo_amd.optim import FlatParams
    FLAGS, model, smc, obs, noise = setup
    nz = Hh.noise_to_hip(noise, "cuda")
    flat = FlatParams(model)
    obs_c = obs.float().cuda()

    def local():
        flat.zero_grad()
        z, _ = smc.get_log_ZSMC(obs_c, None, noise=nz)
        with autograd.deferred_join():
            z.backward()
        return z.detach()
    grads = []
    for _ in range(4):
        z = local()
        torch.cuda.synchronize()
        grads.append((flat.grad.clone(), z.clone()))
    for g, z in grads[1:]:
        assert torch.equal(g, grads[0][0]) and torch.equal(z, grads[0][1])
    step = GraphedStep(local)
    for _ in range(3):
        z = step()
        torch.cuda.synchronize()
        assert torch.equal(flat.grad, grads[0][0]) and torch.equal(z, grads[0][1])
    assert float(grads[0][0].abs().max()) > 0
