"""hipGraph capture of a whole training / evaluation step.

One PSVO training step is ~90 launches on two streams (7 long persistent kernels plus the small
noise, reduction and weight-gradient kernels around them); the Python thread that issues them needs
~2.7 ms per C* step and more than the GPU time of the small AESMC-sized steps.  Capturing the step
once and replaying it removes the host from the loop (MI355X guide: "capture launch-bound inner
loops in hipGraphs").  Every replay executes the same kernels on the same
buffers with fresh random draws (the generators are registered with the graph, so their Philox
offsets advance per replay)."""
import os

import torch


class GraphedStep(object):
    # Replays the host may have in flight.  A replay costs the launching thread ~20 us; a small step (the reference notebook's
    # B = 1, N = 16: 0.3 ms of GPU time) lets it run hundreds of launches of the SAME executable graph ahead of the device.
    # Twice in ~25 runs of the GPU suite the process died with SIGSEGV inside hipGraphLaunch in exactly that test (12 000
    # back-to-back replays; gpurun_out/r3/t1.log, gpurun_out/r3b/full2.log; not reproducible under rocgdb) and never anywhere
    # else, so the run-ahead is bounded: every PERIOD-th replay records an event, and before recording the next one the host
    # waits for the previous one -- at most 2 PERIOD replays are ever in flight and the device always has >= PERIOD queued.
    # Same-box A/B at the smallest step (C2, AESMC, 0.9 ms): 0.8995 ms unbounded against 0.9015 ms with one event per 16
    # replays (+0.2 %).
    PERIOD = int(os.environ.get("PSVO_GRAPH_PERIOD", "16"))      # (0 = unbounded, for A/B)

    def __init__(self, fn, generators=(), warmup=3):
        """fn: a closure over STATIC tensors (inputs are updated in place by the caller) that
        returns a tensor or tuple of tensors living in the graph's memory pool."""
        self.fn = fn
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        for g in generators:
            if g is not None:
                self.graph.register_generator_state(g)
        with torch.cuda.graph(self.graph):
            self.out = fn()
        self._mark = None
        self._n = 0

    def __call__(self):
        self.graph.replay()
        self._n += 1
        if self.PERIOD > 0 and self._n % self.PERIOD == 0:
            if self._mark is not None:
                self._mark.synchronize()
            else:
                self._mark = torch.cuda.Event()
            self._mark.record()
        return self.out
