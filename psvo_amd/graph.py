"""hipGraph capture of a whole training / evaluation step.

One PSVO training step is ~90 launches on two streams (7 long persistent kernels plus the small
noise, reduction and weight-gradient kernels around them); the Python thread that issues them needs
~2.7 ms per C* step and more than the GPU time of the small AESMC-sized steps.  Capturing the step
once and replaying it removes the host from the loop (MI355X guide: "capture launch-bound inner
loops in hipGraphs").  Every replay executes the same kernels on the same
buffers with fresh random draws (the generators are registered with the graph, so their Philox
offsets advance per replay)."""
import torch


class GraphedStep(object):
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

    def __call__(self):
        self.graph.replay()
        return self.out
