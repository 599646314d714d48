"""hipGraph capture of a whole training / evaluation step.

One PSVO training step is ~150 launches (7 long persistent kernels plus the small hoisted-MLP,
noise and reduction kernels around them); at C* the persistent kernels take ~7.7 ms and the
gaps between the many small launches another ~1 ms when issued eagerly from Python.  Capturing
the step once and replaying it removes the host from the loop (MI355X guide: "capture
launch-bound inner loops in hipGraphs").  Every replay executes the same kernels on the same
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
