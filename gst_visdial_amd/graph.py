"""hipGraph replay of a whole train step.

A step of this path is ~1000 kernel launches (two HIP streams); issued from Python they cost ~18 ms of host time,
about as much as the GPU work itself.  `GraphedStep` runs the step function eagerly a few times (so every cached
device table, LDS attribute and arena chunk exists), captures one invocation into a hipGraph and replays it.
Requirements the engine meets: static addresses (bump arena, flat parameter buffers), no host synchronisation in
the step, device-resident mutable state (dropout offset, AdamW step counter), inputs refreshed IN PLACE by the
caller (`tensor.copy_(new)`), learning-rate tables uploaded outside the graph (`FusedAdamW.upload_lr`)."""
import torch


class GraphedStep(object):
    def __init__(self, step_fn, warmup=2):
        self.step_fn = step_fn
        for _ in range(warmup):
            self.out = step_fn()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread polls events concurrently: keep the capture's error mode local
        # to this thread so that polling cannot invalidate it
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local" if dist_on else "global"):
            self.out = step_fn()

    def __call__(self):
        self.graph.replay()
        return self.out
