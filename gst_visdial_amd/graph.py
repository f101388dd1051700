"""hipGraph replay of a whole train step.

A step of this path is ~1000 kernel launches (two HIP streams); issued from Python they cost ~18 ms of host time,
about as much as the GPU work itself.  `GraphedStep` runs the step function eagerly a few times (so every cached
device table, LDS attribute and arena chunk exists), captures one invocation into a hipGraph and replays it.
Requirements the engine meets: static addresses (bump arena, flat parameter buffers), no host synchronisation in
the step, device-resident mutable state (dropout offset, AdamW step counter), inputs refreshed IN PLACE by the
caller (`tensor.copy_(new)`), learning-rate tables uploaded outside the graph (`FusedAdamW.upload_lr`)."""
import torch


def dist_alive():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def quiesce_before_capture():
    """Call right before a stream capture.  c10d's watchdog thread polls the events of eager collectives until it has seen
    them complete (about every 100 ms).  If a capture starts while such a Work is still on its list, the poll can hit an
    event of the capturing stream: the capture is invalidated (hipErrorStreamCaptureInvalidated) and the watchdog's own
    exception aborts the process -- observed in ~1 of 6 runs, 0 of 24 with this wait."""
    torch.cuda.synchronize()
    if dist_alive():
        import time
        time.sleep(2.0)


def capture_error_mode():
    """With a process group alive other threads poll events concurrently: keep the capture's error mode thread-local."""
    return "thread_local" if dist_alive() else "global"


class GraphedStep(object):
    def __init__(self, step_fn, warmup=2):
        self.step_fn = step_fn
        for _ in range(warmup):
            self.out = step_fn()
        quiesce_before_capture()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode()):
            self.out = step_fn()

    def __call__(self):
        self.graph.replay()
        return self.out
