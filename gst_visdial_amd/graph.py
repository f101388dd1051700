"""hipGraph replay of a whole train step.

A step of this path is ~1000 kernel launches (two HIP streams); issued from Python they cost ~18 ms of host time,
about as much as the GPU work itself.  `GraphedStep` runs the step function eagerly a few times (so every cached
device table, LDS attribute and arena chunk exists), captures one invocation into a hipGraph and replays it.
Requirements the engine meets: static addresses (bump arena, flat parameter buffers), no host synchronisation in
the step, device-resident mutable state (dropout offset, AdamW step counter), inputs refreshed IN PLACE by the
caller (`tensor.copy_(new)`), learning-rate tables uploaded outside the graph (`FusedAdamW.upload_lr`).

Every stream capture of this package goes through `capture()` below, which makes the two things that may not happen
while a capture is open impossible instead of unlikely:

  * a cyclic-GC pass.  torch >= 2.9 no longer calls gc.collect() in torch.cuda.graph.__enter__; an automatic pass in the
    middle of a capture that finds a dead object graph owning CUDAGraphs / a private memory pool (e.g. the engine of an
    earlier model with a captured decode session) runs ~CUDAGraph -> pool release -> hipFree under an open "global" mode
    capture: the call is refused, the destructor throws, std::terminate -> SIGABRT (round 1's GPUTEST abort;
    tools/repro_gc_capture.py reproduces it on demand).  Here: collect BEFORE the capture, collector off during it.
  * c10d's watchdog thread polling the event of a not-yet-retired eager collective (see quiesce_before_capture).
"""
import contextlib
import gc
import os
import pickle
import time

import torch


def dist_alive():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def enable_watchdog_introspection():
    """Call BEFORE init_process_group: turns c10d's flight recorder on (a small ring), which is the only public window on
    what the watchdog thread still has on its list; `quiesce_before_capture` then waits on facts instead of on a timer."""
    os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "512")


def _active_collectives():
    """Number of collectives the watchdog has not retired yet, or None when the flight recorder is off / unavailable."""
    try:
        from torch._C._distributed_c10d import _dump_nccl_trace
        if int(os.environ.get("TORCH_NCCL_TRACE_BUFFER_SIZE", os.environ.get("TORCH_FR_BUFFER_SIZE", "0"))) <= 0:
            return None
        tr = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=True))
        ent = tr.get("entries") if isinstance(tr, dict) else None
        return None if ent is None else len(ent)
    except Exception:
        return None


_WARNED_TIMER = False
LAST_QUIESCE = [None]      # how the most recent capture waited for c10d's watchdog: "no-process-group" | "drained" | "timer"


def quiesce_before_capture(timeout=20.0):
    LAST_QUIESCE[0] = _quiesce(timeout)
    return LAST_QUIESCE[0]


def _quiesce(timeout):
    """Call right before a stream capture.  c10d's watchdog thread polls the events of eager collectives until it has seen
    them complete (a pass every ~100 ms).  If a capture starts while such a Work is still on its list, the poll can hit an
    event of the capturing stream: the capture is invalidated (hipErrorStreamCaptureInvalidated) and the watchdog's own
    exception aborts the process (round 1: 1 run in 6).  Deterministic form: (1) device idle -> every eager collective has
    completed; (2) wait until the flight recorder reports no un-retired collective, i.e. the watchdog has popped them all.
    Without the recorder (TORCH_NCCL_TRACE_BUFFER_SIZE unset before init_process_group) fall back to a 2 s sleep (twenty watchdog periods, round 1's validated wait) and warn once."""
    torch.cuda.synchronize()
    if not dist_alive():
        return "no-process-group"
    t0 = time.time()
    n = _active_collectives()
    if n is None:
        # no flight recorder (TORCH_NCCL_TRACE_BUFFER_SIZE was not set before init_process_group, or this torch's
        # _dump_nccl_trace differs): round 1's validated wait -- 20 watchdog periods; 0.5 s let the 1-in-6 abort come back
        global _WARNED_TIMER
        if not _WARNED_TIMER:
            _WARNED_TIMER = True
            import warnings
            warnings.warn("gst_visdial_amd.graph: c10d flight recorder unavailable -- waiting 2 s before the stream capture "
                          "instead of draining the watchdog deterministically; call graph.enable_watchdog_introspection() "
                          "before init_process_group")
        time.sleep(2.0)
        return "timer"
    while n > 0:
        if time.time() - t0 > timeout:
            raise RuntimeError("c10d watchdog still holds %d un-retired collectives after %.0f s; refusing to capture" % (n, timeout))
        time.sleep(0.01)
        n = _active_collectives()
    time.sleep(0.12)          # one more watchdog period: the pass that retired the last Work has left the event API
    return "drained"


def capture_error_mode():
    """With a process group alive other threads poll events concurrently: keep the capture's error mode thread-local."""
    return "thread_local" if dist_alive() else "global"


_GC_DEPTH = [0]


@contextlib.contextmanager
def gc_quiet():
    """Collect now, then keep the cyclic collector off for the duration (re-entrant: an inner use is a no-op, which is how
    a decode session pays for ONE collection although it captures ~20 graphs).  The collection also runs when the caller has
    the collector switched off: the dead cycles must be gone before the capture opens, whoever would have collected them."""
    outer = _GC_DEPTH[0] == 0
    _GC_DEPTH[0] += 1
    was_enabled = gc.isenabled()
    if outer:
        gc.collect()
        gc.disable()
    try:
        yield
    finally:
        _GC_DEPTH[0] -= 1
        if outer and was_enabled:
            gc.enable()


@contextlib.contextmanager
def capture(graph, pool=None, quiesce=True):
    """The only way this package opens a stream capture (see the module docstring).  `quiesce=False`: the caller has
    just captured another graph and issued no collective since (the per-position graphs of a decode session)."""
    if quiesce:
        quiesce_before_capture()
    with gc_quiet():
        kw = dict(capture_error_mode=capture_error_mode())
        if pool is not None:
            kw["pool"] = pool
        with torch.cuda.graph(graph, **kw):
            yield graph


class GraphedStep(object):
    def __init__(self, step_fn, warmup=2):
        self.step_fn = step_fn
        for _ in range(warmup):
            self.out = step_fn()
        self.graph = torch.cuda.CUDAGraph()
        with capture(self.graph):
            self.out = step_fn()

    def __call__(self):
        self.graph.replay()
        return self.out


class SegmentedStep(object):
    """The step as a SEQUENCE of hipGraphs with the collectives issued eagerly between them -- the fall-back for a refused
    whole-step capture at N > 1 (RCCL inside a stream capture has only ever been seen working with ONE rank on this stack), so that
    the first contact with 8 GPUs does not depend on one capture succeeding.  Eager issue costs ~19 ms of host time per step for
    ~12 ms of GPU work; this form costs the host one replay per gradient slice plus the slices' collectives and updates (~40
    calls), with the same kernels on the device.

    How: `pipe.segmenter = self` while `step_fn` runs under a capture that this object opens by hand.  At every slice boundary the
    engine joins its streams into the capture's origin stream (Engine._emit) and BackwardPipeline.run_slice calls `cut(fn)`: the
    open capture ends (one more graph), `fn` -- fork to the communication stream, cast, all-reduce / reduce-scatter, AdamW, record
    the tail event -- is filed as an eager item and NOT run (nothing has executed yet: a collective here would reduce garbage),
    and the next capture opens from the same memory pool.  Replay = the items in order; a last item joins the communication stream.
    What is lost against the whole-step graph: a slice's weight-gradient launch runs in line on the main stream instead of
    beside the backward chain (zero-sum at full-chip kernels), and the update of slice k is issued by the
    host while segment k + 1 is already replaying.  Same requirements as GraphedStep (static addresses, inputs refreshed in place)."""

    def __init__(self, step_fn, pipe, warmup=2):
        if pipe is None or not pipe.collective:
            raise ValueError("SegmentedStep is the N > 1 fall-back: it needs a BackwardPipeline with a collective")
        self.pipe, self.items, self._cur, self.pool = pipe, [], None, None
        for _ in range(warmup):
            self.out = step_fn()
        quiesce_before_capture()
        # "relaxed": loss.backward() runs the tape on autograd's device thread, so the cuts -- hipStreamEndCapture / BeginCapture --
        # happen on a different thread than the first BeginCapture; only a relaxed-mode capture may be ended from another thread
        # (hipErrorStreamCaptureWrongThread otherwise).  Relaxed also lets c10d's watchdog query its events meanwhile.
        self._mode = "relaxed"
        self.stream = torch.cuda.Stream()
        pipe.segmenter = self
        try:
            with gc_quiet():
                torch.cuda.synchronize()
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    self._open()
                    try:
                        self.out = step_fn()
                    except BaseException:
                        self._abort()
                        raise
                    self._close()
                torch.cuda.current_stream().wait_stream(self.stream)
        finally:
            pipe.segmenter = None
        self.items.append(("call", pipe.segment_join))
        self.n_graphs = sum(1 for k, _ in self.items if k == "graph")

    def _open(self):
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        g.capture_begin(pool=self.pool, capture_error_mode=self._mode)
        self._cur = g

    def _close(self):
        import warnings
        with warnings.catch_warnings():
            # the segment behind the LAST cut holds host work only (optimizer bookkeeping): "The CUDA Graph is empty" is expected there
            warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
            self._cur.capture_end()
        self.items.append(("graph", self._cur))
        self._cur = None

    def _abort(self):
        if self._cur is not None:
            try:
                self._cur.capture_end()
            except Exception:          # noqa: BLE001 -- the capture is already invalid; the original error is what matters
                pass
            self._cur = None

    def cut(self, fn):
        """Called by BackwardPipeline.run_slice on the capture's origin stream, every forked stream joined."""
        self._close()
        self.items.append(("call", fn))
        self._open()

    def __call__(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                x()
        return self.out
