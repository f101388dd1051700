"""Step driver: the build's counterpart of `train_gen.forward` (train_gen.py:29-136).

Same batch contract (a dict of per-dialog tensors stacked on a leading batch dimension),
same row semantics: flatten [B, rounds, 1, L] -> [B*rounds, L]; in train mode keep rows whose label row is
not all zero and draw `batch_size` of them with replacement (torch.multinomial on the host RNG, exactly like
the reference); row r of a dialog uses the image tensors of dialog r // rounds.  Differences that do not
change results: image tensors may be passed UNEXPANDED ([B,37,2048] instead of the 10x host expansion of
train_gen.py:311-321) and the dead inputs (image_target / image_label / mlm labels / sep indices / hist len)
are neither gathered nor copied to the device.
"""
import contextlib

import torch

_TEXT_KEYS = ("enc_input_ids", "enc_segments", "enc_att_mask", "dec_input_ids", "dec_att_mask")


def select_rows(batch, params, sample_indices=None, generator=None):
    """Pure index work (device agnostic, bit-exact): returns (rows dict, sample_indices)."""
    ids = batch["enc_input_ids"]
    rows_per_dialog = 1
    for s in ids.shape[1:-1]:
        rows_per_dialog *= s
    flat = {k: batch[k].reshape(-1, batch[k].shape[-1]) for k in _TEXT_KEYS}
    n_rows = flat["enc_input_ids"].shape[0]
    train = "train" in params["mode"]
    if train:
        labels = batch["dec_labels"].reshape(-1, batch["dec_labels"].shape[-1])
        if sample_indices is None:
            cand = (labels.sum(-1) != 0).float()                                   # train_gen.py:66
            sample_indices = torch.multinomial(cand, params["batch_size"], replacement=True, generator=generator)
    elif sample_indices is None:
        sample_indices = torch.arange(n_rows)
    with _single_threaded_host_copies():
        out = {k: v[sample_indices] for k, v in flat.items()}
        if train:
            out["dec_labels"] = labels[sample_indices]
        for k, tail in (("enc_image_feat", 2), ("enc_image_loc", 2), ("enc_image_mask", 1)):
            v = batch[k]
            if v.dim() == tail + 1:                       # unexpanded [B, ...]: row r -> dialog r // rows_per_dialog
                out[k] = v[torch.div(sample_indices, rows_per_dialog, rounding_mode="floor")]
            else:                                         # reference layout [B, rounds, 1, ...]
                out[k] = v.reshape((-1,) + tuple(v.shape[-tail:]))[sample_indices]
    return out, sample_indices


@contextlib.contextmanager
def _single_threaded_host_copies():
    """torch's CPU copy / gather kernels run on the OpenMP pool; on a many-core host (256 logical CPUs on the MI355X boxes) the
    workers keep spinning after each parallel region and starve the ROCm runtime's own threads: measured with
    tools/h2d_probe.py, a 4.8 MB `tensor.copy_` on the host per step slowed the replay of the captured train step from 14.9 to
    24-35 ms (single iterations stalled 65-80 ms) -- whatever the copy to the device looked like -- and OMP_NUM_THREADS=1 brought
    every variant back to 15.2-15.3 ms.  The input path's host work is a few MB of memcpy: do it on one thread."""
    n = torch.get_num_threads()
    if n > 1:
        torch.set_num_threads(1)
    try:
        yield
    finally:
        if n > 1:
            torch.set_num_threads(n)


class PinnedStager(object):
    """Host -> device staging of a step's rows through PINNED host slots into PERSISTENT device buffers -- the build's replacement
    of the reference's per-step host work (train_gen.py:311-321: 10x expansion of five image tensors; :102-116: eleven pageable
    `.to(device)` calls incl. dead inputs).

      h = stager.fill(rows)       # HOST half: gather the sampled rows into this slot's pinned buffers (single-threaded memcpy)
      dev_rows = stager.upload(h) # DEVICE half: H2D into the slot's persistent device buffers (static addresses)

    `depth` slots rotate so that the host half of batch i+1 runs while the device computes batch i (`prefetch`).
    mode "pinned_async" (default): the H2D is issued at `fill` time on a dedicated copy stream and joined into the caller's
    stream with an event at `upload` -- the copy of batch i+1 runs under the compute of batch i and the host never blocks on
    it;  "pinned": blocking `copy_` on the caller's stream;  "pageable": plain `.to(device)` from pageable slots (what the
    reference does, minus the expansion).  Next to the replay of the captured train step (14.86 ms alone) tools/h2d_probe.py
    measures 15.15-15.23 / 15.36-15.40 / 15.44 ms per step for the three."""

    def __init__(self, device, depth=2, mode="pinned_async"):
        if mode not in ("pinned", "pinned_async", "pageable"):
            raise ValueError(mode)
        self.device = torch.device(device)
        self.depth, self.mode = depth, mode
        self.pinned = mode != "pageable"
        self.host = [dict() for _ in range(depth)]
        self.dev = [dict() for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device) if mode == "pinned_async" else None
        self.copied = [None] * depth          # pinned_async: this slot's H2D has finished
        self.consumed = [None] * depth        # pinned_async: event behind the compute that read this slot's device buffers
        self.last_slot = None                 # slot handed out by the latest upload(); its consumer is enqueued AFTER that call
        self._consumer_stream = None          # ... on this stream (the one upload() made wait for the H2D)
        self._last_marked = True
        self.i = 0
        self.bytes_staged = 0

    def _buf(self, table, slot, k, v, **kw):
        b = table[slot].get(k)
        if b is None or b.shape != v.shape or b.dtype != v.dtype:
            b = table[slot][k] = torch.empty(v.shape, dtype=v.dtype, **kw)
        return b

    def _mark_consumed(self):
        """The step that reads the device buffers of the slot handed out by the latest `upload()` is enqueued between that
        call and the NEXT call into the stager (fill or upload), on the stream that is current then.  So every entry point
        first records "everything enqueued so far" as that slot's consumed-event; a later `fill()` that reuses the slot makes
        the copy stream wait on it.  (Round 2 recorded the event only inside the next upload() of a DIFFERENT slot: with the
        fill-before-upload order of `prefetch()` the H2D of batch k waited on nothing relevant and could overwrite the rows
        step k-2 -- forward and, in eager mode, backward -- was still reading; a host that runs ahead of the device, as the
        reference loop with its one sync per 10 iterations does, exposes it.)  The event is recorded on the stream upload() saw."""
        if self.mode == "pinned_async" and self.last_slot is not None and not self._last_marked:
            # on the stream `upload()` ordered the rows on -- NOT on whatever stream is current now: a caller that runs the step
            # inside `with torch.cuda.stream(s):` but pulls the prefetch generator outside that context would otherwise record
            # on the default stream and the copy stream could overwrite rows the step on `s` is still reading
            self.mark_consumed(self._consumer_stream)

    def mark_consumed(self, stream=None):
        """Explicit form: "the consumer of the latest upload()'s rows has been enqueued on `stream`" (default: the stream that
        upload() ordered them on).  A caller whose step runs on yet another stream than the one upload() saw calls this."""
        if self.mode == "pinned_async" and self.last_slot is not None:
            done = torch.cuda.Event()
            done.record(stream if stream is not None else (self._consumer_stream or torch.cuda.current_stream(self.device)))
            self.consumed[self.last_slot] = done
            self._last_marked = True

    def fill(self, rows):
        """Host half (+ in mode pinned_async: the H2D is issued on the copy stream right away)."""
        self._mark_consumed()
        slot = self.i % self.depth
        self.i += 1
        if self.copied[slot] is not None:
            self.copied[slot].synchronize()       # the pinned buffers are about to be rewritten
        keys = []
        with _single_threaded_host_copies():
            for k, v in rows.items():
                if v is not None:
                    self._buf(self.host, slot, k, v, pin_memory=self.pinned).copy_(v)
                    self.bytes_staged += v.numel() * v.element_size()
                    keys.append(k)
        ev = None
        if self.mode == "pinned_async":
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])
            with torch.cuda.stream(self.copy_stream):
                for k in keys:
                    h = self.host[slot][k]
                    self._buf(self.dev, slot, k, h, device=self.device).copy_(h, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.copied[slot] = ev
        return slot, keys, [k for k, v in rows.items() if v is None], ev

    def upload(self, filled):
        """Device half: returns the slot's device tensors, ordered on the current stream."""
        self._mark_consumed()
        slot, keys, nones, ev = filled
        out = {k: None for k in nones}
        if self.mode == "pinned_async":
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            self.last_slot, self._last_marked, self._consumer_stream = slot, False, cur
            out.update({k: self.dev[slot][k] for k in keys})
            return out
        for k in keys:
            h = self.host[slot][k]
            if self.pinned:
                d = self._buf(self.dev, slot, k, h, device=self.device)
                d.copy_(h)
            else:
                d = h.to(self.device)
            out[k] = d
        return out

    def put(self, rows):
        return self.fill(rows)

    def get(self, handle):
        return self.upload(handle)


def prefetch(loader, params, stager, generator=None):
    """Iterate `loader`, yielding device rows; the NEXT batch's row selection, gather and pinned fill (all host work) run while
    the device computes on the current one -- train_gen.py:299-321's per-step host work moved off the critical path."""
    pending = None
    for batch in loader:
        rows, _ = select_rows(batch, params, None, generator)
        h = stager.fill(rows)                 # host work for batch i+1 while the device still runs batch i
        if pending is not None:
            yield stager.upload(pending)
        pending = h
    if pending is not None:
        yield stager.upload(pending)


def forward(model, batch, params, sample_indices=None, generator=None, stager=None):
    """lm_loss, lm_scores = forward(model, batch, params)  -- drop-in for train_gen.forward.
    `stager` (a PinnedStager): rows go through pinned memory and the copy stream instead of pageable synchronous copies."""
    rows, _ = select_rows(batch, params, sample_indices, generator)
    return forward_rows(model, rows, params, stager)


def forward_rows(model, rows, params, stager=None):
    dev = params["device"]
    if all((not torch.is_tensor(v)) or v.is_cuda for v in rows.values()):
        pass                                          # already staged (prefetch)
    elif stager is not None:
        rows = stager.get(stager.put(rows))
    else:
        rows = {k: v.to(dev) for k, v in rows.items()}
    loss, scores = model(
        enc_image_features=rows["enc_image_feat"], enc_image_spatials=rows["enc_image_loc"],
        enc_image_mask=rows["enc_image_mask"], enc_input_ids=rows["enc_input_ids"], enc_segments=rows["enc_segments"],
        enc_attention_mask=rows["enc_att_mask"], dec_input_ids=rows["dec_input_ids"],
        dec_attention_mask=rows["dec_att_mask"], dec_labels=rows.get("dec_labels"))
    if "train" in params["mode"]:
        loss = loss.mean()
    return loss, scores
