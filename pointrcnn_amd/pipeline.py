"""Throughput engine: S batches in flight, each a captured hipGraph on its own HIP stream.

What tools/eval_rcnn.py:459-520 does per batch -- `inputs = torch.from_numpy(pts_input).cuda(non_blocking=True)`, `model(input_data)`,
read the outputs -- keeps ONE batch in flight on one stream.  On this path that exposes the furthest-point-sampling chain: a serial
3.8 ms chain on one workgroup per frame (32 of 256 CUs for a bs32 batch), during which the rest of the chip idles (5.2 k
frames/s).  Batches are independent, so the engine keeps `slots` of them in flight: slot s owns a HIP stream (and, with
GPU_MAX_HW_QUEUES >= slots, a hardware queue), static device input buffers, and a hipGraph of one whole step captured on that
stream.  `submit()` copies a batch's inputs into the slot's buffers (pinned host memory -> asynchronous H2D on the slot's stream, or, with
copy_stream=True, on one copy stream shared by the slots, first come first served; either way overlapping the other slots' kernels) and replays the graph; results come back in submission order.  One batch's FPS then runs
underneath other batches' MLP kernels (16.2-17.8 k frames/s on the same graph).

    pipe = InferencePipeline(lambda inp, slot: model(inp), {"pts_input": example}, slots=20)
    for out in pipe.map(batches):          # batches: iterable of {"pts_input": pinned host or device tensor}
        consume(out)                       # dict of the step's output tensors; valid until `slots` more batches were submitted

    t = pipe.submit(batch)                 # explicit form: ticket numbers count up from 0
    out = pipe.result()                    # the oldest outstanding ticket, blocking on its event only

Contract
  * results are returned in submission order; `result()` waits for that ticket's event, not for the device;
  * at most `slots` tickets are outstanding: `submit()` on a full pipeline raises PipelineFull (collect a result first; `map`
    does the interleaving);
  * the dict returned for ticket t aliases slot (t % slots)'s static output tensors: it is valid until ticket t + slots is
    submitted (`result(clone=True)` copies it out on the slot's stream).  "Valid" is a statement about STREAM ORDER, and the
    pipeline enforces its half of it: `result()` notes the HIP stream the collecting thread is on, and the submit that reuses the
    slot makes the slot's stream (and the copy stream) wait for everything that stream was given up to then -- so device work the
    consumer ENQUEUED on the outputs before ticket t + slots was submitted (a `.clone()`, a kernel, a D2H copy) reads ticket t's
    data even if it has not run yet.  Work the consumer enqueues later, or on a stream other than its current one at `result()`
    time, is the consumer's to order (round 4 did not enforce this: `map()` re-submitted on the slot right after the yield, and a
    consumer-side `.clone()` still queued on the thread's default stream read the NEXT ticket's values -- the wrong result of
    `test_two_pipelines_and_two_threads`; regression: tests/test_pipeline.py::test_consumer_work_enqueued_before_the_slot_is_reused_reads_its_own_ticket).
    Eager slots (graph=False) rebind their outputs every step: `result()` calls `record_stream` on them for the consumer's stream,
    so the allocator does not hand their blocks to the slot's next step while the consumer's reads are pending;
  * a failing submit (bad input shape / dtype, a step that raises in eager mode) leaves the pipeline usable: the slot is free
    again, no ticket was consumed, the tickets in flight are untouched;
  * `copy_stream=True` (what bench.py uses; ONE producer thread): a batch's inputs travel on one copy stream shared by the slots, in
    submission order, and the slot's stream waits for its copy's event.  20 slots each copying on their own stream share the DMA
    engines chunk by chunk, so after a cold start all 20 copies finish together, late, and every batch's sampling chain starts late
    with them (20-step runs with host inputs: 12.1-12.8 k instead of 15.7 k frames/s); first come, first served lets slot 0 start
    after one copy.  The default (False) keeps every copy on its slot's stream.  Both modes are covered by the two-threads test
    (the round-4 failure seen with the copy stream was the consumer race described above, which the copy stream's timing exposed);
  * thread-safe: one state lock around the bookkeeping of submit / result, collectors serialised among themselves; `result()` does
    NOT hold the state lock while it waits for the ticket's event, so a producer thread keeps submitting while a consumer thread
    blocks; several pipelines may coexist (they share one process-wide set of streams:
    torch's pool holds 32 stream handles and a second set of 20 wraps around it -- measured 30 % slower than the first).

`device="cpu"` runs the same ticket / slot bookkeeping with eager steps and no streams: the host logic is testable without a GPU
(the step function is the caller's; nothing here computes).
"""
import os
import threading
import warnings

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue serialise.  Only
# effective when set before the HIP runtime starts, i.e. before the first CUDA call of the process.
_QUEUES_PRESET = "GPU_MAX_HW_QUEUES" in os.environ      # the user's own setting: taken as effective
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import torch  # noqa: E402

# a setdefault made after the HIP runtime started (a model moved to the GPU before this import: the usual order for a library
# user) changes nothing: the process then runs on the default 4 queues whatever the environment says now
_QUEUES_TOO_LATE = (not _QUEUES_PRESET) and torch.cuda.is_initialized()

_STREAMS = {}                 # device index -> list of streams shared by every pipeline of the process
_COPY_STREAMS = {}            # device index -> the input-copy stream shared by every pipeline of the process
_STREAMS_LOCK = threading.Lock()


class PipelineFull(RuntimeError):
    """submit() with `slots` tickets outstanding"""


def shared_streams(device, n):
    """the first n of the process-wide streams of `device` (created on demand, never released)"""
    with _STREAMS_LOCK:
        pool = _STREAMS.setdefault(device.index, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=device))
        return pool[:n]


def shared_copy_stream(device):
    with _STREAMS_LOCK:
        cs = _COPY_STREAMS.get(device.index)
        if cs is None:
            cs = _COPY_STREAMS[device.index] = torch.cuda.Stream(device=device)
        return cs


def _flatten(out):
    """tensors of a step's result (dict / tuple / list / tensor), for cloning and checks"""
    if isinstance(out, torch.Tensor):
        return [out]
    if isinstance(out, dict):
        return [t for v in out.values() for t in _flatten(v)]
    if isinstance(out, (tuple, list)):
        return [t for v in out for t in _flatten(v)]
    return []


def _clone(out):
    if isinstance(out, torch.Tensor):
        return out.clone()
    if isinstance(out, dict):
        return {k: _clone(v) for k, v in out.items()}
    if isinstance(out, (tuple, list)):
        return type(out)(_clone(v) for v in out)
    return out


class InferencePipeline:
    def __init__(self, step_fn, example_inputs, slots=20, device=None, graph=True, warmup=2, copy_stream=False):
        """step_fn(inputs: dict of the slot's static tensors, slot: int) -> tensor / dict / tuple of tensors.  It is run `warmup`
        times eagerly per slot (lazy initialisation, weight packing, allocator pool) and then captured; everything it launches
        must go to torch's current stream (every C-ABI call of this package does).
        example_inputs: dict name -> tensor giving shape and dtype of every input (contents are the slot's initial data, so a
        pipeline can also be replayed on resident inputs: submit(None)); or a sequence of `slots` such dicts, one per slot (every
        slot is then warmed up and captured on its own data)."""
        per_slot = None
        if isinstance(example_inputs, (list, tuple)):
            per_slot = list(example_inputs)
            if len(per_slot) != int(slots):
                raise ValueError("InferencePipeline: %d example dicts for %d slots" % (len(per_slot), int(slots)))
            example_inputs = per_slot[0]
        some = next(iter(example_inputs.values()))
        if device is not None:
            self.device = torch.device(device)
        elif some.is_cuda:
            self.device = some.device
        elif torch.cuda.is_available():
            self.device = torch.device("cuda", torch.cuda.current_device())
        else:
            raise RuntimeError("InferencePipeline needs a GPU (device='cpu' runs the slot bookkeeping only, for tests)")
        self.on_gpu = self.device.type == "cuda"
        self.slots = int(slots)
        assert self.slots >= 1
        self.step_fn = step_fn
        self.graphed = bool(graph) and self.on_gpu
        queues = 4 if _QUEUES_TOO_LATE else int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        if self.on_gpu and self.slots > queues:
            warnings.warn("InferencePipeline: %d slots but %s: streams that share a hardware queue serialise "
                          "(export GPU_MAX_HW_QUEUES before the first HIP call of the process)" %
                          (self.slots, "HIP was initialised before pointrcnn_amd.pipeline was imported, so the default 4 hardware "
                           "queues are in use" if _QUEUES_TOO_LATE else "GPU_MAX_HW_QUEUES=%d" % queues))
        self.streams = shared_streams(self.device, self.slots) if self.on_gpu else [None] * self.slots
        self.inputs = [{k: v.detach().to(self.device, copy=True) for k, v in (per_slot[s_] if per_slot else example_inputs).items()}
                       for s_ in range(self.slots)]
        self.outputs = [None] * self.slots
        self.graphs = [None] * self.slots
        self.events = [torch.cuda.Event() if self.on_gpu else None for _ in range(self.slots)]
        if os.environ.get("PRCNN_PIPELINE_COPY_STREAM") in ("0", "1"):      # dev A/B switch
            copy_stream = os.environ["PRCNN_PIPELINE_COPY_STREAM"] == "1"
        self.copy_stream = shared_copy_stream(self.device) if (self.on_gpu and copy_stream) else None
        self.copy_events = [torch.cuda.Event() if self.copy_stream is not None else None for _ in range(self.slots)]
        self._lock = threading.RLock()           # ticket / slot bookkeeping
        self._collect_lock = threading.RLock()   # one collector at a time (results are handed out in order)
        self._consumer = [None] * self.slots     # slot -> HIP stream its last result was handed to (see `result`)
        self._order_consumers = True             # False = round 4's behaviour (the negative control of the regression test)
        self.consumer_waits = 0                  # submits that found the slot's consumer stream still busy and waited for it
        self._consumer_events = [torch.cuda.Event() if self.on_gpu else None for _ in range(self.slots)]
        self._next_ticket = 0          # ticket the next submit() gets
        self._next_result = 0          # oldest outstanding ticket
        self._failed = {}              # ticket -> exception raised by its (eager) step
        self._closed = False
        try:
            for s in range(self.slots):
                self._prepare_slot(s, warmup)
        except Exception:
            self.close()
            raise

    # ---- construction ---------------------------------------------------------------------------------------------
    def _prepare_slot(self, s, warmup):
        if not self.on_gpu:
            return
        stream = self.streams[s]
        stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(max(1, warmup)):
                out = self.step_fn(self.inputs[s], s)
        stream.synchronize()
        if not self.graphed:
            self.outputs[s] = out
            return
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g, stream=stream):
            out = self.step_fn(self.inputs[s], s)
        self.graphs[s], self.outputs[s] = g, out

    # ---- submission -----------------------------------------------------------------------------------------------
    @property
    def outstanding(self):
        return self._next_ticket - self._next_result

    def _wait_for_consumer(self, s):
        """slot s is about to be overwritten: whatever the consumer of its last result has enqueued so far on the stream it
        collected on runs first (module docstring, third contract item).  Costs one stream query when that stream is idle."""
        cs, self._consumer[s] = self._consumer[s], None
        if not self._order_consumers:
            return
        if cs is None or cs == self.streams[s] or cs.query():
            return
        self.consumer_waits += 1
        ev = self._consumer_events[s]
        ev.record(cs)
        self.streams[s].wait_event(ev)
        if self.copy_stream is not None:
            self.copy_stream.wait_event(ev)

    def _validate(self, s, batch):
        if batch is None:
            return
        unknown = set(batch) - set(self.inputs[s])
        if unknown:
            raise KeyError("InferencePipeline.submit: unknown input(s) %s (expected %s)" % (sorted(unknown), sorted(self.inputs[s])))
        for k, v in batch.items():
            dst = self.inputs[s][k]
            if not isinstance(v, torch.Tensor) or tuple(v.shape) != tuple(dst.shape) or v.dtype != dst.dtype:
                raise ValueError("InferencePipeline.submit: input %r must be a %s tensor of shape %s, got %s" %
                                 (k, dst.dtype, tuple(dst.shape), "%s %s" % (v.dtype, tuple(v.shape)) if isinstance(v, torch.Tensor) else type(v)))

    def _load(self, s, batch, ready):
        """copy a (validated) batch into slot s's static inputs: on the copy stream, or on the slot's stream"""
        if batch is None:
            return
        if self.copy_stream is not None:
            cs = self.copy_stream
            cs.wait_event(self.events[s])          # the slot's previous replay has read its inputs (no-op before the first one)
            if ready is not None:
                cs.wait_event(ready)
            with torch.cuda.stream(cs):
                for k, v in batch.items():
                    self.inputs[s][k].copy_(v, non_blocking=True)
            self.copy_events[s].record(cs)
            self.streams[s].wait_event(self.copy_events[s])
            return
        if self.on_gpu and ready is not None:
            self.streams[s].wait_event(ready)
        for k, v in batch.items():
            self.inputs[s][k].copy_(v, non_blocking=True)

    def submit(self, batch=None, ready=None):
        """enqueue one batch; -> ticket.  batch: dict name -> tensor (pinned host memory: asynchronous H2D on the shared copy stream,
        the slot's stream waits for it; device tensors: D2D, `ready` = event after their producer if that ran on another stream),
        or None = replay on the data already resident in the slot's input buffers."""
        with self._lock:
            if self._closed:
                raise RuntimeError("InferencePipeline is closed")
            if self.outstanding >= self.slots:
                raise PipelineFull("%d tickets outstanding on %d slots: collect result() first" % (self.outstanding, self.slots))
            t = self._next_ticket
            s = t % self.slots
            if not self.on_gpu:
                self._validate(s, batch)
                self._load(s, batch, None)
                try:
                    with torch.no_grad():
                        self.outputs[s] = self.step_fn(self.inputs[s], s)
                except Exception as e:  # noqa: BLE001  (reported by result() for this ticket, in order)
                    self._failed[t] = e
                self._next_ticket = t + 1
                return t
            stream = self.streams[s]
            self._validate(s, batch)                          # raises before anything of this ticket is enqueued
            self._wait_for_consumer(s)
            with torch.cuda.stream(stream):
                self._load(s, batch, ready)
                if self.graphed:
                    self.graphs[s].replay()
                else:
                    try:
                        with torch.no_grad():
                            self.outputs[s] = self.step_fn(self.inputs[s], s)
                    except Exception as e:  # noqa: BLE001
                        self._failed[t] = e
                self.events[s].record(stream)
            self._next_ticket = t + 1
            return t

    # ---- collection -----------------------------------------------------------------------------------------------
    def result(self, clone=False):
        """outputs of the oldest outstanding ticket (blocks on that ticket's event only, without holding the state lock: other
        threads keep submitting meanwhile); raises the step's exception if its eager step failed -- the ticket is consumed either
        way.  The calling thread's current HIP stream is noted as the consumer of the slot (module docstring)."""
        with self._collect_lock:
            with self._lock:
                if self.outstanding <= 0:
                    raise RuntimeError("InferencePipeline.result: nothing outstanding")
                t = self._next_result
                s = t % self.slots
                ev, cloned = self.events[s], None
            if self.on_gpu:
                ev.synchronize()                   # the ticket stays outstanding while we wait: its slot cannot be resubmitted
            with self._lock:
                err = self._failed.pop(t, None)
                out = self.outputs[s]
                if err is None and self.on_gpu:
                    if clone:                      # enqueued on the slot's stream: ahead of any later replay of the slot
                        with torch.cuda.stream(self.streams[s]):
                            out = _clone(out)
                        cloned = torch.cuda.Event()
                        cloned.record(self.streams[s])
                    else:
                        cs = torch.cuda.current_stream(self.device)
                        self._consumer[s] = cs
                        if not self.graphed and cs != self.streams[s] and self._order_consumers:
                            for x in _flatten(out):
                                if x.is_cuda:
                                    x.record_stream(cs)
                elif err is None and clone:
                    out = _clone(out)
                self._next_result = t + 1
            if err is not None:
                raise err
            if cloned is not None:
                cloned.synchronize()
            return out

    def map(self, batches, clone=False):
        """run an iterable of batches through the pipeline, `slots` in flight, yielding results in order"""
        for b in batches:
            if self.outstanding >= self.slots:
                yield self.result(clone)
            self.submit(b)
        while self.outstanding > 0:
            yield self.result(clone)

    def drain(self):
        """wait for everything in flight and drop the results"""
        with self._collect_lock:
            while self.outstanding > 0:
                try:
                    self.result()
                except Exception:  # noqa: BLE001
                    pass

    def close(self):
        with self._lock:
            self._closed = True
            if self.on_gpu:
                if self.copy_stream is not None:
                    self.copy_stream.synchronize()
                for s in self.streams:
                    s.synchronize()
            self.graphs = [None] * self.slots
            self.outputs = [None] * self.slots
            self.inputs = [{} for _ in range(self.slots)]
            self._next_result = self._next_ticket
            self._failed.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
