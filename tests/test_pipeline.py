"""pointrcnn_amd/pipeline.py -- the throughput engine: S batches in flight, results in submission order.

CPU half: the ticket / slot bookkeeping with device="cpu" (eager steps, no streams).  GPU half: every slot's graph reproduces the
single-stream eager path bit for bit on the default.yaml RPN, order under load, H2D from pinned memory, re-entrancy (two pipelines,
two threads), exception safety of a failing submit."""
import threading

import numpy as np
import pytest
import torch


def _pipe(slots=3, **kw):
    from pointrcnn_amd.pipeline import InferencePipeline
    calls = []

    def step(inp, slot):
        calls.append(slot)
        return {"y": inp["x"] * 2 + 1, "slot": torch.tensor(slot)}
    return InferencePipeline(step, {"x": torch.zeros(4)}, slots=slots, device="cpu", **kw), calls


def test_results_come_back_in_submission_order_and_slots_rotate():
    pipe, calls = _pipe(3)
    tickets = [pipe.submit({"x": torch.full((4,), float(i))}) for i in range(3)]
    assert tickets == [0, 1, 2] and pipe.outstanding == 3
    for i in range(3):
        out = pipe.result()
        assert torch.equal(out["y"], torch.full((4,), 2.0 * i + 1)) and int(out["slot"]) == i
    assert pipe.outstanding == 0 and calls == [0, 1, 2]
    assert pipe.submit({"x": torch.ones(4)}) == 3 and int(pipe.result()["slot"]) == 0        # ticket 3 -> slot 0 again


def test_full_pipeline_refuses_and_map_interleaves():
    from pointrcnn_amd.pipeline import PipelineFull
    pipe, _ = _pipe(2)
    pipe.submit({"x": torch.zeros(4)})
    pipe.submit({"x": torch.zeros(4)})
    with pytest.raises(PipelineFull):
        pipe.submit({"x": torch.zeros(4)})
    assert pipe.outstanding == 2                       # the refused submit consumed no ticket
    pipe.drain()
    got = [float(o["y"][0]) for o in pipe.map({"x": torch.full((4,), float(i))} for i in range(7))]
    assert got == [2.0 * i + 1 for i in range(7)]
    with pytest.raises(RuntimeError):
        pipe.result()


def test_resident_replay_and_clone():
    pipe, _ = _pipe(2)
    pipe.submit({"x": torch.full((4,), 5.0)})
    a = pipe.result(clone=True)
    pipe.submit(None)                                  # slot 1: still the example data
    pipe.result()
    pipe.submit(None)                                  # slot 0 again: the resident 5.0
    b = pipe.result()
    assert torch.equal(a["y"], b["y"]) and a["y"].data_ptr() != b["y"].data_ptr()


def test_per_slot_examples_are_each_slots_resident_data():
    """example_inputs as a sequence: slot s is warmed up / captured on -- and keeps resident -- its own batch"""
    from pointrcnn_amd.pipeline import InferencePipeline
    step = lambda inp, slot: {"y": inp["x"] + 0.5}                            # noqa: E731
    ex = [{"x": torch.full((4,), float(10 * s))} for s in range(3)]
    pipe = InferencePipeline(step, ex, slots=3, device="cpu")
    got = [float(o["y"][0]) for o in pipe.map(None for _ in range(6))]        # two rounds on the resident inputs
    assert got == [0.5, 10.5, 20.5, 0.5, 10.5, 20.5]
    with pytest.raises(ValueError):
        InferencePipeline(step, ex[:2], slots=3, device="cpu")


def test_bad_batch_consumes_no_ticket_and_leaves_the_pipeline_usable():
    pipe, calls = _pipe(2)
    pipe.submit({"x": torch.ones(4)})
    for bad in ({"x": torch.ones(5)}, {"x": torch.ones(4, dtype=torch.float64)}, {"z": torch.ones(4)}, {"x": [1, 2, 3, 4]}):
        with pytest.raises((ValueError, KeyError)):
            pipe.submit(bad)
    assert pipe.outstanding == 1 and calls == [0]
    assert pipe.submit({"x": torch.full((4,), 2.0)}) == 1
    assert float(pipe.result()["y"][0]) == 3.0 and float(pipe.result()["y"][0]) == 5.0


def test_a_step_that_raises_is_reported_for_its_ticket_only():
    from pointrcnn_amd.pipeline import InferencePipeline

    def step(inp, slot):
        if float(inp["x"][0]) < 0:
            raise ArithmeticError("negative input")
        return inp["x"] + 1
    pipe = InferencePipeline(step, {"x": torch.zeros(2)}, slots=3, device="cpu")
    for v in (1.0, -1.0, 2.0):
        pipe.submit({"x": torch.full((2,), v)})
    assert float(pipe.result()[0]) == 2.0
    with pytest.raises(ArithmeticError):
        pipe.result()
    assert float(pipe.result()[0]) == 3.0 and pipe.outstanding == 0
    pipe.submit({"x": torch.full((2,), 7.0)})
    assert float(pipe.result()[0]) == 8.0
    pipe.close()
    with pytest.raises(RuntimeError):
        pipe.submit(None)


# ---- on the MI355X --------------------------------------------------------------------------------------------------
def _small_rpn(dev):
    from pointrcnn_amd import rpn
    torch.manual_seed(5)
    return rpn.randomize_bn_stats(rpn.RPN(), seed=4).to(dev).eval()


@pytest.mark.gpu
@pytest.mark.parametrize("copy_stream", [False, True])
def test_every_slot_reproduces_the_single_stream_path_bit_for_bit(dev, copy_stream):
    """default.yaml RPN + proposal layer, bs2 x 16384 points, 5 slots, 12 different batches from pinned host memory: each result ==
    the eager single-stream forward of that batch (what tools/eval_rcnn.py's loop computes), in submission order"""
    from pointrcnn_amd import rpn
    from pointrcnn_amd.pipeline import InferencePipeline
    from pointrcnn_amd.proposal_layer import ProposalLayer
    model, pl = _small_rpn(dev), ProposalLayer("TEST")

    def step(inp, slot):
        o = model(inp)
        o["rois"], o["roi_scores_raw"] = pl(o["rpn_cls"][:, :, 0], o["rpn_reg"], o["backbone_xyz"])
        return o
    hosts = [rpn.synthetic_clouds(2, 16384, seed0=3000 + 2 * i).pin_memory() for i in range(12)]
    keys = ("rpn_cls", "rpn_reg", "backbone_features", "rois", "roi_scores_raw")
    with torch.no_grad():
        want = []
        for h in hosts:
            o = step({"pts_input": h.to(dev)}, 0)
            want.append({k: o[k].clone() for k in keys})
    with InferencePipeline(step, {"pts_input": hosts[0]}, slots=5, copy_stream=copy_stream) as pipe:
        assert pipe.graphed and pipe.device.type == "cuda"
        n = 0
        for out in pipe.map({"pts_input": h} for h in hosts):
            for k in keys:
                assert torch.equal(out[k], want[n][k]), (n, k)
            n += 1
        assert n == 12
        # explicit form, device-resident inputs produced on another stream
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            d = hosts[7].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        pipe.submit({"pts_input": d}, ready=ev)
        out = pipe.result(clone=True)
        assert all(torch.equal(out[k], want[7][k]) for k in keys)


def _delay_kernel(ms=40.0):
    """keep the current stream busy for ~ms (torch's spin kernel counts device clock ticks whose rate differs between chips: calibrated)"""
    if not hasattr(_delay_kernel, "ticks_per_ms"):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        a.record()
        torch.cuda._sleep(1_000_000)
        b.record()
        b.synchronize()
        _delay_kernel.ticks_per_ms = 1_000_000 / max(a.elapsed_time(b), 1e-3)
    torch.cuda._sleep(int(ms * _delay_kernel.ticks_per_ms))


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("copy_stream", [False, True])
def test_consumer_work_enqueued_before_the_slot_is_reused_reads_its_own_ticket(dev, graph, copy_stream):
    """The round-4 wrong result, made deterministic: the consumer's `.clone()` of ticket t's outputs sits behind 40 ms of other work
    on the consumer's stream while `map()` already re-submits on the same slot.  The pipeline makes the slot (and the copy stream)
    wait for what the consumer has enqueued (pipeline.py `_wait_for_consumer`, `record_stream` for eager slots), so every clone
    holds its own ticket's values, and the pipeline reports that it had to wait (consumer_waits).  Negative control (graphed slots
    with per-slot copies, where the overwrite is deterministic): with that ordering switched off -- round 4's behaviour -- the same
    loop reads later tickets' values."""
    from pointrcnn_amd.pipeline import InferencePipeline
    step = lambda inp, slot: {"y": inp["x"] * 2 + 1}                          # noqa: E731
    hosts = [torch.full((1 << 22,), float(i)).pin_memory() for i in range(8)]

    def run(order):
        pipe = InferencePipeline(step, {"x": hosts[0]}, slots=2, graph=graph, copy_stream=copy_stream)
        pipe._order_consumers = order
        side, got = torch.cuda.Stream(), []
        with torch.cuda.stream(side):
            for o in pipe.map({"x": h} for h in hosts):
                _delay_kernel()
                got.append(o["y"].clone())
                del o
        side.synchronize()
        waits = pipe.consumer_waits
        pipe.close()
        return [(i, float(g[0])) for i, g in enumerate(got) if not bool((g == 2.0 * i + 1).all())], waits
    wrong, waits = run(True)
    assert wrong == [], wrong
    assert waits == len(hosts) - 2, waits          # every re-submission found its consumer busy and was ordered behind it
    if graph and not copy_stream:
        wrong, waits = run(False)
        assert waits == 0 and wrong != [], "the negative control no longer reproduces the race: this test proves nothing"


@pytest.mark.gpu
@pytest.mark.parametrize("copy_stream", [False, True])
def test_two_pipelines_and_two_threads(dev, copy_stream):
    """two pipelines (graphed and eager slots on shared streams, per-slot copies or the shared copy stream), two threads, each
    consumer cloning its results on its thread's default stream right before `map` reuses the slot; 50 rounds (round 4: one wrong
    tensor per run at pa's ticket 2, the one slot whose stream the eager pipeline does not share, i.e. whose next replay starts at
    once)"""
    from pointrcnn_amd import rpn
    from pointrcnn_amd.pipeline import InferencePipeline
    model = _small_rpn(dev)
    step = lambda inp, slot: model(inp)                                       # noqa: E731
    hosts = [rpn.synthetic_clouds(1, 16384, seed0=4000 + i).pin_memory() for i in range(8)]
    with torch.no_grad():
        want = [model({"pts_input": h.to(dev)})["rpn_reg"].clone() for h in hosts]
    pa = InferencePipeline(step, {"pts_input": hosts[0]}, slots=3, copy_stream=copy_stream)
    pb = InferencePipeline(step, {"pts_input": hosts[0]}, slots=2, graph=False, copy_stream=copy_stream)    # eager slots next to graphed ones
    errs = []

    def worker(pipe, order):
        try:
            torch.cuda.set_device(dev)
            for rep in range(50):
                got = [o["rpn_reg"].clone() for o in pipe.map({"pts_input": hosts[i]} for i in order)]
                for i, g in zip(order, got):
                    assert torch.equal(g, want[i]), (rep, i)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(pa, [0, 1, 2, 3, 4, 5, 6, 7])), threading.Thread(target=worker, args=(pb, [7, 5, 3, 1, 6]))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    pa.close(); pb.close()
    assert not errs, errs


@pytest.mark.gpu
def test_a_producer_keeps_submitting_while_a_consumer_waits_for_a_result(dev):
    """result() must not hold the state lock while it blocks on the ticket's event: with the oldest ticket stuck behind
    a 300 ms kernel, a second thread's submit() returns long before that ticket completes"""
    import time
    from pointrcnn_amd.pipeline import InferencePipeline

    def step(inp, slot):
        if slot == 0:
            _delay_kernel(300.0)
        return inp["x"] + 1
    _delay_kernel(1.0)                                                        # calibrate outside the pipeline's streams
    torch.cuda.synchronize()
    pipe = InferencePipeline(step, {"x": torch.zeros(16, device=dev)}, slots=2, graph=False, warmup=1)
    torch.cuda.synchronize()
    pipe.submit(None)                                                         # ticket 0: 300 ms
    waited, submitted = {}, {}

    def consumer():
        torch.cuda.set_device(dev)
        t0 = time.perf_counter()
        pipe.result()
        waited["s"] = time.perf_counter() - t0
    th = threading.Thread(target=consumer)
    th.start()
    time.sleep(0.05)                                                          # the consumer is inside result() now
    t0 = time.perf_counter()
    pipe.submit(None)                                                         # ticket 1 on slot 1
    submitted["s"] = time.perf_counter() - t0
    th.join()
    pipe.drain(); pipe.close()
    assert waited["s"] > 0.15 and submitted["s"] < 0.1, (waited, submitted)


@pytest.mark.gpu
def test_failing_submit_on_the_gpu_leaves_the_other_tickets_intact(dev):
    from pointrcnn_amd import rpn
    from pointrcnn_amd.pipeline import InferencePipeline, PipelineFull
    model = _small_rpn(dev)
    hosts = [rpn.synthetic_clouds(1, 16384, seed0=5000 + i).pin_memory() for i in range(3)]
    with torch.no_grad():
        want = [model({"pts_input": h.to(dev)})["rpn_cls"].clone() for h in hosts]
    pipe = InferencePipeline(lambda inp, slot: model(inp), {"pts_input": hosts[0]}, slots=2)
    pipe.submit({"pts_input": hosts[0]})
    with pytest.raises(ValueError):
        pipe.submit({"pts_input": torch.zeros(1, 100, 3).pin_memory()})
    pipe.submit({"pts_input": hosts[1]})
    with pytest.raises(PipelineFull):
        pipe.submit({"pts_input": hosts[2]})
    assert torch.equal(pipe.result()["rpn_cls"], want[0])
    pipe.submit({"pts_input": hosts[2]})
    assert torch.equal(pipe.result()["rpn_cls"], want[1]) and torch.equal(pipe.result()["rpn_cls"], want[2])
    pipe.close()
