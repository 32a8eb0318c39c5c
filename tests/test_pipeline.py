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


@pytest.mark.gpu
def test_two_pipelines_and_two_threads(dev):
    from pointrcnn_amd import rpn
    from pointrcnn_amd.pipeline import InferencePipeline
    model = _small_rpn(dev)
    step = lambda inp, slot: model(inp)                                       # noqa: E731
    hosts = [rpn.synthetic_clouds(1, 16384, seed0=4000 + i).pin_memory() for i in range(8)]
    with torch.no_grad():
        want = [model({"pts_input": h.to(dev)})["rpn_reg"].clone() for h in hosts]
    pa = InferencePipeline(step, {"pts_input": hosts[0]}, slots=3)
    pb = InferencePipeline(step, {"pts_input": hosts[0]}, slots=2, graph=False)       # eager slots next to graphed ones
    errs = []

    def worker(pipe, order):
        try:
            torch.cuda.set_device(dev)
            got = [o["rpn_reg"].clone() for o in pipe.map({"pts_input": hosts[i]} for i in order)]
            for i, g in zip(order, got):
                assert torch.equal(g, want[i]), i
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
