"""GPU tests added in round 4: the advisor's round-3 findings (prefetch ordering / identity) and the roipool3d shapes the
VERDICT asked for (M = 300, the --save_rpn_feature path, and M = 64, the training shape)."""
import numpy as np
import pytest
import torch

from util import enlarge, kitti_cloud, rand_boxes3d

pytestmark = pytest.mark.gpu


def _intensity_rpn(dev):
    from pointrcnn_amd import rpn
    cfg = type("Cfg", (rpn.RPNConfig,), {"USE_INTENSITY": True, "SA_NPOINTS": [1024, 256, 64, 16], "NUM_POINTS": 4096})
    torch.manual_seed(3)
    return rpn.randomize_bn_stats(rpn.RPN(cfg=cfg), seed=2).to(dev).eval()


def test_prefetch_samples_of_an_intensity_cloud_arriving_by_async_copy(dev):
    """(B, N, 4) input: the xyz slice is a copy kernel.  It must run on the side stream behind the event of the H2D copy that
    produced the cloud -- the forward that picks the prefetched sample sets up returns the bits of a plain forward"""
    from pointrcnn_amd import rpn
    model = _intensity_rpn(dev)
    net = model.backbone_net
    B, N = 4, 4096
    host = torch.cat([rpn.synthetic_clouds(B, N, seed0=900), torch.rand(B, N, 1)], 2).pin_memory()
    with torch.no_grad():
        want = model({"pts_input": host.to(dev)})
        torch.cuda.synchronize()
        loader = torch.cuda.Stream()
        busy = torch.empty((4096, 4096), device=dev)
        for rep in range(3):
            pc = torch.empty((B, N, 4), device=dev)
            with torch.cuda.stream(loader):
                for _ in range(4):                       # the copy sits behind a few ms of other work on the loader stream
                    busy = busy @ busy * 0
                pc.copy_(host, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(loader)
            net.prefetch_samples(pc, ready=ready)
            assert net._prefetched is not None
            torch.cuda.current_stream().wait_event(ready)
            got = model({"pts_input": pc})
            assert net._prefetched is None
            for k in ("rpn_cls", "rpn_reg", "backbone_features"):
                assert torch.equal(got[k], want[k]), (rep, k)


def test_prefetched_sample_sets_are_dropped_for_another_tensor_at_the_same_address(dev):
    from pointrcnn_amd import rpn
    model = _intensity_rpn(dev)
    net = model.backbone_net
    B, N = 2, 4096
    a = torch.cat([rpn.synthetic_clouds(B, N, seed0=910), torch.rand(B, N, 1)], 2).to(dev)
    b_host = torch.cat([rpn.synthetic_clouds(B, N, seed0=920), torch.rand(B, N, 1)], 2)
    with torch.no_grad():
        want_b = model({"pts_input": b_host.to(dev)})
        net.prefetch_samples(a)
        ptr = a.data_ptr()
        torch.cuda.synchronize()
        del a                                              # the announced batch is dropped; a new tensor may land at its address
        b = b_host.to(dev)
        same_address = b.data_ptr() == ptr
        got = model({"pts_input": b})
    for k in ("rpn_cls", "rpn_reg"):
        assert torch.equal(got[k], want_b[k]), (k, same_address)


@pytest.mark.parametrize("M,N,S", [(300, 16384, 512), (64, 16384, 512), (300, 2000, 512)])
def test_roipool3d_shapes_of_save_rpn_feature_and_training(dev, cpu, M, N, S):
    """M = 300 (`--save_rpn_feature`, README.md:142-145) and M = 64 (TRAIN: ROI_PER_IMAGE); bit-equal to the oracle,
    plain and canonical forms, one empty RoI, C = 130"""
    from pointrcnn_amd import ops
    B, C = 2, 130
    xyz = kitti_cloud(B, N, seed=50 + M)
    boxes = np.stack([rand_boxes3d(xyz[b], M, seed=60 + b) for b in range(B)])
    boxes[0, -1, 0] += 500.0
    feat = np.random.default_rng(70).normal(size=(B, N, C)).astype(np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    big = np.stack([enlarge(boxes[b], 1.0) for b in range(B)])
    pooled, empty = ops.roipool3d(T(xyz), T(big), T(feat), S)
    wp, we = cpu.roipool3d(xyz, big, feat, S)
    assert np.array_equal(empty.cpu().numpy(), we) and we[0, -1] == 1
    assert np.array_equal(pooled.cpu().numpy(), wp)


def test_graphs_captured_on_one_batch_replay_on_another_at_bs32(dev, monkeypatch):
    """Round-4 regression.  bench.py's engine with every slot captured on slot 0's batch (PRCNN_BENCH_SAME_EXAMPLE) and the other
    slots' inputs replaced afterwards died with a GPU memory fault in group_compact_kernel at bs32 from the second set-abstraction
    level on (rocgdb; tools/graph_fault_probe2.py): the counters its list offsets come from were cleared by hipMemsetAsync, which a
    capture turns into a memset node of the hipGraph, and were not zero when the kernel ran in a replay.  They are cleared by a
    kernel now (common.h prcnn_fill_words).  Here: that exact path, 3 slots, and slot 1's replay == the eager step on its batch."""
    import argparse
    import bench
    from pointrcnn_amd import rpn
    from pointrcnn_amd.proposal_layer import ProposalLayer
    monkeypatch.setenv("PRCNN_BENCH_SAME_EXAMPLE", "1")
    torch.manual_seed(1234)
    model = rpn.randomize_bn_stats(rpn.RPN(), seed=7).to(dev).eval()
    pl = ProposalLayer("TEST")
    args = argparse.Namespace(streams=3, batch=32, npoints=16384, warmup=2, graph="on", workload="rpn")
    ib = bench.InferenceBench(args, model, dev, 0, 1, "uniform", pl, None).prepare()
    try:
        assert ib.pipe.graphed
        other = rpn.synthetic_clouds(32, 16384, seed0=bench.shard_seed0(0, 1, 1, 32)).to(dev)
        with torch.no_grad():
            want = ib.step_from({"pts_input": other}, 0)
            want = {k: want[k].clone() for k in ("rpn_cls", "rpn_reg", "rois")}
        assert torch.equal(ib.pipe.inputs[1]["pts_input"], other)
        for _ in range(2):                                   # replayed twice: the counters of the first replay are the second's stale bytes
            t = ib.pipe.submit(None)
            while t % 3 != 1:
                ib.pipe.result()
                t = ib.pipe.submit(None)
            while ib.pipe.outstanding > 1:
                ib.pipe.result()
            got = ib.pipe.result()
            for k in want:
                assert torch.equal(got[k], want[k]), k
    finally:
        ib.release()
