"""CPU, world_size 2, gloo: the multi-GPU plumbing of bench.py -- the REAL functions (shard_seed0, reduce_elapsed,
whole_job_value, resolve_world, self_launch_command), not a restatement.  Inference shards frames across ranks with NO
data-path collective; the only distributed traffic is the barrier and the max-over-ranks of the elapsed time."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    from pointrcnn_amd import rpn
    args = bench.parse(["--gpus", str(world), "--batch", "3"])
    w, r, lr, relaunch = bench.resolve_world(args, os.environ, ndev=world)
    assert (w, r, lr, relaunch) == (world, rank, rank, False)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = [bench.shard_seed0(rank, world, slot, args.batch) for slot in range(2)]
    clouds = rpn.synthetic_clouds(args.batch, 256, seed0=seeds[0])
    gathered = [torch.empty_like(clouds) for _ in range(world)]
    dist.all_gather(gathered, clouds)                       # test-only: prove the shards are disjoint + deterministic
    dist.barrier()
    elapsed = bench.reduce_elapsed(0.010 * (rank + 1), dist, "cpu")      # rank 1 is the slow one
    dist.barrier()
    q.put((rank, elapsed, [g.sum().item() for g in gathered], clouds[0, :2].tolist(), seeds))
    dist.destroy_process_group()


def test_frame_sharding_and_max_over_ranks_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(abs(r[1] - 0.020) < 1e-12 for r in res)              # every rank sees the MAX (slowest rank)
    assert res[0][2] == res[1][2]                                     # same global view on both ranks
    assert res[0][2][0] != res[0][2][1]                               # rank shards are different frames
    sys.path.insert(0, ROOT)
    import bench
    from pointrcnn_amd import rpn
    whole = rpn.synthetic_clouds(6, 256, seed0=100)                   # frames 0..5 == rank0's 0..2 + rank1's 3..5
    assert whole[0, :2].tolist() == res[0][3] and whole[3, :2].tolist() == res[1][3]
    # slots of all ranks tile the frame sequence without overlap: seeds 100, 103 | 106, 109
    assert res[0][4] == [100, 106] and res[1][4] == [103, 109]
    assert bench.whole_job_value(3, world, 5, res[0][1]) == 3 * world * 5 / 0.020           # value = all ranks' frames / max time


def test_gpus_flag_cannot_silently_run_on_fewer_devices():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse(["--gpus", "8"])
    with pytest.raises(SystemExit):                                   # 8 requested, 1 visible, no launcher: refuse
        bench.resolve_world(args, {}, ndev=1)
    with pytest.raises(SystemExit):                                   # launcher world size disagrees with --gpus
        bench.resolve_world(args, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, ndev=8)
    assert bench.resolve_world(args, {}, ndev=8) == (8, 0, 0, True)   # 8 visible, no launcher: self-launch
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "5"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "5"]
    assert bench.resolve_world(bench.parse([]), {}, ndev=1) == (1, 0, 0, False)
    # defaults per workload: the BASELINE metric's batch 32, config 4's 16 per GPU for training
    assert bench.parse([]).batch == 32 and bench.parse(["--workload", "train"]).batch == 16
