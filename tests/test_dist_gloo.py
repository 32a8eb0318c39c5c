"""CPU, world_size 2, gloo: the multi-GPU path of bench.py shards frames across ranks with NO data-path
collective; the only distributed traffic is the barrier and the max-over-ranks of the elapsed time.  This test
runs exactly that logic (frame sharding by rank + barrier-bracketed timing + MAX reduce) over gloo."""
import os
import socket
import sys

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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointrcnn_amd import rpn
    batch = 3
    # bench.py: clouds_cpu = synthetic_clouds(batch, npoints, seed0=100 + rank * batch)
    clouds = rpn.synthetic_clouds(batch, 256, seed0=100 + rank * batch)
    gathered = [torch.empty_like(clouds) for _ in range(world)]
    dist.all_gather(gathered, clouds)                       # test-only: prove the shards are disjoint + deterministic
    dist.barrier()
    elapsed = torch.tensor([0.010 * (rank + 1)], dtype=torch.float64)     # rank 1 is the slow one
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, float(elapsed.item()), [g.sum().item() for g in gathered], clouds[0, :2].tolist()))
    dist.destroy_process_group()


def test_frame_sharding_and_max_over_ranks_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(abs(r[1] - 0.020) < 1e-12 for r in res)              # every rank sees the MAX (slowest rank)
    assert res[0][2] == res[1][2]                                     # same global view on both ranks
    assert res[0][2][0] != res[0][2][1]                               # rank shards are different frames
    sys.path.insert(0, ROOT)
    from pointrcnn_amd import rpn
    whole = rpn.synthetic_clouds(6, 256, seed0=100)                   # frames 0..5 == rank0's 0..2 + rank1's 3..5
    assert whole[0, :2].tolist() == res[0][3] and whole[3, :2].tolist() == res[1][3]
    frames, steps = 3 * world * 5, 5
    assert frames / res[0][1] == 3 * world * steps / 0.020           # value = all ranks' frames / max time
