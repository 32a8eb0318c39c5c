"""BASELINE config 4 on CPU: the RPN training step under DistributedDataParallel over gloo, world size 2.

The step is pointrcnn_amd.train_functions.RPNTrainer (the code bench.py --workload train runs on GPUs over RCCL); the point
operators underneath are routed to the CPU oracle (tests/cpu_ops.py) because this container has no GPU.  Checked:
  * the all-reduced gradient of 2 ranks x 1 frame equals the single-process gradient of the 2-frame batch within 1e-5
    (global loss normalisation makes DDP's average the reference's DataParallel global-batch loss; BatchNorm in eval mode so
    that batch statistics do not depend on the split),
  * every rank holds identical parameters after optimizer.step(),
  * with BatchNorm in training mode the ranks still agree with each other (DDP invariants) and rank 0's buffers are broadcast.
"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_cfg():
    from pointrcnn_amd import rpn
    return type("SmallRPN", (rpn.RPNConfig,), {
        "NUM_POINTS": 1024, "SA_NPOINTS": [256, 64, 16, 8],
        "SA_RADIUS": [[0.2, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]], "SA_NSAMPLE": [[8, 16], [8, 16], [8, 16], [8, 16]],
        "SA_MLPS": [[[8, 8, 16], [8, 8, 16]], [[16, 16, 32], [16, 16, 32]], [[32, 32, 64], [32, 32, 64]], [[64, 64, 128], [64, 64, 128]]],
        "FP_MLPS": [[32, 32], [64, 64], [64, 64], [64, 64]], "CLS_FC": [32], "REG_FC": [32]})


def _batch(frames, seed=0):
    """(pts, cls_label, reg_label) for `frames` frames: 1024 points in a 6 m cube, 3 GT boxes per frame"""
    sys.path.insert(0, TESTS)
    import oracle
    from util import enlarge, rand_boxes3d
    cpu = oracle.cpu()
    pts = np.stack([np.random.default_rng(seed + f).uniform([-3, -1, 10], [3, 2, 16], (1024, 3)).astype(np.float32) for f in range(frames)])
    gt = np.stack([rand_boxes3d(pts[f], 3, seed=seed + 10 + f, jitter=0.05) for f in range(frames)])
    gt[..., 3:6] *= 1.5
    cls, reg = cpu.rpn_labels(pts, gt)
    return torch.from_numpy(pts), torch.from_numpy(cls).long(), torch.from_numpy(reg)


def _model(bn_eval):
    sys.path.insert(0, TESTS)
    import cpu_ops
    from pointrcnn_amd import rpn, train_functions as tf
    m = tf.init_rpn_head_weights(rpn.RPN(cfg=_small_cfg()))
    cpu_ops.fill_params_by_name(m, seed=3)
    return m


def _freeze_norm_and_dropout(trainer, bn_eval):
    """RPNTrainer.step puts the model in train(); for the parity check BatchNorm uses running statistics and dropout is off"""
    orig = trainer.model.train

    def train_then_freeze(mode=True):
        orig(mode)
        for mod in trainer.model.modules():
            if isinstance(mod, torch.nn.Dropout) or (bn_eval and isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d))):
                mod.eval()
        return trainer.model
    trainer.model.train = train_then_freeze


def _drop_foreground(cls, reg, frame):
    """frame `frame` of the batch gets no foreground point at all (an empty scene: every label background)"""
    cls, reg = cls.clone(), reg.clone()
    cls[frame] = 0
    reg[frame] = 0
    return cls, reg


def _worker(rank, world, port, q, bn_eval, empty_frame=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_ops
    from pointrcnn_amd import train_functions as tf
    pts, cls, reg = _batch(world, seed=0)
    if empty_frame is not None:
        cls, reg = _drop_foreground(cls, reg, empty_frame)
    shard = {"pts_input": pts[rank:rank + 1], "rpn_cls_label": cls[rank:rank + 1], "rpn_reg_label": reg[rank:rank + 1]}
    model = _model(bn_eval)
    if rank == 1:                                   # DDP must broadcast rank 0's parameters and buffers
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    trainer = tf.RPNTrainer(model, ddp=True, optimizer="sgd")
    _freeze_norm_and_dropout(trainer, bn_eval)
    with cpu_ops.oracle_ops():
        tb = {}
        loss = trainer.step(shard, tb)
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    q.put((rank, float(loss.item()), {n: g.numpy() for n, g in grads.items()}, params.numpy(), tb["rpn_fg_sum"]))
    dist.barrier()
    dist.destroy_process_group()


def _run(bn_eval, empty_frame=None, target=None, extra=()):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target or _worker, args=(r, world, port, q, bn_eval, empty_frame) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _check_against_single_process(res, empty_frame=None):
    sys.path.insert(0, TESTS)
    import cpu_ops
    from pointrcnn_amd import train_functions as tf
    # single process, the whole 2-frame batch, the reference's (global) loss normalisation
    pts, cls, reg = _batch(2, seed=0)
    if empty_frame is not None:
        cls, reg = _drop_foreground(cls, reg, empty_frame)
    model = _model(True)
    trainer = tf.RPNTrainer(model, ddp=False, optimizer="sgd")
    _freeze_norm_and_dropout(trainer, True)
    with cpu_ops.oracle_ops():
        tb = {}
        loss = trainer.step({"pts_input": pts, "rpn_cls_label": cls, "rpn_reg_label": reg}, tb)
    assert res[0][4] + res[1][4] == tb["rpn_fg_sum"]
    assert (min(res[0][4], res[1][4]) > 0) if empty_frame is None else (res[empty_frame][4] == 0 and tb["rpn_fg_sum"] > 0)
    # the global loss is the AVERAGE of the rescaled per-rank losses
    assert abs(0.5 * (res[0][1] + res[1][1]) - float(loss.item())) <= 1e-5 * max(1.0, abs(float(loss.item())))
    single = {n: p.grad.numpy() for n, p in model.named_parameters() if p.grad is not None}
    assert set(single) == set(res[0][2]) == set(res[1][2])
    worst = 0.0
    for n, g in single.items():
        scale = max(1.0, float(np.abs(g).max()))
        assert np.array_equal(res[0][2][n], res[1][2][n]), "ranks disagree on the all-reduced gradient of %s" % n
        worst = max(worst, float(np.abs(res[0][2][n] - g).max()) / scale)
    assert worst <= 1e-5, worst
    assert np.array_equal(res[0][3], res[1][3])      # identical parameters on both ranks after the step (rank 1 started perturbed)
    after = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy()
    assert np.abs(after - res[0][3]).max() <= 1e-6   # and equal to the single-process step


def test_ddp_gradient_equals_single_process_global_batch():
    _check_against_single_process(_run(bn_eval=True))


def test_ddp_rank_without_foreground_neither_hangs_nor_changes_the_gradient():
    """One rank's shard is an empty scene (fg_sum == 0 branch, train_functions.py:117-118 of the reference).  Its regression
    head must still take part in the bucketed all-reduce -- with the reg outputs out of the autograd graph DDP would wait for
    that bucket forever on the peers and raise on this rank's next forward -- and the averaged gradient is still the
    single-process gradient of the 2-frame batch (the reference's DataParallel handles this case by construction)."""
    _check_against_single_process(_run(bn_eval=True, empty_frame=1), empty_frame=1)


def _loss_worker(rank, world, port, q, bn_eval, empty_frame, loss_cls):
    """loss-level check for the LOSS_CLS settings: grads w.r.t. this rank's own logits under `dist`"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointrcnn_amd import train_functions as tf
    cls_o, reg_o, cls_l, reg_l = _loss_case(world)
    cfg = type("C", (tf.RPNLossConfig,), {"LOSS_CLS": loss_cls})
    a = cls_o[rank:rank + 1].clone().requires_grad_(True)
    b = reg_o[rank:rank + 1].clone().requires_grad_(True)
    loss = tf.get_rpn_loss(a, b, cls_l[rank:rank + 1], reg_l[rank:rank + 1], cfg, dist=dist)
    loss.backward()
    q.put((rank, float(loss.item()), a.grad.numpy(), b.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _loss_case(frames, n=512, seed=5):
    g = torch.Generator().manual_seed(seed)
    cls_o = torch.randn(frames, n, 1, generator=g)
    reg_o = torch.randn(frames, n, 76, generator=g) * 0.3
    cls_l = (torch.rand(frames, n, generator=g) < 0.2).long() - (torch.rand(frames, n, generator=g) < 0.05).long()
    cls_l[1, : n // 2] = cls_l[1, : n // 2].clamp(max=0)             # ragged foreground counts across the ranks
    reg_l = torch.randn(frames, n, 7, generator=g)
    reg_l[..., 3:6] = reg_l[..., 3:6].abs() + 1.0
    return cls_o, reg_o, cls_l, reg_l


def test_every_cls_loss_is_the_global_batch_loss_under_data_parallel():
    """focal, dice and BCE: world x the DDP-averaged gradient (= the mean of the ranks' gradients w.r.t. their own outputs, here
    compared output by output) is the gradient of the ONE loss the reference computes over the gathered batch"""
    from pointrcnn_amd import train_functions as tf
    for loss_cls in ("SigmoidFocalLoss", "DiceLoss", "BinaryCrossEntropy"):
        res = _run(True, None, target=_loss_worker, extra=(loss_cls,))
        cls_o, reg_o, cls_l, reg_l = _loss_case(2)
        cfg = type("C", (tf.RPNLossConfig,), {"LOSS_CLS": loss_cls})
        a, b = cls_o.clone().requires_grad_(True), reg_o.clone().requires_grad_(True)
        loss = tf.get_rpn_loss(a, b, cls_l, reg_l, cfg)
        loss.backward()
        assert abs(0.5 * (res[0][1] + res[1][1]) - float(loss.item())) <= 1e-5 * max(1.0, abs(float(loss.item()))), loss_cls
        for r in range(2):
            for got, want in ((res[r][2], a.grad[r:r + 1].numpy()), (res[r][3], b.grad[r:r + 1].numpy())):
                assert np.abs(got / 2 - want).max() <= 1e-6 * max(1.0, float(np.abs(want).max())), (loss_cls, r)


def _rcnn_case(world, R=96, seed=8):
    g = torch.Generator().manual_seed(seed)
    cls_o = torch.randn(world * R, 1, generator=g)
    reg_o = torch.randn(world * R, 46, generator=g) * 0.3
    label = (torch.rand(world * R, generator=g) < 0.3).long() - (torch.rand(world * R, generator=g) < 0.1).long()
    valid = ((label > 0) | (torch.rand(world * R, generator=g) < 0.05)).long()
    valid[R: R + R // 2] = 0                                          # ragged counts; with empty_rank: no regression target at all there
    rois = torch.zeros(world * R, 7)
    rois[:, 3:6] = torch.rand(world * R, 3, generator=g) + 1.0
    gt = torch.randn(world * R, 7, generator=g) * 0.4
    gt[:, 3:6] = gt[:, 3:6].abs() + 1.0
    return cls_o, reg_o, label, valid, rois, gt


def _rcnn_ret(case, sl, leaf=True):
    a, b = case[0][sl].clone().requires_grad_(leaf), case[1][sl].clone().requires_grad_(leaf)
    return a, b, {"rcnn_cls": a, "rcnn_reg": b, "cls_label": case[2][sl], "reg_valid_mask": case[3][sl], "roi_boxes3d": case[4][sl],
                  "gt_of_rois": case[5][sl]}


def _rcnn_loss_worker(rank, world, port, q, bn_eval, empty_rank, loss_cls):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, TESTS)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointrcnn_amd import train_functions as tf
    from pointrcnn_amd.rcnn import RCNNConfig
    case = _rcnn_case(world)
    if empty_rank is not None:
        R = case[0].shape[0] // world
        case[3][empty_rank * R:(empty_rank + 1) * R] = 0
    R = case[0].shape[0] // world
    a, b, ret = _rcnn_ret(case, slice(rank * R, (rank + 1) * R))
    loss = tf.get_rcnn_loss(ret, type("C", (RCNNConfig,), {"LOSS_CLS": loss_cls}), dist=dist)
    loss.backward()
    q.put((rank, float(loss.item()), a.grad.numpy(), b.grad.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_rcnn_loss_is_the_global_batch_loss_under_data_parallel():
    """get_rcnn_loss with `dist`: world x the DDP-averaged gradient is the gradient of the one loss the reference's DataParallel
    computes over the gathered RoIs -- BCE and focal classification, ragged foreground counts, a rank without any regression
    target (which must still take part in the count exchange and keep its regression head in the graph)"""
    from pointrcnn_amd import train_functions as tf
    from pointrcnn_amd.rcnn import RCNNConfig
    for loss_cls, empty_rank in (("BinaryCrossEntropy", None), ("SigmoidFocalLoss", None), ("BinaryCrossEntropy", 1)):
        res = _run(True, empty_rank, target=_rcnn_loss_worker, extra=(loss_cls,))
        case = _rcnn_case(2)
        R = case[0].shape[0] // 2
        if empty_rank is not None:
            case[3][empty_rank * R:(empty_rank + 1) * R] = 0
        a, b, ret = _rcnn_ret(case, slice(None))
        loss = tf.get_rcnn_loss(ret, type("C", (RCNNConfig,), {"LOSS_CLS": loss_cls}))
        loss.backward()
        assert abs(0.5 * (res[0][1] + res[1][1]) - float(loss.item())) <= 1e-5 * max(1.0, abs(float(loss.item()))), loss_cls
        for r in range(2):
            for got, want in ((res[r][2], a.grad[r * R:(r + 1) * R].numpy()), (res[r][3], b.grad[r * R:(r + 1) * R].numpy())):
                assert np.abs(got / 2 - want).max() <= 1e-6 * max(1.0, float(np.abs(want).max())), (loss_cls, r)


def test_ddp_with_training_mode_batchnorm_keeps_ranks_in_sync():
    res = _run(bn_eval=False)
    assert np.array_equal(res[0][3], res[1][3])
    for n in res[0][2]:
        assert np.array_equal(res[0][2][n], res[1][2][n]), n
    assert np.isfinite(res[0][1]) and np.isfinite(res[1][1])
