"""`bench.py --workload reference` (bench_reference.py): the reference's unchanged lib/net + the loop body of tools/eval_rcnn.py on the
drop-in, the figure INTEGRATION.md section 1.1 quotes.  A short run of the measurement itself: the reference modules really come from the
reference tree, the operators from the drop-in, the backbone is the mirror's bit for bit, and the line carries what the docs say it carries."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_route_measurement_runs_and_matches_the_mirror(dev):
    sys.path.insert(0, ROOT)
    import bench_reference
    if not bench_reference.available():
        pytest.skip("no reference tree on this box (neither $PRCNN_REFERENCE / /root/reference nor oracle/_ref/reference_py.tar.gz)")
    from pointrcnn_amd import rpn
    import pointrcnn_amd
    torch.manual_seed(1234)
    mirror = rpn.randomize_bn_stats(rpn.RPN(), seed=7).to(dev).eval()
    clouds = rpn.synthetic_clouds(4, 16384, seed0=100)
    with torch.no_grad():
        mo = mirror({"pts_input": clouds.to(dev)})
    rep = bench_reference.measure(dev, clouds, mirror, steps=3, warmup=1, nms_type="normal", mirror_out=mo)
    assert rep["backbone_bit_identical_to_mirror"] is True
    assert rep["rpn_reg_max_abs_diff_vs_mirror"] <= 1e-5 * max(1.0, float(mo["rpn_reg"].abs().max()))
    assert rep["rois_shape"] == [4, 100, 7]
    assert rep["value_model_only"] > 0 and rep["value_eval_loop"] > 0       # (rates of a 3-step loop: reported, not compared -- a hiccup of the box flips any inequality)
    assert set(rep["stage_ms"]) == {"h2d", "backbone", "heads", "seg", "proposal_layer", "d2h"}
    # what ran above the operators is the reference's file, what ran below is this package
    cfg, PointRCNN = bench_reference.load()
    import inspect
    import lib.rpn.proposal_layer as ref_pl
    import iou3d_cuda
    tree = os.path.realpath(bench_reference._reference_tree())
    assert os.path.realpath(inspect.getsourcefile(PointRCNN)).startswith(tree)
    assert os.path.realpath(ref_pl.__file__).startswith(tree)
    assert os.path.realpath(iou3d_cuda.__file__).startswith(os.path.realpath(pointrcnn_amd.DROPIN_DIR))
    assert cfg.RPN.NMS_TYPE == "normal" and cfg.RCNN.ENABLED is False
