timeout 300 python -m pytest tests/test_gpu_pointnet2_ops.py -x -q -m gpu 2>&1 | tail -2
python - <<'PY'
import os, torch
from pointrcnn_amd import ops, rpn
from pointrcnn_amd.opbench import timeit
dev = torch.device("cuda:0")
B, N = 32, 16384
xyz = rpn.synthetic_clouds(B, N, device=dev)
xyz1 = ops.gather_rows(xyz, ops.furthest_point_sample(xyz, 4096))
d2, i3, w3 = ops.three_nn(xyz, xyz1, want_weight=True)
for C in (256, 128, 512):
    kf = torch.randn(B, C, 4096, device=dev)
    for lay in ("pm", "rows"):
        if lay == "rows": os.environ["PRCNN_INTERP_LAYOUT"] = "rows"
        else: os.environ.pop("PRCNN_INTERP_LAYOUT", None)
        t = timeit(lambda: ops.three_interpolate(kf, i3, w3))
        comp = B * (N * 24 + C * 4096 * 4 + C * N * 4)
        print("three_interpolate C=%d %s: %.1f us  %.3f of 6.3 TB/s (compulsory)" % (C, lay, t * 1e6, comp / t / 6.3e12))
PY
