"""Which side of test_sa_module_fused_training_equals_composed_torch_path moved in round 4?  

One SA-module training step (the test's C = 0 case: 3-channel first layer; and C = 8), 50 repetitions of each of
  fused      -- the hand-written path (csrc/mlp_train.h)
  miopen     -- the composed path with torch.backends.cudnn enabled (MIOpen convolution / batch norm: what the test compared with)
  aten       -- the composed path with torch.backends.cudnn disabled (ATen's own kernels: what the test compares with now)
against the float64 CPU autograd restatement tests/test_gpu_train_mlp.py::sa_reference_f64.  Prints, per side, the largest (over the
repetitions) median and max relative deviation of every weight gradient, and whether the repetitions were bit-identical.

    python tools/train_flake_probe.py [reps] > gpurun_out/train_flake.json
"""
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main(reps=50):
    import test_gpu_train_mlp as T
    dev = torch.device("cuda", 0)
    out = {}
    for C in (0, 8):
        pm, m0, _ = T._modules("sa", dev, 5 + C, npoint=200, radii=[0.15, 0.3], nsamples=[16, 32],
                               mlps=[[C, 16, 16, 32], [C, 32, 48, 64]], use_xyz=True, bn=True)
        g = torch.Generator().manual_seed(C)
        xyz = torch.rand(3, 1500, 3, generator=g).to(dev)
        feat = None if C == 0 else torch.randn(3, C, 1500, generator=g).to(dev)
        gout = torch.randn((3, 96, 200), generator=g).to(dev)
        _, _, g64, _ = T.sa_reference_f64(m0, xyz, feat, gout)
        for side in ("fused", "miopen", "aten"):
            worst, first, same = {}, None, True
            for rep in range(reps):
                m = copy.deepcopy(m0)
                fa = None if feat is None else feat.clone().requires_grad_(True)
                pm.TRAIN_FUSED = side == "fused"
                try:
                    with torch.backends.cudnn.flags(enabled=side != "aten"):
                        _, o = m(xyz, fa)
                        o.backward(gout)
                finally:
                    pm.TRAIN_FUSED = True
                grads = {n: p.grad.double().cpu().reshape(g64[n].shape) for n, p in m.named_parameters()}
                for n, v in grads.items():
                    d = (v - g64[n]).abs().reshape(-1)
                    sc = float(g64[n].abs().max())
                    w = worst.setdefault(n, [0.0, 0.0])
                    w[0], w[1] = max(w[0], float(d.median()) / sc), max(w[1], float(d.max()) / sc)
                if first is None:
                    first = grads
                else:
                    same = same and all(torch.equal(grads[n], first[n]) for n in grads)
            top = max(worst.items(), key=lambda kv: kv[1][0])
            out["C%d_%s" % (C, side)] = {"worst_median_rel": top[1][0], "worst_median_param": top[0],
                                         "worst_max_rel": max(v[1] for v in worst.values()), "bit_identical_over_reps": same}
    print(json.dumps({"reps": reps, "sides": out}, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 50)
