#!/usr/bin/env python3
"""dev probe (round 4): does the fused TRAINING path read allocator blocks it has not written?  Fills the caching
allocator's free blocks with a float pattern, then runs the fused-vs-composed training parity tests in the same process.

    python tools/train_poison_probe.py inf | 1000 | 0"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402
import torch  # noqa: E402

val = float(sys.argv[1]) if len(sys.argv) > 1 else float("inf")
small = [torch.full((128 * 1024,), val, dtype=torch.float32, device="cuda") for _ in range(2048)]      # 512 KB blocks: the small pool
mid = [torch.full((4 << 20,), val, dtype=torch.float32, device="cuda") for _ in range(256)]           # 16 MB blocks
big = torch.full((2 << 30,), val, dtype=torch.float32, device="cuda")                                 # one 8 GB block
torch.cuda.synchronize()
del small, mid, big
print("poisoned the allocator's free blocks with", val, flush=True)
os.environ["PRCNN_TEST_MLP_MODES"] = "f32"
sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_gpu_train_mlp.py"), "-q", "-m", "gpu", "-p", "no:cacheprovider", "--tb=line",
                      "-k", "fused_training_equals_composed"]))
