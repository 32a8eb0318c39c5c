"""dev probe: what a pure WRITE stream reaches on this GPU (the ceiling of output-dominated ops such as roipool3d, whose 1.02 GB of
pooled rows are written once and never read): torch fill / copy of 1 GiB buffers, timed with events."""
import torch
dev = torch.device("cuda:0")
n = 1 << 28                      # 1 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)


def t(fn, it=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3


gb = n * 4 / 1e9
print("fill_ (write only)      %.0f GB/s" % (gb / t(lambda: x.fill_(1.0))))
print("zero_ (memset)          %.0f GB/s" % (gb / t(lambda: x.zero_())))
print("copy_ (read + write)    %.0f GB/s moved (%.0f written)" % (2 * gb / t(lambda: y.copy_(x)), gb / t(lambda: y.copy_(x))))
print("sum (read only)         %.0f GB/s" % (gb / t(lambda: x.sum())))
