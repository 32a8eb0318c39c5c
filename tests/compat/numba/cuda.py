"""Launch emulator for the subset of numba.cuda the reference's evaluator uses (tools/kitti_object_eval_python/rotate_iou.py:
`@cuda.jit(sig, device=True, inline=True)` device functions, one `@cuda.jit(sig, fastmath=False)` kernel launched as
`kernel[grid, block, stream](...)`, `cuda.local.array`, `cuda.shared.array`, `cuda.syncthreads`, `blockIdx` / `threadIdx`,
`select_device`, `stream().auto_synchronize()`, `to_device(...).copy_to_host(...)`).  A block runs as `blockDim` Python threads
with a barrier for syncthreads; blocks run one after the other.  TEST INFRASTRUCTURE: slow, only for the few hundred boxes of the
end-to-end test."""
import contextlib
import itertools
import threading

import numpy as np

_tls = threading.local()


class _Dim3:
    def __init__(self, which):
        self._which = which

    def _get(self):
        return getattr(_tls, self._which)

    x = property(lambda self: self._get()[0])
    y = property(lambda self: self._get()[1])
    z = property(lambda self: self._get()[2])


threadIdx, blockIdx, blockDim, gridDim = _Dim3("threadIdx"), _Dim3("blockIdx"), _Dim3("blockDim"), _Dim3("gridDim")


class _Local:
    @staticmethod
    def array(shape, dtype):
        return np.zeros(shape, dtype)


class _Shared:
    @staticmethod
    def array(shape, dtype):
        blk = _tls.block
        k = _tls.shared_calls
        _tls.shared_calls = k + 1
        with blk["lock"]:
            if k not in blk["shared"]:
                blk["shared"][k] = np.zeros(shape, dtype)
            return blk["shared"][k]


local, shared = _Local, _Shared


def syncthreads():
    _tls.block["barrier"].wait()


def _dim3(v):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v,)
    return tuple(int(x) for x in v) + (1,) * (3 - len(v))


class _Kernel:
    def __init__(self, fn):
        self.fn = fn

    def __getitem__(self, cfg):
        grid, block = _dim3(cfg[0]), _dim3(cfg[1])

        def launch(*args):
            nthreads = block[0] * block[1] * block[2]
            for bz, by, bx in itertools.product(range(grid[2]), range(grid[1]), range(grid[0])):
                blk = {"barrier": threading.Barrier(nthreads), "lock": threading.Lock(), "shared": {}}
                errors = []

                def run(tid):
                    _tls.threadIdx = (tid % block[0], (tid // block[0]) % block[1], tid // (block[0] * block[1]))
                    _tls.blockIdx, _tls.blockDim, _tls.gridDim = (bx, by, bz), block, grid
                    _tls.block, _tls.shared_calls = blk, 0
                    try:
                        self.fn(*args)
                    except BaseException as e:  # noqa: BLE001
                        errors.append(e)
                        blk["barrier"].abort()
                threads = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
                for t in threads:
                    t.start()
                for t in threads:
                    t.join()
                real = [e for e in errors if not isinstance(e, threading.BrokenBarrierError)]
                if real:
                    raise real[0]
        return launch


def jit(*args, **kwargs):
    """device=True: the function itself; otherwise a kernel object to be launched with [grid, block(, stream)]"""
    def wrap(f):
        return f if kwargs.get("device") else _Kernel(f)
    if len(args) == 1 and callable(args[0]) and not isinstance(args[0], str):
        return wrap(args[0])
    return wrap


class _DeviceArray:
    def __init__(self, a):
        self.a = np.array(a, copy=True)

    def copy_to_host(self, ary=None, stream=None):
        if ary is None:
            return self.a.copy()
        ary[...] = self.a
        return ary

    def __getitem__(self, k):
        return self.a[k]

    def __setitem__(self, k, v):
        self.a[k] = v

    def __len__(self):
        return len(self.a)


def to_device(a, stream=None):
    return _DeviceArray(a)


class _Stream:
    @contextlib.contextmanager
    def auto_synchronize(self):
        yield self

    def synchronize(self):
        pass


def stream():
    return _Stream()


def select_device(device_id):
    return None
