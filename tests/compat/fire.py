"""`import fire` (tools/kitti_object_eval_python/evaluate.py:2; used only by that file's own __main__).  Environment shim."""


def Fire(*a, **k):
    raise RuntimeError("the `fire` CLI is not available in this image")
