"""`from tensorboardX import SummaryWriter` (tools/eval_rcnn.py:22, train_rcnn.py:7): absent in this image.  Environment shim: scalars
are dropped."""


class SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass
