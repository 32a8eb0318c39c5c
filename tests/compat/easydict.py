"""Minimal stand-in for the `easydict` package (absent in this image; the reference's lib/config.py:1 imports it).
TEST INFRASTRUCTURE: lets tests import the reference's unchanged lib/ when /root/reference is present."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {}, **kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, dict) and not isinstance(value, EasyDict):
            value = EasyDict(value)
        elif isinstance(value, (list, tuple)):
            value = type(value)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__
