"""Environment shim, imported by `site` at interpreter start when tests/compat is on PYTHONPATH: the two py3.10 / PyYAML 6
incompatibilities of the reference's UNCHANGED sources that no stand-in package can cover:
  * lib/config.py:187 `yaml.load(f)` -- PyYAML >= 6 requires a Loader;
  * tools/train_utils/fastai_optim.py:3 `from collections import Iterable` -- moved to collections.abc in py3.10."""
import collections
import collections.abc

if not hasattr(collections, "Iterable"):
    collections.Iterable = collections.abc.Iterable

try:
    import yaml
except ImportError:          # nothing to adapt
    yaml = None
if yaml is not None and not getattr(yaml.load, "_prcnn_compat", False):
    _load = yaml.load

    def load(stream, Loader=None, **kw):
        return _load(stream, Loader=Loader or yaml.FullLoader, **kw)
    load._prcnn_compat = True
    yaml.load = load
