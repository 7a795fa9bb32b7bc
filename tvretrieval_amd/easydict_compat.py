"""Attribute-style dict used for `XML(config)` and for un-pickling reference checkpoints.

The reference stores `model.config` (an `easydict.EasyDict`) inside its checkpoint dict
(xml/train.py:219-223).  `easydict` is not installed on the target image, so a compatible class is
provided here and, only when the real package is absent, registered under the module name
`easydict` so that `torch.load(..., weights_only=False)` can resolve `easydict.EasyDict`.
Missing attributes raise AttributeError (required for copy.deepcopy / pickle protocol probing).
"""
import sys
import types


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        if d is None:
            d = {}
        if kwargs:
            d = dict(d, **kwargs)
        for k, v in d.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return EasyDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(EasyDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, EasyDict._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def __reduce__(self):
        # Always pickle as `easydict.EasyDict` -- the name the reference's checkpoints carry and its loader resolves
        # (xml/train.py:219-223, xml/inference.py:536-540): the real class when the package is installed, this class
        # (registered under the module name `easydict`) when it is not.  Never as tvretrieval_amd.easydict_compat.*.
        mod = register_easydict_module()
        return (mod.EasyDict, (dict(self),))

    def __deepcopy__(self, memo):
        import copy
        return EasyDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def register_easydict_module():
    """Make `import easydict` resolve to this class if the real package is absent."""
    if "easydict" in sys.modules:
        return sys.modules["easydict"]
    try:
        import easydict  # noqa: F401
        return sys.modules["easydict"]
    except ImportError:
        mod = types.ModuleType("easydict")
        mod.EasyDict = EasyDict
        EasyDict.__module__ = "easydict"
        sys.modules["easydict"] = mod
        return mod
