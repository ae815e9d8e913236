"""Attribute dictionary with the interface of the `easydict` package (reference config.py:3: `from easydict import
EasyDict as edict`): nested dicts become EasyDicts, keys are attributes.  Independent 20-line implementation."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __delattr__(self, k):
        del self[k]

    def update(self, e=None, **f):
        d = dict(e or {})
        d.update(f)
        for k, v in d.items():
            self[k] = v

    def pop(self, k, *a):
        return super().pop(k, *a)

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]
