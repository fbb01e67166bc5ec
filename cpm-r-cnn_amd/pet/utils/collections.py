"""Attribute-style dict used for the global cfg tree (counterpart of pet/utils/collections.py)."""


class AttrDict(dict):
    _LOCK = "__frozen__"

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        object.__setattr__(self, AttrDict._LOCK, False)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, AttrDict._LOCK):
            raise AttributeError('Attempted to set "{}" to "{}", but AttrDict is immutable'.format(name, value))
        self[name] = value

    def immutable(self, flag):
        object.__setattr__(self, AttrDict._LOCK, bool(flag))
        for v in self.values():
            if isinstance(v, AttrDict):
                v.immutable(flag)

    def is_immutable(self):
        return object.__getattribute__(self, AttrDict._LOCK)
