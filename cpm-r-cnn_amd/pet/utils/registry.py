"""String-keyed factories (counterpart of pet/utils/registry.py:6-39): `@REG.register("name")`."""


class Registry(dict):
    def register(self, name, fn=None):
        if fn is not None:
            self[name] = fn
            return fn

        def deco(f):
            self[name] = f
            return f
        return deco
