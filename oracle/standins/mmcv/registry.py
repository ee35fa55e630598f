"""Minimal mmcv.utils.Registry look-alike (plumbing only, no arithmetic)."""
import copy


def build_from_cfg(cfg, registry, default_args=None):
    args = copy.copy(dict(cfg))
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f'{typ} is not in the {registry.name} registry')
    return cls(**args)


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def build(self, cfg, **default_args):
        return build_from_cfg(cfg, self, default_args or None)
