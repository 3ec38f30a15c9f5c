"""Module / ModuleList (T/tensorrt_llm/module.py): a tree of sub-modules and Parameters; calling a module runs its
forward() inside the current network (define-and-run)."""
from .parameter import Parameter


class Module(object):

    def __init__(self) -> None:
        object.__setattr__(self, '_modules', {})
        object.__setattr__(self, '_parameters', {})
        object.__setattr__(self, '_network_outputs', {})

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def __setattr__(self, name, value) -> None:
        mods, params = self.__dict__.get('_modules'), self.__dict__.get('_parameters')
        if isinstance(value, Parameter):
            if params is None:
                raise AttributeError('assign parameters after Module.__init__()')
            mods.pop(name, None)
            self.__dict__.pop(name, None)
            params[name] = value
        elif isinstance(value, Module):
            if mods is None:
                raise AttributeError('assign sub-modules after Module.__init__()')
            params.pop(name, None)
            self.__dict__.pop(name, None)
            mods[name] = value
        else:
            if mods is not None and name in mods:
                del mods[name]
            if params is not None and name in params:
                del params[name]
            object.__setattr__(self, name, value)

    def __getattr__(self, name):
        d = self.__dict__
        if '_parameters' in d and name in d['_parameters']:
            return d['_parameters'][name]
        if '_modules' in d and name in d['_modules']:
            return d['_modules'][name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def register_parameter(self, name, param):
        if param is None:
            self.__dict__['_parameters'].pop(name, None)
            object.__setattr__(self, name, None)
        else:
            setattr(self, name, param)

    def register_network_output(self, name, value):
        self._network_outputs[name] = value

    def named_modules(self, prefix=''):
        yield prefix, self
        for name, m in self._modules.items():
            if m is None:
                continue
            yield from m.named_modules(prefix + ('.' if prefix else '') + name)

    def named_children(self):
        return list(self._modules.items())

    def named_parameters(self, prefix=''):
        for mod_prefix, m in self.named_modules(prefix):
            for name, p in m._parameters.items():
                if p is not None:
                    yield mod_prefix + ('.' if mod_prefix else '') + name, p

    def parameters(self):
        return [p for _, p in self.named_parameters()]

    def children(self):
        return list(self._modules.values())


class ModuleList(Module):

    def __init__(self, modules) -> None:
        super().__init__()
        for i, m in enumerate(modules):
            self._modules[str(i)] = m

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return ModuleList(list(self._modules.values())[idx])
        n = len(self._modules)
        if idx < 0:
            idx += n
        return self._modules[str(idx)]

    def __setitem__(self, idx, module):
        self._modules[str(idx)] = module

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())
