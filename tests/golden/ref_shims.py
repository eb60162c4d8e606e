"""In-memory stand-ins that let the UNMODIFIED reference (/root/reference) be imported in this
container.  Used ONLY by tests/golden/make_golden.py to generate the
committed fixture vectors; nothing here travels to the GPU box as code that is executed there
(the GPU box has no /root/reference).

Missing third-party modules: `tree` (dm-tree), `Bio`, `esm`, `anarci`, `pyrosetta`, `ml_collections`.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF = os.environ.get('ABX_REFERENCE', '/root/reference')


def _map_structure(fn, nested):
    if isinstance(nested, dict):
        return {k: _map_structure(fn, v) for k, v in nested.items()}
    if isinstance(nested, (list, tuple)):
        return type(nested)(_map_structure(fn, v) for v in nested)
    return fn(nested)


class _AnyMeta(type):
    """Metaclass whose classes answer any attribute with another such class (PDB.Structure.Structure)."""

    def __getattr__(cls, name):
        if name.startswith('__'):
            raise AttributeError(name)
        obj = _AnyMeta(name, (), {'__init__': lambda self, *a, **k: None})
        setattr(cls, name, obj)
        return obj


class _Mock(types.ModuleType):
    """Permissive mock package: any attribute is another mock / dummy class."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        obj = _AnyMeta(name, (), {'__init__': lambda self, *a, **k: None})
        setattr(self, name, obj)
        return obj


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ('Bio', 'esm', 'anarci', 'pyrosetta')

    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Mock(spec.name)

    def exec_module(self, module):
        pass


class ConfigDict(dict):
    """Recursive attribute-access dict (stand-in for ml_collections.ConfigDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = ConfigDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def install():
    os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
    sys.dont_write_bytecode = True
    if 'tree' not in sys.modules:
        m = types.ModuleType('tree')
        m.map_structure = _map_structure
        sys.modules['tree'] = m
    if not any(isinstance(f, _MockFinder) for f in sys.meta_path):
        sys.meta_path.append(_MockFinder())
    if 'ml_collections' not in sys.modules:
        m = types.ModuleType('ml_collections')
        m.ConfigDict = ConfigDict
        sys.modules['ml_collections'] = m
    if REF not in sys.path:
        sys.path.insert(0, REF)
