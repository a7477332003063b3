"""The plugin seam: resolve "pkg.mod.Class" strings from the settings JSON (reference sys_tools.py:7-22)."""
import importlib
import sys


def find_class(class_str):
    """`mod, _, cls = class_str.rpartition('.')`; import mod; return its attribute cls.
    Raises ImportError if the class is missing, like the reference."""
    mod_str, _sep, cls = class_str.rpartition(".")
    if not mod_str:
        raise ImportError("Class path %r has no module part" % class_str)
    importlib.import_module(mod_str)
    try:
        return getattr(sys.modules[mod_str], cls)
    except AttributeError:
        raise ImportError("Class %s cannot be found in %s" % (cls, mod_str))


def create_object(class_str, *args, **kwargs):
    return find_class(class_str)(*args, **kwargs)
