"""Minimal gin-config stand-in (gin-config 0.3.0 is not installed on the target image): parses the reference's
``*.gin`` files -- ``Name.param = <python literal>`` lines -- into a binding table and fills configurable
functions from it, which is all the reference uses gin for (train_gan.py:103-121,233-235; augment/*.py)."""
import ast
import functools
import inspect
import os

_BINDINGS = {}
REQUIRED = object()


def clear_config():
    _BINDINGS.clear()


def bind_parameter(key, value):
    scope, param = key.rsplit('.', 1)
    _BINDINGS.setdefault(scope, {})[param] = value


def parse_config(text):
    for raw in text.splitlines():
        line = raw.split('#', 1)[0].strip()
        if not line or '=' not in line:
            continue
        key, val = line.split('=', 1)
        bind_parameter(key.strip(), ast.literal_eval(val.strip()))


def parse_config_files_and_bindings(config_files, bindings=None, skip_unknown=True):
    for f in config_files or []:
        with open(f) as fh:
            parse_config(fh.read())
    for b in bindings or []:
        parse_config(b)


def get_bindings(scope):
    if scope not in _BINDINGS:
        raise KeyError("no gin bindings for '%s' (parse configs/defaults/augment.gin first)" % scope)
    return dict(_BINDINGS[scope])


def configurable(name_or_fn=None, module=None, whitelist=None, blacklist=None):
    def deco(fn, name=None):
        name = name or fn.__name__

        @functools.wraps(fn)
        def wrapper(*a, **kw):
            sig = inspect.signature(fn)
            bound = sig.bind_partial(*a, **kw)
            for k, v in _BINDINGS.get(name, {}).items():
                if k in sig.parameters and k not in bound.arguments and (whitelist is None or k in whitelist):
                    kw[k] = v
            for k, prm in sig.parameters.items():
                if prm.default is REQUIRED and k not in kw and k not in bound.arguments:
                    raise ValueError("required parameter '%s.%s' has no binding" % (name, k))
            return fn(*a, **kw)
        return wrapper
    if callable(name_or_fn):
        return deco(name_or_fn)
    return lambda fn: deco(fn, name_or_fn)


CONFIG_ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs')
