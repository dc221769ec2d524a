"""``tyro.cli`` for (nested, possibly frozen) dataclasses -- the subset used by
reference ``configs.py`` and ``scripts/dino_v2_vlad.py:63-119`` / ``demo/anyloc_vlad_generate.py``:
kebab-case flags, dotted nesting (``--prog.vg-dataset-name``), ``Literal``, ``Union[X, None]``,
``List[int]`` (nargs +), ``--flag / --no-flag`` booleans, ``Path``/str/int/float."""
import argparse
import dataclasses
import sys
import typing
from pathlib import Path


def _unwrap_optional(tp):
    if typing.get_origin(tp) is typing.Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        if len(args) == 1:
            return args[0], True
    return tp, False


def _conv(tp):
    if tp in (int, float, str):
        return tp
    if tp is Path:
        return Path
    return str


def _add(parser, cls, prefix, defaults, registry):
    hints = typing.get_type_hints(cls)
    for f in dataclasses.fields(cls):
        tp = hints.get(f.name, str)
        if f.default is not dataclasses.MISSING:
            default = f.default
        elif f.default_factory is not dataclasses.MISSING:      # type: ignore[attr-defined]
            default = f.default_factory()                         # type: ignore[misc]
        else:
            default = dataclasses.MISSING
        if prefix + f.name in defaults:
            default = defaults[prefix + f.name]
        name = prefix + f.name
        flag = "--" + name.replace("_", "-")
        base, optional = _unwrap_optional(tp)
        if dataclasses.is_dataclass(base):
            sub_defaults = {}
            if default is not dataclasses.MISSING and default is not None:
                for sf in dataclasses.fields(base):
                    sub_defaults[name + "." + sf.name] = getattr(default, sf.name)
            registry[name] = ("dataclass", base)
            _add(parser, base, name + ".", {**defaults, **sub_defaults}, registry)
            continue
        origin = typing.get_origin(base)
        kw = dict(dest=name, default=default if default is not dataclasses.MISSING else None,
                  required=default is dataclasses.MISSING)
        if base is bool:
            parser.add_argument(flag, dest=name, action="store_true", default=bool(default))
            parser.add_argument("--" + (prefix + "no_" + f.name).replace("_", "-"), dest=name,
                                action="store_false")
        elif origin is typing.Literal:
            choices = list(typing.get_args(base))
            parser.add_argument(flag, choices=choices, type=type(choices[0]), **kw)
        elif origin in (list, typing.List):
            (elem,) = typing.get_args(base) or (str,)
            parser.add_argument(flag, nargs="+", type=_conv(elem), **kw)
        elif origin is dict or base is dict:
            # a dict field with a default becomes one flag per key (tyro: --db-samples.st-lucia 1)
            dflt = default if isinstance(default, dict) else {}
            for key, val in dflt.items():
                parser.add_argument(f"{flag}.{str(key).replace('_', '-')}", dest=f"{name}.{key}",
                                    type=type(val) if val is not None else str, default=val)
            registry[name] = ("dict", list(dflt.keys()))
            continue
        elif origin in (tuple, typing.Tuple):
            targs = [t for t in typing.get_args(base) if t is not Ellipsis] or [str]
            parser.add_argument(flag, nargs=len(typing.get_args(base)) if Ellipsis not in typing.get_args(base) else "+",
                                type=_conv(targs[0]), **kw)
            registry[name] = ("tuple", None)
            continue
        else:
            conv = _conv(base)
            if optional:
                parser.add_argument(flag, type=lambda s, c=conv: None if s == "None" else c(s), **kw)
            else:
                parser.add_argument(flag, type=conv, **kw)
        registry[name] = ("leaf", None)


def _build(cls, prefix, ns, registry):
    kwargs = {}
    for f in dataclasses.fields(cls):
        name = prefix + f.name
        kind, sub = registry[name]
        if kind == "dataclass":
            kwargs[f.name] = _build(sub, name + ".", ns, registry)
        elif kind == "dict":
            kwargs[f.name] = {k: getattr(ns, f"{name}.{k}") for k in sub}
        elif kind == "tuple":
            v = getattr(ns, name)
            kwargs[f.name] = tuple(v) if isinstance(v, list) else v
        else:
            kwargs[f.name] = getattr(ns, name)
    return cls(**kwargs)


def cli(cls, *, description=None, args=None, default=None, **_ignored):
    if not dataclasses.is_dataclass(cls):
        raise TypeError("this tyro stand-in only parses dataclasses")
    parser = argparse.ArgumentParser(description=description or cls.__doc__)
    registry = {}
    defaults = {}
    if default is not None:
        for f in dataclasses.fields(cls):
            defaults[f.name] = getattr(default, f.name)
    _add(parser, cls, "", defaults, registry)
    ns = parser.parse_args(sys.argv[1:] if args is None else args)
    return _build(cls, "", ns, registry)
