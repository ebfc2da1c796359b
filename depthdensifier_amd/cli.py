"""Dataclass command line with tyro's flag spellings (``tyro`` itself is used when importable).

The reference's entry points are ``tyro.cli(main)`` (``scripts/test.py:373-374``,
``scripts/run_batch.py:113-114``): nested dataclasses become dotted kebab-case flags, e.g.
``--root-dir``, ``--output-dir``, ``--config.filtering.vote-threshold 3``,
``--config.processing.downsample-density 1``.  This module accepts exactly those spellings (and
the ``--no-x`` / ``--x`` form for booleans) without the dependency.
"""

from __future__ import annotations

import argparse
import dataclasses
import sys
import typing
from pathlib import Path
from typing import Any, Optional, Sequence


def _kebab(s: str) -> str:
    return s.replace("_", "-")


def _unwrap_optional(tp):
    if typing.get_origin(tp) is typing.Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        if len(args) == 1:
            return args[0], True
    return tp, False


def _add_fields(parser: argparse.ArgumentParser, cls, prefix: str, dests: list) -> None:
    hints = typing.get_type_hints(cls)
    for f in dataclasses.fields(cls):
        tp, optional = _unwrap_optional(hints[f.name])
        flag = f"{prefix}{_kebab(f.name)}"
        if dataclasses.is_dataclass(tp):
            _add_fields(parser, tp, flag + ".", dests)
            continue
        required = f.default is dataclasses.MISSING and f.default_factory is dataclasses.MISSING
        dest = flag.replace(".", "__").replace("-", "_")
        dests.append((flag, dest))
        if tp is bool:
            parser.add_argument(f"--{flag}", dest=dest, action="store_true", default=None)
            head, _, tail = flag.rpartition(".")
            parser.add_argument(f"--{head + '.' if head else ''}no-{tail}", dest=dest, action="store_false", default=None)
        else:
            conv = Path if tp is Path else tp if tp in (int, float, str) else str
            parser.add_argument(f"--{flag}", dest=dest, type=conv, required=required, default=None)


def _build(cls, ns: argparse.Namespace, prefix: str):
    hints = typing.get_type_hints(cls)
    kwargs = {}
    for f in dataclasses.fields(cls):
        tp, _ = _unwrap_optional(hints[f.name])
        flag = f"{prefix}{_kebab(f.name)}"
        if dataclasses.is_dataclass(tp):
            kwargs[f.name] = _build(tp, ns, flag + ".")
            continue
        val = getattr(ns, flag.replace(".", "__").replace("-", "_"))
        if val is not None:
            kwargs[f.name] = val
    return cls(**kwargs)


def parse(cls, argv: Optional[Sequence[str]] = None, description: Optional[str] = None):
    """Instance of dataclass ``cls`` from the command line."""
    parser = argparse.ArgumentParser(description=description or cls.__doc__)
    dests: list = []
    _add_fields(parser, cls, "", dests)
    ns = parser.parse_args(argv)
    return _build(cls, ns, "")


def cli(main, argv: Optional[Sequence[str]] = None) -> Any:
    """``tyro.cli(main)`` for a ``main(config: SomeDataclass)`` function."""
    try:
        import tyro  # noqa: F401
        return tyro.cli(main, args=argv)
    except ImportError:
        pass
    hints = typing.get_type_hints(main)
    cls = next(v for k, v in hints.items() if k != "return" and dataclasses.is_dataclass(v))   # the config parameter
    return main(parse(cls, sys.argv[1:] if argv is None else argv, description=main.__doc__))
