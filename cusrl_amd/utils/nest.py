"""Flatten / rebuild nested containers of tensors by dotted leaf path.

Behavioural counterpart of cusrl/utils/nest.py:30-306 (``get_schema``, ``iterate_nested``,
``flatten_nested``, ``map_nested``, ``reconstruct_nested``): a mapping key or sequence index becomes one
path segment, segments are joined with ``.``, and an empty prefix / suffix is dropped.
"""

from __future__ import annotations

from collections.abc import Callable, Iterator, Mapping
from typing import Any

__all__ = ["flatten_nested", "get_schema", "iterate_nested", "map_nested", "reconstruct_nested"]


def _join(prefix: Any, key: Any, sep: str) -> str:
    prefix, key = str(prefix), str(key)
    if prefix and key:
        return prefix + sep + key
    return prefix or key


def _children(node: Any):
    """(key, child) pairs of a container node, or None for a leaf."""
    if isinstance(node, Mapping):
        return node.items()
    if isinstance(node, (list, tuple)):
        return enumerate(node)
    return None


def _rebuild_like(node: Any, values: list):
    if isinstance(node, Mapping):
        return dict(zip(node.keys(), values))
    return tuple(values) if isinstance(node, tuple) else list(values)


def get_schema(value: Any, prefix: str = "", max_depth: int | None = None, separator: str = ".") -> Any:
    """Same container structure as ``value`` with every leaf replaced by its dotted path."""
    kids = None if max_depth is not None and max_depth <= 0 else _children(value)
    if kids is None:
        return prefix
    depth = None if max_depth is None else max_depth - 1
    return _rebuild_like(value, [get_schema(v, _join(prefix, k, separator), depth, separator) for k, v in kids])


def iterate_nested(data: Any, prefix: str = "", *, max_depth: int | None = None, separator: str = ".") -> Iterator[tuple[str, Any]]:
    """Yield ``(dotted_path, leaf)`` in depth-first insertion order."""
    kids = None if max_depth is not None and max_depth <= 0 else _children(data)
    if kids is None:
        yield prefix, data
        return
    depth = None if max_depth is None else max_depth - 1
    for key, child in kids:
        yield from iterate_nested(child, _join(prefix, key, separator), max_depth=depth, separator=separator)


def flatten_nested(data: Any, prefix: str = "", *, max_depth: int | None = None, separator: str = ".") -> dict[str, Any]:
    return dict(iterate_nested(data, prefix, max_depth=max_depth, separator=separator))


def reconstruct_nested(flattened: Mapping[str, Any], schema: Any) -> Any:
    """Inverse of flattening: look every schema leaf (a path string) up in ``flattened``."""
    kids = _children(schema)
    if kids is None:
        return flattened[schema]
    return _rebuild_like(schema, [reconstruct_nested(flattened, child) for _, child in kids])


def map_nested(func: Callable[[Any], Any], data: Any) -> Any:
    kids = _children(data)
    if kids is None:
        return func(data)
    return _rebuild_like(data, [map_nested(func, child) for _, child in kids])
