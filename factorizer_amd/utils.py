"""Constructor plumbing shared by every module: the ``(cls, {kwargs})`` partial-module
convention of the reference (utils/helpers.py:36-51, 91-147).  YAML configs pass lists, user
code passes tuples; both must work."""
from __future__ import annotations

import functools
import inspect
from collections.abc import Sequence
from typing import Any, Callable


def as_tuple(obj: Any) -> tuple:
    """Sequence (not str) -> tuple; anything else -> 1-tuple (helpers.py:36-51)."""
    if isinstance(obj, Sequence) and not isinstance(obj, str):
        return tuple(obj)
    return (obj,)


def has_args(obj: Any, keywords) -> bool:
    """True if callable `obj` accepts all of `keywords` (helpers.py:67-88)."""
    if not callable(obj):
        return False
    try:
        params = inspect.signature(obj).parameters
    except ValueError:
        return False
    return all(k in params for k in as_tuple(keywords))


def is_partializable(obj: Any) -> bool:
    """helpers.py:131-147."""
    if callable(obj):
        return True
    return isinstance(obj, Sequence) and not isinstance(obj, str) and len(obj) > 0 and callable(obj[0])


def partialize(spec) -> Callable:
    """``cls`` -> cls ; ``(cls, (args...), {kwargs}, scalar...)`` -> functools.partial
    (helpers.py:91-128): dict items update kwargs, non-str sequences extend args, anything
    else is appended as one positional arg."""
    if callable(spec):
        return spec
    if isinstance(spec, Sequence) and not isinstance(spec, str) and spec and callable(spec[0]):
        args, kwargs = [], {}
        for item in spec[1:]:
            if isinstance(item, dict):
                kwargs.update(item)
            elif isinstance(item, Sequence) and not isinstance(item, str):
                args.extend(item)
            else:
                args.append(item)
        return functools.partial(spec[0], *args, **kwargs)
    raise TypeError(f"Expected a callable or valid tuple, got {type(spec).__name__}")


class TensorSpec:
    """Shape, dtype and device of a tensor that does not exist yet — what the dispatch predicates (`_fusable`, `prologue_params`,
    ...) read of their argument.  A producer asks "would the consumer take the native path for my output?" without allocating it."""

    def __init__(self, shape, dtype, device):
        self.shape, self.dtype, self.device = tuple(int(s) for s in shape), dtype, device

    @property
    def is_cuda(self):
        return self.device.type == "cuda"

    def dim(self):
        return len(self.shape)

    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n
