"""TF-free restatement of the three `utils_jgm.toolbox` helpers the hot path's callers
rely on (reference imports: ecog2txt/subjects.py:16, trainers.py:19): `auto_attribute`,
`str2int_hook`, `wer_vector`, plus a small mutable record type.  `utils_jgm` is not vendored
by the reference (README.md:19-23); the semantics below are restated from its USES."""
import functools
import inspect

import numpy as np


def auto_attribute(method=None, *, CHECK_MANIFEST=False):
    """Decorator for __init__: assign every argument to self.<name>.

    Semantics restated from usage (subjects.py:28-47, 281-295; data_generators.py:47-73;
    README.md:42):
      * every named parameter is set as an attribute (through property setters when the
        class defines them, hence the `_name` shadow-variable idiom of the reference);
      * with CHECK_MANIFEST=True the first positional argument after self is a dict
        `manifest`; a parameter whose value is None takes manifest[name] when present;
      * parameters whose names start with '_' are NOT assigned (subjects.py:45-46);
      * assignment happens BEFORE the body of __init__ runs.
    """
    def decorate(init):
        sig = inspect.signature(init)

        @functools.wraps(init)
        def wrapper(self, *args, **kwargs):
            bound = sig.bind(self, *args, **kwargs)
            bound.apply_defaults()
            items = list(bound.arguments.items())[1:]
            manifest = items[0][1] if (CHECK_MANIFEST and items) else None
            for name, value in items:
                if name.startswith('_'):
                    continue
                if sig.parameters[name].kind in (inspect.Parameter.VAR_POSITIONAL, inspect.Parameter.VAR_KEYWORD):
                    continue
                if CHECK_MANIFEST and value is None and isinstance(manifest, dict) and name in manifest:
                    value = manifest[name]
                setattr(self, name, value)
            return init(self, *args, **kwargs)
        return wrapper
    if method is not None:
        return decorate(method)
    return decorate


def str2int_hook(d):
    """json object_hook: keys that look like integers become ints (subjects.py:72-74)."""
    out = {}
    for k, v in d.items():
        try:
            k = int(k)
        except (TypeError, ValueError):
            pass
        out[k] = v
    return out


def edit_distance(ref, hyp):
    """Levenshtein distance between two token lists."""
    n, m = len(ref), len(hyp)
    prev = list(range(m + 1))
    for i in range(1, n + 1):
        cur = [i] + [0] * m
        ri = ref[i - 1]
        for j in range(1, m + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ri != hyp[j - 1]))
        prev = cur
    return prev[m]


def wer_vector(references, hypotheses):
    """Per-pair word error rate: edit distance / reference length (usage: subjects.py:546-552,
    plotters.py:1229-1232).  Sequences are lists of tokens (or strings, split on whitespace)."""
    out = []
    for ref, hyp in zip(references, hypotheses):
        if isinstance(ref, str):
            ref = ref.split()
        if isinstance(hyp, str):
            hyp = hyp.split()
        out.append(edit_distance(list(ref), list(hyp)) / max(len(ref), 1))
    return np.array(out, dtype=np.float64)


class MutableNamedTuple:
    """Attribute record with a fixed field list (trainers.py:770-771 subclasses it with __slots__)."""
    __slots__ = []

    def __init__(self, **kwargs):
        for k in self.__slots__:
            setattr(self, k, kwargs.pop(k, None))
        if kwargs:
            raise TypeError('unexpected fields: %s' % sorted(kwargs))

    def __repr__(self):
        return '%s(%s)' % (type(self).__name__, ', '.join('%s=%r' % (k, getattr(self, k)) for k in self.__slots__))
