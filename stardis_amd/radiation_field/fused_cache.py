"""Per-object cache of what the fused call derives from the plasma's pandas tables, verified by content.

What `fused_inputs` derives from the plasma (sorted line tables, level tables, density vectors) is kept per OBJECT and VERIFIED
BY CONTENT: an entry is found by the identity of the frames / series it was read from (a strong reference is held, so an id
cannot be recycled) and served only if a 128-bit digest of the values it was derived from (_witness: xxh3 over the objects'
value blocks, ~40 us per MB) still matches — a table edited in place between two calls is derived again, as the reference
(radiation_field/base.py:71-117 recomputes everything per call) would see the edit.  Cached values are private copies (nothing
in them aliases the plasma's memory).  A handful of entries, least recently used first out.

The knobs (CACHE, _MEMO_MAX, MEMO_MAX_BYTES) live on the public module `stardis_amd.radiation_field.fused` and are read there
on every call.
"""
import threading

import numpy as np

_MEMO = {}

try:
    import xxhash as _xx

    def _digest(buf):
        return _xx.xxh3_128_digest(buf)
except ImportError:  # (no xxhash: correct, ~1 GB/s instead of ~20)
    import hashlib as _hl

    def _digest(buf):
        return _hl.blake2b(buf, digest_size=16).digest()


def _knob(name):
    from stardis_amd.radiation_field import fused

    return getattr(fused, name)


def clear_cache():
    _MEMO.clear()


class _Same:
    """Compares equal only to a wrapper of the very same object: the immutable parts of a pandas object (its Index objects,
    which pandas replaces rather than edits) inside a witness tuple."""

    __slots__ = ("obj",)

    def __init__(self, obj):
        self.obj = obj

    def __eq__(self, other):
        return type(other) is _Same and other.obj is self.obj

    __hash__ = None


class _Never:
    """A witness that matches nothing (an object whose content cannot be read as arrays): derived again on every call."""

    def __eq__(self, other):
        return False

    __hash__ = None


def _hash_array(a):
    if not isinstance(a, np.ndarray):
        a = np.asarray(a)  # (pandas extension arrays)
    if a.dtype.kind == "O":  # strings / mixed objects: pandas' value hash per element (a buffer of pointers says nothing)
        import pandas as pd

        return (a.shape, "O", _digest(np.ascontiguousarray(pd.util.hash_array(a.reshape(-1), categorize=False))))
    if a.flags.c_contiguous:
        d = _digest(a)
    elif a.flags.f_contiguous:  # a DataFrame block made from a row-major table
        d = _digest(a.T)
    else:
        d = _digest(np.ascontiguousarray(a))
    return (a.shape, a.dtype.str, d)


def _witness(obj):
    """What a cached derivation of `obj` is checked against: a digest of every value it holds (+ the identity of its axes).
    ~1 us for a per-depth vector, ~40 us per MB of table."""
    if isinstance(obj, np.ndarray):
        return _hash_array(obj)
    kind = type(obj).__name__
    try:
        if kind in ("DataFrame", "Series"):
            mgr = getattr(obj, "_mgr", None)
            arrays = getattr(mgr, "arrays", None)
            if arrays is None:
                arrays = (obj.to_numpy(),)
            parts = [_Same(obj.index)]
            if kind == "DataFrame":
                parts.append(_Same(obj.columns))
                for name in ("blknos", "blklocs"):  # which column sits where in which block
                    loc = getattr(mgr, name, None)
                    if loc is not None:
                        parts.append(_hash_array(np.asarray(loc)))
            parts.extend(_hash_array(a) for a in arrays)
            return tuple(parts)
        if hasattr(obj, "is_unique") and hasattr(obj, "get_indexer"):  # a pandas Index: immutable
            return (_Same(obj),)
    except Exception:  # noqa: BLE001  (anything exotic: not cacheable)
        pass
    return _Never()


_CALL = threading.local()  # .seen: {id(object): witness} for the duration of ONE try_fused call ON THIS THREAD (several derivations read the same tables)


class one_call:
    """with one_call(): the witnesses of the objects read inside are computed once (the tables may be edited before the next
    call, so nothing outlives the block; per thread, so concurrent calls never see each other's witnesses)."""

    def __enter__(self):
        self.outer = getattr(_CALL, "seen", None)
        _CALL.seen = {}

    def __exit__(self, *exc):
        _CALL.seen = self.outer


def _witness_once(obj):
    seen = getattr(_CALL, "seen", None)
    if seen is None:
        return _witness(obj)
    w = seen.get(id(obj))
    if w is None:
        w = seen[id(obj)] = _witness(obj)
    return w


def _memo(tag, objects, extra, build, private=()):
    """build() for these source objects, or the value kept from an earlier call if the objects are the same AND hold the same
    values.  `private`: positions in `objects` of values this module made itself (cached derivations handed on): identity only."""
    if not _knob("CACHE"):
        return build()
    key = (tag, tuple(id(o) for o in objects), extra)
    now = tuple(_Same(o) if i in private else _witness_once(o) for i, o in enumerate(objects))
    hit = _MEMO.get(key)
    if hit is not None and hit[3] == now:
        _MEMO[key] = _MEMO.pop(key)  # most recently used last
        return hit[1]
    _MEMO.pop(key, None)
    value = build()
    _MEMO[key] = (objects, value, _nbytes(value), now)
    while len(_MEMO) > _knob("_MEMO_MAX") or (len(_MEMO) > 1 and sum(e[2] for e in _MEMO.values()) > _knob("MEMO_MAX_BYTES")):
        _MEMO.pop(next(iter(_MEMO)))
    return value


def _nbytes(value, depth=0):
    """numpy bytes reachable from a cached value (tuples, lists, dicts, objects with a __dict__), for the cache's byte bound"""
    if isinstance(value, np.ndarray):
        return value.nbytes
    if depth > 4:
        return 0
    if isinstance(value, dict):
        return sum(_nbytes(v, depth + 1) for v in value.values())
    if isinstance(value, (tuple, list)):
        return sum(_nbytes(v, depth + 1) for v in value)
    if hasattr(value, "__dict__"):
        return sum(_nbytes(v, depth + 1) for v in vars(value).values())
    return 0
