"""Test-only stand-in for paramz.caching.

Default: identity (every call recomputes).  With `ENABLED = True` (tools/cpu_reference_vs_port.py sets it for the
"reference_cached" CPU baseline) `Cache_this` memoises the last `limit` results per decorated method, keyed on the
identity of the instance and of its array arguments plus the exact bits of the instance's parameters -- what the real
paramz Cacher achieves through observers for `Stationary.K`, `_scaled_dist` and `dK_dr_via_X`
(reference `GPy/kern/src/stationary.py:105,117,150`, `@Cache_this(limit=3, ignore_args=())`): the gradient step of one
`GP.parameters_changed` reuses the K and r of the inference step."""
import numpy as np

ENABLED = False


def _param_bits(obj):
    ps = getattr(obj, "parameters", None)
    if not ps:
        return b""
    return b"|".join(np.asarray(p, dtype=float).tobytes() for p in ps)


def Cache_this(limit=5, ignore_args=(), force_kwargs=()):
    def deco(f):
        store = []                       # [(key, value)], most recent last

        def wrapper(self, *args, **kw):
            if not ENABLED or kw:
                return f(self, *args, **kw)
            key = (id(self), tuple(id(a) for a in args), _param_bits(self))
            for i, (k, v, _keep) in enumerate(store):
                if k == key:
                    store.append(store.pop(i))
                    return v
            v = f(self, *args, **kw)
            store.append((key, v, args))          # holding the arguments keeps their ids from being recycled
            if len(store) > limit:
                store.pop(0)
            return v
        wrapper.__name__ = getattr(f, "__name__", "cached")
        wrapper.__doc__ = getattr(f, "__doc__", None)
        wrapper._cache_store = store
        return wrapper
    return deco


class Cacher(object):
    def __init__(self, operation, limit=3, ignore_args=(), force_kwargs=()):
        self.operation = operation

    def __call__(self, *a, **k):
        return self.operation(*a, **k)
