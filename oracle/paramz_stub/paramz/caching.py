"""Identity caching (the real paramz memoises K/_scaled_dist/dK_dr_via_X, limit=3)."""
def Cache_this(limit=5, ignore_args=(), force_kwargs=()):
    def deco(f):
        return f
    return deco

class Cacher(object):
    def __init__(self, operation, limit=3, ignore_args=(), force_kwargs=()):
        self.operation = operation
    def __call__(self, *a, **k):
        return self.operation(*a, **k)
