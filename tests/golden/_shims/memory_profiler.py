"""Inert stand-in for `memory_profiler` (absent from this image)."""


def profile(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f
