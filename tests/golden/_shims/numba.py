"""Inert stand-in for `numba` (absent from this image): jit is the identity."""


def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


njit = jit
