"""Inert stand-in for the `ray` package (absent from this image), used ONLY by
tests/golden/make_golden.py inside the build container to import the reference's
Michaelis-Menten scripts unmodified.

It performs no arithmetic: `remote(f)` wraps f so that `.remote(*args)` records
the call, and `get(list)` evaluates the recorded calls (optionally on a fork
pool, one task per particle like the reference's one-Ray-task-per-particle
fan-out, Micmem_likelihood.py:83-87) and returns the results in order.
Every sweep (inputs and outputs) is appended to SWEEPS so the harness can
store it as a golden vector.
"""
import multiprocessing as _mp
import os as _os

SWEEPS = []          # list of (list_of_args, list_of_results) per ray.get call
N_WORKERS = int(_os.environ.get("GOLDEN_WORKERS", "8"))
_TASKS = None        # inherited by forked workers


class _Handle:
    __slots__ = ("f", "args")

    def __init__(self, f, args):
        self.f = f
        self.args = args


class _Remote:
    def __init__(self, f):
        self._f = f

    def remote(self, *args):
        return _Handle(self._f, args)


def remote(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return _Remote(a[0])
    return lambda f: _Remote(f)


def _run_index(i):
    h = _TASKS[i]
    return h.f(*h.args)


def get(handles):
    global _TASKS
    single = isinstance(handles, _Handle)
    hs = [handles] if single else list(handles)
    if N_WORKERS > 1 and len(hs) > 1:
        _TASKS = hs
        ctx = _mp.get_context("fork")
        with ctx.Pool(N_WORKERS) as pool:
            out = pool.map(_run_index, range(len(hs)), chunksize=max(1, len(hs) // (4 * N_WORKERS)))
        _TASKS = None
    else:
        out = [h.f(*h.args) for h in hs]
    SWEEPS.append(([h.args for h in hs], out))
    return out[0] if single else out


def init(*a, **k):
    return None


def shutdown(*a, **k):
    return None


def is_initialized():
    return True
