class _Absent:
    def __init__(self, *a, **k):
        raise RuntimeError("assimulo is absent from this image; the DAE path cannot run")


class IDA(_Absent):
    pass


class Radau5DAE(_Absent):
    pass
