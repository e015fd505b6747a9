class Implicit_Problem:
    def __init__(self, *a, **k):
        raise RuntimeError("assimulo is absent from this image; the DAE path cannot run")
