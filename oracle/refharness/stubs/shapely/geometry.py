class _Box:
    def __init__(self, *bounds):
        self.bounds = bounds


def box(*bounds):
    return _Box(*bounds)


Polygon = _Box


class Point:
    def __init__(self, *a, **k):
        self.coords = a
