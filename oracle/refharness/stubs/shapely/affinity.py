def translate(*a, **k):
    raise NotImplementedError('stub shapely')
