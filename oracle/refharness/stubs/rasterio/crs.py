class CRS:
    pass
