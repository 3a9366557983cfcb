class Affine:
    pass
