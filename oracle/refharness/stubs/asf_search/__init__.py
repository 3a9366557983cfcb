"""Import stub for the golden-vector harness (no network here): RAiDER.s1_azimuth_timing imports asf_search at module
level; only its pure date/weight functions are exercised."""


class PRODUCT_TYPE:
    SLC = 'SLC'


def geo_search(*args, **kwargs):
    raise RuntimeError('asf_search stub: no network in the golden-vector harness')
