"""Import stub (RAiDER.s1_orbits imports the `s1_orbits` package at module level)."""


def fetch_for_scene(*args, **kwargs):
    raise RuntimeError('s1_orbits stub: no network in the golden-vector harness')
