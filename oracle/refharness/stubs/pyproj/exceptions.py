"""Stub of pyproj.exceptions (oracle harness only)."""


class ProjError(RuntimeError):
    pass


class CRSError(ProjError):
    pass
