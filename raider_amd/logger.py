"""Logging for the shim: a plain stdlib logger named like the reference's (tools/RAiDER/logger.py) but
without its side effect of creating debug.log / error.log in the working directory."""
import logging

logger = logging.getLogger('RAiDER')
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter('%(levelname)s: %(message)s'))
    logger.addHandler(_h)
    logger.setLevel(logging.WARNING)
