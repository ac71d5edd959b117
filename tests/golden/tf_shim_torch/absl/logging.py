"""The handful of absl.logging names the reference's util/logging.py uses."""
import logging as _logging

INFO = _logging.INFO
_log = _logging.getLogger('absl-shim')


def set_verbosity(level):
    _log.setLevel(level)


def info(msg, *args):
    _log.info(msg, *args)


def warning(msg, *args):
    _log.warning(msg, *args)


def error(msg, *args):
    _log.error(msg, *args)


def debug(msg, *args):
    _log.debug(msg, *args)
