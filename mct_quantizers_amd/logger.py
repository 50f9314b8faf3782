"""Logging shim with the reference's convention that ``error``/``critical`` log AND raise.

Mirrors the behaviour (not the implementation) of mct_quantizers/logger.py:109-117,163-173:
constructors and wrappers report misuse through ``Logger.error(msg)`` which raises
``Exception(msg)``.
"""
import logging

_LOG = logging.getLogger("mct_quantizers_amd")


class Logger:
    @staticmethod
    def debug(msg: str):
        _LOG.debug(msg)

    @staticmethod
    def info(msg: str):
        _LOG.info(msg)

    @staticmethod
    def warning(msg: str):
        _LOG.warning(msg)

    @staticmethod
    def error(msg: str):
        _LOG.error(msg)
        raise Exception(msg)

    @staticmethod
    def critical(msg: str):
        _LOG.critical(msg)
        raise Exception(msg)
