"""Logger singleton (T/tensorrt_llm/logger.py:30-53): level via set_level() or TLLM_LOG_LEVEL."""
import logging
import os


class Logger:
    _LEVELS = {'internal_error': logging.CRITICAL, 'error': logging.ERROR, 'warning': logging.WARNING,
               'info': logging.INFO, 'verbose': logging.DEBUG}

    def __init__(self):
        self._logger = logging.getLogger('TRT-LLM')
        if not self._logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter('[%(asctime)s] [TRT-LLM] [%(levelname).1s] %(message)s', '%m/%d/%Y-%H:%M:%S'))
            self._logger.addHandler(h)
            self._logger.propagate = False
        self.set_level(os.environ.get('TLLM_LOG_LEVEL', 'warning').lower())

    def set_level(self, level: str):
        self._level = level
        self._logger.setLevel(self._LEVELS.get(level, logging.WARNING))

    @property
    def level(self):
        return self._level

    def error(self, msg):
        self._logger.error(msg)

    def warning(self, msg):
        self._logger.warning(msg)

    def info(self, msg):
        self._logger.info(msg)

    def verbose(self, msg):
        self._logger.debug(msg)

    debug = verbose


logger = Logger()
