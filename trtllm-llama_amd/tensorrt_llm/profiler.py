"""Wall-clock tag timer (T/tensorrt_llm/profiler.py:4-55): start(tag) / stop(tag) / elapsed_time_in_sec(tag)."""
import time

_timers = {}


class _Timer:

    def __init__(self):
        self.total = 0.0
        self.t0 = None

    def start(self):
        self.t0 = time.time()

    def stop(self):
        if self.t0 is not None:
            self.total += time.time() - self.t0
            self.t0 = None


def start(tag):
    _timers.setdefault(tag, _Timer()).start()


def stop(tag):
    if tag in _timers:
        _timers[tag].stop()


def elapsed_time_in_sec(tag):
    return _timers[tag].total if tag in _timers else None


def reset():
    _timers.clear()


def summary():
    from .logger import logger
    for tag, t in _timers.items():
        logger.info(f'{tag}: {t.total:.6f} sec')
