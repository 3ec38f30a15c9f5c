"""default network context (T/tensorrt_llm/_common.py): `with net_guard(network):` makes `default_net()` return it,
and every functional op appends a node to it."""
import contextlib
import threading

_tls = threading.local()


def default_net():
    net = getattr(_tls, 'net', None)
    assert net is not None, 'Use builder to create network first, and use `with net_guard(network)` to activate it'
    return net


def has_default_net():
    return getattr(_tls, 'net', None) is not None


def default_trtnet():
    """The reference returns the underlying trt.INetworkDefinition; here the Network records nodes itself."""
    return default_net()


def set_network(network):
    _tls.net = network


@contextlib.contextmanager
def net_guard(network):
    assert network is not None
    old = getattr(_tls, 'net', None)
    _tls.net = network
    try:
        yield
    finally:
        _tls.net = old


@contextlib.contextmanager
def precision(dtype):
    """`with precision("float32")` of the reference pins TensorRT layer precision; kernels here always keep
    fp32 statistics/accumulators, so this is a no-op kept for source compatibility."""
    yield


def _is_building(f):
    """Decorator of Builder.build_engine: sets IS_BUILDING=1 so that the collective plugins no-op at build time
    (T/tensorrt_llm/builder.py:13-32, P/common/plugin.h:145-157)."""
    import functools
    import os

    @functools.wraps(f)
    def wrapper(*args, **kwargs):
        os.environ['IS_BUILDING'] = '1'
        try:
            return f(*args, **kwargs)
        finally:
            os.environ['IS_BUILDING'] = '0'

    return wrapper
