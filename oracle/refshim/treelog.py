'''Import shim for the ``treelog`` logging package (container-only; see
nutils_poly.py in this directory).  Messages are dropped; iterators pass
through; file helpers hand out in-memory buffers.'''
import builtins, contextlib, io, types


def _drop(*args, **kwargs):
    pass


debug = info = user = warning = error = _drop


def withcontext(f):
    return f


@contextlib.contextmanager
def context(title, *initargs, **initkwargs):
    yield _drop


class _Iter:
    def __init__(self, title, *iterables, **kwargs):
        self._it = builtins.zip(*iterables) if len(iterables) != 1 else builtins.iter(iterables[0])

    def __enter__(self):
        return self._it

    def __exit__(self, *exc):
        return False

    def __iter__(self):
        return self._it


@contextlib.contextmanager
def _wrap(titles, iterable):
    yield builtins.iter(iterable)


iter = types.SimpleNamespace(percentage=_Iter, fraction=_Iter, plain=_Iter, wrap=_wrap)


class NullLog:
    def __init__(self, *args, **kwargs):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class StdoutLog(NullLog): pass
class RichOutputLog(NullLog): pass
class LoggingLog(NullLog): pass
class HtmlLog(NullLog):
    filename = 'log.html'
class FilterLog(NullLog): pass
class RecordLog(NullLog):
    def replay(self, log=None):
        pass
class TeeLog(NullLog): pass


@contextlib.contextmanager
def set(log):
    yield log


@contextlib.contextmanager
def add(log):
    yield log


@contextlib.contextmanager
def userfile(name, mode='w', **kwargs):
    yield io.BytesIO() if 'b' in mode else io.StringIO()


infofile = userfile
proto = types.SimpleNamespace(Level=types.SimpleNamespace(debug=0, info=1, user=2, warning=3, error=4))
