'''Import shim for ``appdirs`` (container-only).'''
import tempfile, os
def user_cache_dir(*args, **kwargs):
    return os.path.join(tempfile.gettempdir(), 'nutils_refshim_cache')
