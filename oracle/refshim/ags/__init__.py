'''Import shim for ``ags`` (string (de)serialisation used only by the reference
CLI and env-variable defaults; container-only).'''
class _Codec:
    @staticmethod
    def loads(s, T):
        if T is bool:
            return s.lower() in ('1', 'true', 'yes', 'on')
        return T(s)
    @staticmethod
    def dumps(v, T=None):
        return str(v)
yaml = ucsl = _Codec
def load(*args, **kwargs):
    raise NotImplementedError('ags.load is not available in the import shim')
