'''Import shim for ``stringly`` (CLI docstring parsing only; container-only).'''
import types
util = types.SimpleNamespace(DocString=None)
