"""Import alias for the on-disk package directory ``wsi-hgnn_amd/``.

The product package lives in ``wsi-hgnn_amd/`` (the name the project layout
prescribes), which is not a valid Python identifier.  This shim makes
``import wsi_hgnn_amd`` resolve to that directory: it points ``__path__`` at it
and executes its ``__init__.py`` in this module's namespace, so
``wsi_hgnn_amd.models``, ``wsi_hgnn_amd.pooling`` … are the files under
``wsi-hgnn_amd/``.  No code lives here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "wsi-hgnn_amd")
if not _os.path.isdir(_real):  # pragma: no cover
    raise ImportError("package directory %r is missing" % _real)
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
