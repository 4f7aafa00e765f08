"""Import shim: ``bodyct-dram-emph-subtype_amd/`` (the package directory mandated by the
project layout) is not a valid Python identifier, so this module turns itself into that
package: ``import bodyct_dram_emph_subtype_amd as dram``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "bodyct-dram-emph-subtype_amd")]
__package__ = __name__
if globals().get("__spec__") is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
