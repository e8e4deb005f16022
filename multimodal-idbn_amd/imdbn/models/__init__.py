"""Model classes with the reference's import surface (reference ``imdbn/models/__init__.py:5-35``)."""
import sys
from types import ModuleType

from .rbm import RBM
from .idbn import iDBN
from .imdbn import iMDBN
from .imdbn_bimodal import iMDBN_BiModal
from . import gdbn_model_complete  # noqa: F401  (monolith path used by reference-written pickles)

__all__ = ["RBM", "iDBN", "iMDBN", "iMDBN_BiModal"]

# Legacy Groundeep pickles name their classes src.classes.{rbm_model,dbn_model,gdbn_model}.*
_this = sys.modules[__name__]
_src = sys.modules.get("src") or ModuleType("src")
_cls = sys.modules.get("src.classes") or ModuleType("src.classes")
_cls.rbm_model = _cls.dbn_model = _cls.gdbn_model = _this
_src.classes = _cls
sys.modules.setdefault("src", _src)
sys.modules.setdefault("src.classes", _cls)
for _n in ("rbm_model", "dbn_model", "gdbn_model"):
    sys.modules.setdefault(f"src.classes.{_n}", _this)
