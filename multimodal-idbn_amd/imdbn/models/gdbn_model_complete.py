"""Monolith module path of the reference (``imdbn/models/gdbn_model_complete.py``).

``from imdbn.models import RBM, iDBN, iMDBN`` in the reference resolves to classes defined in a
module of this name, so reference-written pickles carry ``imdbn.models.gdbn_model_complete.RBM``
etc..  The arithmetic there is identical to the extraction modules; here both
paths are the same engine-backed classes.
"""
from .rbm import RBM
from .idbn import iDBN
from .imdbn import iMDBN

__all__ = ["RBM", "iDBN", "iMDBN"]
