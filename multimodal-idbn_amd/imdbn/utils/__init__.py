"""Host-side helpers of the training loops and the device-resident evaluation side-car."""
from .batches import batches
from .energy_utils import class_free_energies, rbm_free_energy
from . import probe_utils

__all__ = ["batches", "rbm_free_energy", "class_free_energies", "probe_utils"]
