"""Host-side helpers of the training loops and the device-resident evaluation side-car."""
from .batches import batches, rows_on_device
from .energy_utils import class_free_energies, rbm_free_energy
from . import probe_utils

__all__ = ["batches", "rows_on_device", "rbm_free_energy", "class_free_energies", "probe_utils"]
