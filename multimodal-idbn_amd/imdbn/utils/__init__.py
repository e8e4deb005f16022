"""Host-side helpers of the training loops."""
from .batches import batches
from .energy_utils import class_free_energies, rbm_free_energy

__all__ = ["batches", "rbm_free_energy", "class_free_energies"]
