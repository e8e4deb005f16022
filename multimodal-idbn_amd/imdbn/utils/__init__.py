"""Host-side helpers of the training loops."""
from .batches import batches

__all__ = ["batches"]
