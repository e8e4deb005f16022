"""Stimulus datasets served from device memory (reference import path ``imdbn.datasets``, README.md:44)."""
from .uniform_dataset import DeviceLoader, UniformDataset, create_dataloaders_uniform

__all__ = ["UniformDataset", "DeviceLoader", "create_dataloaders_uniform"]
