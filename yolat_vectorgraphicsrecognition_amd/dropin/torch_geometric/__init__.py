"""Minimal torch_geometric surface used by the YOLaT model file and collate (Data only)."""
from . import data, nn, transforms  # noqa: F401
