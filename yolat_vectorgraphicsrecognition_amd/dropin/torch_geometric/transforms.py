"""`import torch_geometric.transforms as T` (cad_recognition/train.py:16) — imported by the reference, never used on
the YOLaT path; an empty namespace so the import resolves."""
