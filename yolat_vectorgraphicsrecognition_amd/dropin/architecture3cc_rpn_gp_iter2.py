"""Drop-in for cad_recognition/architecture3cc_rpn_gp_iter2.py (same public names)."""
from yolat_vectorgraphicsrecognition_amd.architecture import (  # noqa: F401
    Backbone, SparseCADGCN, DetectionLoss)
