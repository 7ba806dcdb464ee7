"""gcn_lib.sparse surface of the reference (gcn_lib/sparse/__init__.py:1-3) backed by HIP kernels."""
from yolat_vectorgraphicsrecognition_amd.nn_modules import (  # noqa: F401
    MultiSeq, MLP, GraphConv, ResBlock, PlainDynBlock, DenseDynBlock, DilatedKnnGraph,
    AttrRelativeEdgeConvGlobalPool2, act_layer, norm_layer)
