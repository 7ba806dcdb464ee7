"""torch_scatter.scatter for the two calls on the YOLaT hot path (sorted index; mean / max)."""
from yolat_vectorgraphicsrecognition_amd.nn_modules import scatter  # noqa: F401
