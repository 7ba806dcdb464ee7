from yolat_vectorgraphicsrecognition_amd.data import Data  # noqa: F401
