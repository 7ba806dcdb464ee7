"""torch_geometric.nn names the reference imports (train.py:17); see data_parallel.py."""
from . import data_parallel  # noqa: F401
