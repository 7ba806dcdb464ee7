"""`from torch_geometric.nn.data_parallel import DataParallel` (cad_recognition/train.py:17).  The reference only uses
it in a branch that raises NameError (train.py:204-205, undefined SparseDeepGCN); data parallelism here is one process
per GPU (yolat_vectorgraphicsrecognition_amd.trainer), so instantiating this name is an error."""


class DataParallel(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("single-process DataParallel is not part of the MI355X path: launch one process per "
                                  "GPU and use yolat_vectorgraphicsrecognition_amd.Trainer (RCCL gradient all-reduce)")
