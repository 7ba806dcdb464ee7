import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "host: host-side code of the library (no GPU needed); runs in the CPU suite and, "
                                       "on a GPU box, is ALSO selected by -m gpu")
    # the extension is required even for the CPU-side tests (symbol / loading checks): build it once
    so = os.path.join(REPO, "yolat_vectorgraphicsrecognition_amd", "libyolat_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "yolat_vectorgraphicsrecognition_amd", "csrc"),
                               "-j8"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        # host-code tests (proposal generation, collate, ABI / checkpoint handling) cost nothing on the GPU box:
        # give them the gpu mark there so that the driver's `-m gpu` pass covers them too (this hook runs before
        # the -m deselection of _pytest.mark)
        for it in items:
            if "host" in it.keywords and "gpu" not in it.keywords:
                it.add_marker(pytest.mark.gpu)
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
