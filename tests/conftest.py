import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, in-tree
    needed = [os.path.join(ROOT, "cuda-raytracing-optimized_amd", "librt_mi355x.so"),
              os.path.join(ROOT, "cuda-raytracing-optimized_amd", "librt_host.so"),
              os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in needed):
        import subprocess
        subprocess.run(["make", "-C", ROOT, "-j4", "all"], check=True)


@pytest.fixture(scope="session")
def rt():
    import cuda_raytracing_optimized_amd as rt
    return rt


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    return oracle
