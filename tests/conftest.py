import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

# A scene builds its escape masks when it is about to render its third frame (a one-shot render is better off without them,
# csrc/pt_gpu.hip scene_upload); the tests render one or two frames per scene and mean to test the masked pipeline: from frame one.
import os  # noqa: E402
os.environ.setdefault("PT_ESCAPE_AFTER", "0")

GOLDEN = ROOT / "tests" / "golden"
SCENES = GOLDEN / "scenes"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first():
    """On the GPU box torch must create its HIP context before libptgpu.so does, otherwise torch reports
    'No HIP GPUs are available' in the same process (bench.py does the same)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass


@pytest.fixture(scope="session")
def pta():
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def scene_cache(pta):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pta.HostScene.load_isf(SCENES / name / "scene.isf")
        return cache[name]
    return get


@pytest.fixture(scope="session")
def gpu_scene_cache(pta, scene_cache):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = pta.GpuScene(scene_cache(name), device=0)
        return cache[name]
    return get
