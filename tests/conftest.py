import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the CPU-side artefacts once per session (host lib + oracle); the HIP lib is built by
    __graft_entry__.build() / `python -m dnascent_amd.build` and must already exist for -m gpu."""
    from dnascent_amd import build
    build.build_host()
    build.build_oracle()
    yield


@pytest.fixture(scope="session")
def model():
    from dnascent_amd import synth
    return synth.pore_model()
