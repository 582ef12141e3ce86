"""Shared fixtures.  GPU tests are marked `gpu`; everything else runs on a CPU-only host."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are kept out of the history): build them once,
    in-tree, when a compiler is there.  Nothing is rebuilt when the files exist."""
    import shutil
    need = [os.path.join(ROOT, "path-tracing_amd", "libpt_hip.so"), os.path.join(ROOT, "path-tracing_amd", "pt_render"),
            os.path.join(ROOT, "path-tracing_amd", "pt_prepare"), os.path.join(ROOT, "oracle", "libpt_oracle.so"),
            os.path.join(ROOT, "oracle", "libpt_oracle_shared.so")]
    if all(os.path.exists(p) for p in need):
        return
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        entry.build()


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def scenes(pkg):
    from path_tracing_amd import scenes as s
    return s


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle with libm transcendentals (the default restatement)."""
    return entry.load_oracle_module().Oracle(shared_math=False)


@pytest.fixture(scope="session")
def oracle_shared():
    """CPU oracle built with the portable polynomial math of the kernels' exact-math mode."""
    return entry.load_oracle_module().Oracle(shared_math=True)


@pytest.fixture(scope="session")
def orc_mod():
    return entry.load_oracle_module()


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
