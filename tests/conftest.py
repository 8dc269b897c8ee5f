"""pytest configuration: markers + loading the hyphen-named package and the oracle binding."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_package():
    """import lammps-ucg-dev_amd/ (the directory name has a hyphen) as lammps_ucg_dev_amd"""
    name = "lammps_ucg_dev_amd"
    if name in sys.modules:
        return sys.modules[name]
    pdir = os.path.join(ROOT, "lammps-ucg-dev_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pdir, "__init__.py"),
                                                  submodule_search_locations=[pdir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    """the CPU oracle binding -- tests only"""
    odir = os.path.join(ROOT, "oracle")
    if odir not in sys.path:
        sys.path.insert(0, odir)
    import orc
    return orc


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def orc():
    return load_oracle()


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    """one HIP context for the whole GPU session"""
    ctx = pkg.capi.Context(-1)
    yield ctx
    ctx.close()


@pytest.fixture
def fresh_ctx(pkg):
    """a context of its own (fix state, step counters and RNG streams start from scratch)"""
    ctx = pkg.capi.Context(-1)
    yield ctx
    ctx.close()

