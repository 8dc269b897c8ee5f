"""USER-UCG/GPU for MI355X: HIP implementation of the UCG hot path of LAMMPS-UCG.

The directory name carries a hyphen, so it is loaded by path (see ``load_package`` in
tests/conftest.py, bench.py and __graft_entry__.py) under the module name
``lammps_ucg_dev_amd``.
"""
from . import synth  # noqa: F401
from . import capi  # noqa: F401
from . import multi  # noqa: F401
from . import ucgio  # noqa: F401
