"""feddlib_amd: MI355X-native FE assembly + Schwarz/GMRES hot path behind a C ABI (include/fedd_hip.h).

The product is feddlib_amd/csrc (HIP kernels + C ABI) and feddlib_amd/host (C++ facade mirroring the
reference's Problem / FE / Domain / BCBuilder surface).  This Python package only loads the shared
library for tests and bench.py."""
from . import capi  # noqa: F401
