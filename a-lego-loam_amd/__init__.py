"""a-lego-loam_amd — MI355X-native hot path of A-LeGO-LOAM (host-side Python mirror).

The product is the C-ABI library `libalego_mi355x.so` (HIP kernels + C++ host, built
from csrc/).  This package only holds thin ctypes bindings used by tests/ and bench.py:
`params` (alego_params mirror), `synth` (synthetic scan generator) and `binding`
(the C ABI of include/alego_mi355x.h).  The directory name contains a hyphen, so it is
loaded through `load_package()` in the repo-root helper `alego_loader.py`.
"""
