"""devo_amd — MI355X-native (gfx950) implementation of DEVO's recurrent update + bundle-adjustment hot path:
altcorr (patch correlation lookup), fastba (sparse Gauss-Newton BA) and the lietorch SE3 ops, as hand-written
HIP kernels behind a C ABI (include/devo_hip.h, devo_amd/lib/libdevo_hip.so) and the reference's own Python
extension-module interfaces (devo_amd.backends.{cuda_corr, cuda_ba, lietorch_backends}).

Importing this package does not load the HIP library; the first call into a backend does, and fails loudly
if it is missing.  There is no CPU fallback anywhere in the package.
"""
__version__ = "0.1.0"
