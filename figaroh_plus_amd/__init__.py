"""MI355X-native regressor + base-parameter QR + LS path behind FIGAROH's function API.

Only the hot path named in BASELINE.json lives here:
``tools.regressor`` / ``tools.qrdecomposition`` / ``tools.robot`` /
``identification.identification_tools`` mirror the reference modules of the
same names under ``src/figaroh/``; every numeric routine in them calls the
hand-written HIP kernels in ``csrc/`` through the ctypes C-ABI declared in
``include/figh.h``.  There is no CPU fallback: importing ``_lib`` without a
built ``libfigh.so`` raises.
"""
__version__ = "0.1.0"
