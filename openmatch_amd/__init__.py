"""MI355X-native dense-retrieval hot path with the thunlp/OpenMatch v2 Python API.

Layout: `csrc/` hand-written HIP kernels for gfx950 behind a C ABI (include/openmatch_hip.h),
`native.py` its ctypes binding, and the reference-shaped host modules (`arguments`, `modeling`,
`retriever`, `trainer`, `loss`, `utils`, `dataset`).  See DESIGN.md.
"""
__version__ = "0.1.0"
