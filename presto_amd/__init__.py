"""presto_amd: MI355X-native acceleration search (the accelsearch hot path).

The compute lives in libaccelgpu.so (hand-written HIP for gfx950 behind a C
ABI, see include/accelgpu.h); this package is the thin Python host layer used
by the tests, the benchmark and the batch driver.
"""
from .api import AccelSearchGPU, AccelGPUError, HIT_DTYPE, CAND_DTYPE, plan, load_library  # noqa: F401
