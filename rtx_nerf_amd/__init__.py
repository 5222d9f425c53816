"""rtx_nerf_amd -- MI355X-native implementation of the owensgroup/rtx_nerf hot
path (ray/grid traversal, sampler, frequency-encoded fused MLP, volume
rendering) behind the reference's own operator interface.

Layout: csrc/ (HIP kernels + the C ABI of include/rtxn.h -> librtxn.so),
_lib.py (ctypes loader), api.py (host mirror of the reference interface),
render.py (stage orchestration), scenes.py (seeded synthetic inputs).
"""
__version__ = "0.1.0"
