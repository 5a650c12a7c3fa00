"""attentionalpoolingaction_amd -- MI355X-native attentional pooling (hot path only).

csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/apa.h) -> libapa_hip.so
custom_ops/  ctypes loader and python wrappers, shaped like the reference's src/custom_ops
"""
__version__ = '0.1.0'
