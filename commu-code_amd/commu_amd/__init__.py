"""commu_amd -- MI355X-native Transformer-XL training + sampling path of ComMU.

Host side mirrors the reference's Python interface (commu.model.model.MemTransformerLM,
train.py, commu.midi_generator.midi_inferrer); the arithmetic runs in hand-written HIP
kernels behind the C ABI of include/commu_hip.h (libcommu_hip.so).  No CPU fallback.
"""
__version__ = "0.1.0"
