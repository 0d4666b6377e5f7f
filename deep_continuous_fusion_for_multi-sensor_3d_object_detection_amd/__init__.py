"""MI355X-native implementation of the continuous-fusion train-step hot path.

Module surface mirrors the reference repository (model / loss / train / test /
data_import_carla); compute is HIP kernels in libdcf_hip.so through a C ABI (include/dcf_hip.h).
"""
__version__ = "0.1.0"
