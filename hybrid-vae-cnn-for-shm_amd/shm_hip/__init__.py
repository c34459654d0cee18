"""MI355X-native LSTM-VAE / CNN hot path (HIP kernels behind a C ABI; see include/shm_hip.h)."""
from . import _capi  # noqa: F401  (raises ImportError when libshm_hip.so has not been built)
