"""multivae_amd — MI355X-native training core for multimodal VAEs (MMVAE / MoPoE / MVTCAE).

Drop-in for the hot path of AgatheSenellart/MultiVae: the `multivae.models` BaseMultiVAE / encoder-decoder
plugin surface and the `multivae.trainers.BaseTrainer` loop are mirrored here (same names, arguments and
error behaviour); the arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of
`include/mvk.h` (libmvk.so).  There is no CPU compute path.
"""
__version__ = "0.1.0"
