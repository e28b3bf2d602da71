"""mmhand_amd — MI355X-native implementation of the MM-HAND conv-GAN training step.

Hot path only: Generator / Discriminator forward-backward, L1 + perceptual + GAN
losses and Adam, as hand-written HIP kernels behind the C-ABI in include/mmhand_hip.h.
"""
__version__ = "0.1.0"
