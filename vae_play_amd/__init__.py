"""vae_play_amd -- MI355X (gfx950) native back end for the convolutional-VAE training step of
kungyao/vae-play.  Hot path only: drop-in Encoder/Decoder/reparameterize,
HIP kernels behind a C ABI (include/vaeplay_hip.h), flat-arena optimiser, data-parallel step.
Importing the package does not touch the GPU; using any op without libvaeplay_hip.so raises.
"""
from . import _lib  # noqa: F401
from .networks import (VAE, Decoder, DecoderBlock, DirectDecoder, Discriminator, Encoder, EncoderBlock,  # noqa: F401
                       VaeGan, init_parameters, reparameterize)
from .functional import (binary_cross_entropy, get_conv_precision, kl_divergence, set_conv_precision,  # noqa: F401
                         vae_loss)

__all__ = ["VAE", "VaeGan", "Encoder", "Decoder", "Discriminator", "DirectDecoder", "EncoderBlock", "DecoderBlock",
           "reparameterize", "init_parameters",
           "binary_cross_entropy", "kl_divergence", "vae_loss", "set_conv_precision", "get_conv_precision"]
