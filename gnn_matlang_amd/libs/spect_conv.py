from ..spect_conv import SpectConv, SpectConCatConv, ML3Layer, glorot, zeros  # noqa: F401
