from e4s_amd.stylegan2 import (Blur, ConstantInput, ConvLayer, Discriminator, Downsample, EqualConv2d,  # noqa: F401
                               EqualLinear, Generator, ModulatedConv2d, NoiseInjection, PixelNorm, ResBlock,
                               ScaledLeakyReLU, StyledConv, ToRGB, Upsample, make_kernel)
