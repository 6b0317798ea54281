from .gdn import GDN, GDN1
from .layers import (AttentionBlock, MaskedConv2d, ResidualBlock, ResidualBlockUpsample, ResidualBlockWithStride,
                     conv3x3, subpel_conv3x3)

__all__ = ["GDN", "GDN1", "AttentionBlock", "MaskedConv2d", "ResidualBlock", "ResidualBlockUpsample",
           "ResidualBlockWithStride", "conv3x3", "subpel_conv3x3"]
