from .discriminator import Discriminator
from .generator import PConvUNet
from .pconv import PConv2d

__all__ = ["PConv2d", "PConvUNet", "Discriminator"]
