from .base import CompressionModel
from .utils import conv, deconv, update_registered_buffers

__all__ = ["CompressionModel", "conv", "deconv", "update_registered_buffers"]
