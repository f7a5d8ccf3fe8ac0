"""``b_model.unet`` -- same import path as Flex-Rate.../b_model/unet.py."""
from vcamd.flex import UNet, UNetConvBlock, UNetUpBlock  # noqa: F401
