"""``b_model.layers`` -- same import path as Flex-Rate.../b_model/layers.py."""
from vcamd.flex import FlowCompressor, Gain_Module, ResidualCompressor  # noqa: F401
