"""``model.layers`` -- same import path as LHBDC/model/layers.py."""
from vcamd.lhbdc import Mask, MVCompressor, ResidualCompressor  # noqa: F401
