"""``src.model.compression_bottlenecks`` -- ICIP2024/src/model/compression_bottlenecks.py names."""
from vcamd.icip2024 import Offset_ELIC, Res_ELIC  # noqa: F401
