"""``src.model.elic`` -- ICIP2024/src/model/elic.py names (forward / rate estimate only)."""
from vcamd.icip2024 import ELIC, ResidualBottleneckBlock  # noqa: F401
