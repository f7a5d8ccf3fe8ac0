"""``src.model.elic`` -- the block ICIP2024's B-frame codec takes from elic.py (the ELIC intra codec itself is out of scope)."""
from vcamd.icip2024 import ResidualBottleneckBlock  # noqa: F401
