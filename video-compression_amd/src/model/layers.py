"""``src.model.layers`` -- ICIP2024/src/model/layers.py names."""
from vcamd.icip2024 import CheckerboardContext  # noqa: F401
