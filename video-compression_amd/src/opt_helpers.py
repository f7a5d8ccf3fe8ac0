"""``src.opt_helpers`` -- ICIP2024/src/opt_helpers.py:23-51 (flow-resolution search) on the HIP path."""
from vcamd.icip2024 import get_best_down_ratio_prediction, prediction_flowonly  # noqa: F401
