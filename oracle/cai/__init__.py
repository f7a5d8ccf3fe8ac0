"""oracle.cai -- restatement of the CompressAI 1.1.8 surface used by the reference hot path.
TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at this third-party boundary (see oracle/__init__.py)."""
from . import ans, entropy_models, layers, models  # noqa: F401
