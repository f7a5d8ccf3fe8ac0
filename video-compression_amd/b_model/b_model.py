"""``b_model.b_model`` -- same import path as Flex-Rate.../b_model/b_model.py."""
from vcamd.flex import BidirFlowRef  # noqa: F401
