"""``model.flow`` -- same import path as LHBDC/model/flow.py."""
from vcamd import hip as _hip
from vcamd.lhbdc import Network  # noqa: F401


def backwarp(tenInput, tenFlow):
    """LHBDC/model/flow.py:15-25 (convention W1) on the device."""
    return _hip.nhwc_to_nchw(_hip.warp(_hip.WARP_W1, _hip.nchw_to_nhwc(tenInput), _hip.nchw_to_nhwc(tenFlow)))
