"""``src.model.helpers`` -- ICIP2024/src/model/helpers.py names."""
from vcamd.icip2024 import (FlowNET, MS_Feature, OffsetDiversity, OffsetTemproalEnc, Reconstuctor,  # noqa: F401
                            ResidualTemproalEnc, conv, deconv)
