"""``src.utils`` -- the GOP-16 bookkeeping of ICIP2024/src/utils.py:153-250."""
from vcamd.icip2024 import (get_order_typ_list, get_scales, image_compress, select_references,  # noqa: F401
                            update_buffer)
