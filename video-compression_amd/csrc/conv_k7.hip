// 7x7 convolutions of SPyNet's Basic blocks (LHBDC/model/flow.py:52-62).
#include "conv_mfma.h"
int VC_DISPATCH(k7)(hipStream_t st, const ConvArgs &a, int stride, int cfg, int ck)
{
    if (cfg == VC_CFG_N4) return (stride == 1 && ck == 8) ? launch_conv_n4<7, 7, 1, 8>(st, a) : VC_EINVAL;
    if (stride != 1) return VC_EINVAL;
    if (ck == 8 && cfg == VC_CFG_N32) return launch_conv_p<7, 7, 1, 8, CfgN32>(st, a);
    if (ck == 8 && cfg == VC_CFG_N32T16) return launch_conv_p<7, 7, 1, 8, CfgN32T16>(st, a);      // 8 -> 32: one chunk, half the halo per output
    if (ck != 16) return VC_EINVAL;
    switch (cfg) {
    case VC_CFG_N32T16: return launch_conv_p<7, 7, 1, 16, CfgN32T16>(st, a);
    case VC_CFG_N128: return launch_conv_p<7, 7, 1, 16, CfgN128>(st, a);
    case VC_CFG_N64: return launch_conv_p<7, 7, 1, 16, CfgN64>(st, a);
    case VC_CFG_N32: return launch_conv_p<7, 7, 1, 16, CfgN32>(st, a);
    case VC_CFG_N16: return launch_conv_p<7, 7, 1, 16, CfgN16>(st, a);
    }
    return VC_EINVAL;
}
