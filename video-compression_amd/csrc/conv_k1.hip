// 1x1 convolutions: GDN/IGDN contraction (stride 1) and the stride-2 skip projections.
#include "conv_mfma.h"
int VC_DISPATCH(k1)(hipStream_t st, const ConvArgs &a, int stride, int cfg, int ck)
{
    if (ck != 32) return VC_EINVAL;
    if (stride == 1) {
        switch (cfg) {
        case VC_CFG_N128: return launch_conv_p<1, 1, 1, 32, CfgN128>(st, a);
        case VC_CFG_N64: return launch_conv_p<1, 1, 1, 32, CfgN64>(st, a);
        case VC_CFG_N32: return launch_conv_p<1, 1, 1, 32, CfgN32>(st, a);
        case VC_CFG_N16: return launch_conv_p<1, 1, 1, 32, CfgN16>(st, a);
        }
    } else if (stride == 2) {
        switch (cfg) {
        case VC_CFG_N128: return launch_conv_p<1, 1, 2, 32, CfgN128>(st, a);
        case VC_CFG_N64: return launch_conv_p<1, 1, 2, 32, CfgN64>(st, a);
        case VC_CFG_N32: return launch_conv_p<1, 1, 2, 32, CfgN32>(st, a);
        }
    }
    return VC_EINVAL;
}
