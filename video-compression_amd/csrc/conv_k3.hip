// 3x3 convolutions: analysis/synthesis transforms, hyper-networks, U-Nets (stride 1 and 2).
#include "conv_mfma.h"
int VC_DISPATCH(k3)(hipStream_t st, const ConvArgs &a, int stride, int cfg, int ck)
{
    if (cfg == VC_CFG_N4) return (stride == 1 && ck == 16) ? launch_conv_n4<3, 3, 1, 16>(st, a) : VC_EINVAL;
    if (stride == 1 && ck == 32) {
        switch (cfg) {
        case VC_CFG_N128B: return launch_conv_p<3, 3, 1, 32, CfgN128b>(st, a);
        case VC_CFG_N128: return launch_conv_p<3, 3, 1, 32, CfgN128>(st, a);
        case VC_CFG_N64: return launch_conv_p<3, 3, 1, 32, CfgN64>(st, a);
        case VC_CFG_N32: return launch_conv_p<3, 3, 1, 32, CfgN32>(st, a);
        case VC_CFG_N16: return launch_conv_p<3, 3, 1, 32, CfgN16>(st, a);
        }
    } else if (stride == 2 && ck == 8) {
        switch (cfg) {
        case VC_CFG_N128: return launch_conv_p<3, 3, 2, 8, CfgN128>(st, a);
        case VC_CFG_N64: return launch_conv_p<3, 3, 2, 8, CfgN64>(st, a);
        case VC_CFG_N32: return launch_conv_p<3, 3, 2, 8, CfgN32>(st, a);
        }
    }
    return VC_EINVAL;
}
