// 5x5 convolutions: the LHBDC mask U-Net (LHBDC/model/layers.py:202-209, stride 1) and the stride-2
// analysis / hyper-analysis layers of the I-frame codec mbt2018_mean (compressai.models.MeanScaleHyperprior,
// used at LHBDC/test/testing.py:209).
#include "conv_mfma.h"
int VC_DISPATCH(k5)(hipStream_t st, const ConvArgs &a, int stride, int cfg, int ck)
{
    if (cfg == VC_CFG_N4) return (stride == 1 && ck == 8) ? launch_conv_n4<5, 5, 1, 8>(st, a) : VC_EINVAL;
    if (stride == 2 && ck == 8) {
        switch (cfg) {
        case VC_CFG_N128: return launch_conv_p<5, 5, 2, 8, CfgN128>(st, a);
        case VC_CFG_N64: return launch_conv_p<5, 5, 2, 8, CfgN64>(st, a);
        case VC_CFG_N32: return launch_conv_p<5, 5, 2, 8, CfgN32>(st, a);
        }
        return VC_EINVAL;
    }
    if (stride != 1 || ck != 16) return VC_EINVAL;
    switch (cfg) {
    case VC_CFG_N128: return launch_conv_p<5, 5, 1, 16, CfgN128>(st, a);
    case VC_CFG_N64: return launch_conv_p<5, 5, 1, 16, CfgN64>(st, a);
    case VC_CFG_N32: return launch_conv_p<5, 5, 1, 16, CfgN32>(st, a);
    case VC_CFG_N16: return launch_conv_p<5, 5, 1, 16, CfgN16>(st, a);
    }
    return VC_EINVAL;
}
