// Dispatcher of the fp16-path LDS-DMA convolution pipeline (conv_dma.h, VC_CFG_DMA).
//   3x3 stride 1: analysis / synthesis transforms and residual blocks (LHBDC/model/layers.py:123-166,
//                 ICIP2024/src/model/compression_bottlenecks.py:72-551);  7x7: SPyNet's Basic blocks (LHBDC/model/flow.py:52-62).
#include "conv_dma.h"

static bool dma_views_ok(const ConvArgs &a)
{
    // half-precision input in whole 16-byte channel groups; 16-byte epilogue accesses
    return a.in_f16 && a.vec4 && a.vec_out && a.epi == VC_EPI_NONE && a.in_xform == VC_IN_NONE;
}

int conv_dispatch_dma(hipStream_t st, ConvArgs a, int k, int stride)
{
    if (stride != 1 || !dma_views_ok(a) || (a.Cin & 31)) return VC_EINVAL;
    const int nchunk = a.Cin / 32;
    const int nt = (a.Cout % 128 == 0) ? 4 : (a.Cout == 64 ? 2 : (a.Cout == 32 ? 1 : 0));
    if (!nt) return VC_EINVAL;
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + 15) / 16;
    a.nblks = a.Cout / (32 * nt);
    a.total_blocks = a.tiles_x * a.tiles_y * a.nblks * a.N;
    if (k == 3) {
        if (nt == 4 && nchunk == 4) return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6>>(st, a);
        if (nt == 4 && nchunk == 2) return launch_conv_dma<DmaCfg<3, 3, 2, 4, 6>>(st, a);
        if (nt == 4 && nchunk == 6) return launch_conv_dma<DmaCfg<3, 3, 6, 4, 6>>(st, a);
        if (nt == 4 && nchunk == 8) return launch_conv_dma<DmaCfg<3, 3, 8, 4, 6>>(st, a);
    } else if (k == 7) {
        if (nt == 1 && nchunk == 2) return launch_conv_dma<DmaCfg<7, 7, 2, 1, 5>>(st, a);
        if (nt == 2 && nchunk == 1) return launch_conv_dma<DmaCfg<7, 7, 1, 2, 5>>(st, a);
    }
    return VC_EINVAL;
}
