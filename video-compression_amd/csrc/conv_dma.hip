// Dispatcher of the fp16-path LDS-DMA convolution pipeline (conv_dma.h, VC_CFG_DMA).
//   3x3 stride 1: analysis / synthesis transforms and residual blocks (LHBDC/model/layers.py:123-166,
//                 ICIP2024/src/model/compression_bottlenecks.py:72-551);  7x7: SPyNet's Basic blocks (LHBDC/model/flow.py:52-62).
#include <stdlib.h>
#include "conv_dma.h"

static bool dma_views_ok(const ConvArgs &a)
{
    // half-precision input in whole 16-byte channel groups; 16-byte epilogue accesses
    return a.in_f16 && a.vec4 && a.vec_out && a.epi == VC_EPI_NONE && a.in_xform == VC_IN_NONE;
}

// the exact fp32 instances (DmaCfg::F32), fp32 tensors either side: SPyNet's two big 7x7 layers (LHBDC/model/flow.py:52-62) and
// the 3x3 stride-1 layers of the residual blocks / U-Nets (LHBDC/model/layers.py:123-166, Flex-Rate.../b_model/unet.py:9-91)
static int conv_dispatch_dma_f32(hipStream_t st, ConvArgs a, int k, int stride)
{
    if (stride != 1 || (k != 7 && k != 3 && k != 5) || a.in_f16 || a.out_f16 || !a.vec4 || !a.vec_out || a.epi != VC_EPI_NONE ||
        a.in_xform != VC_IN_NONE || a.tail_wpk || a.res_f16)
        return VC_EINVAL;
    if ((long long)(k + 15) * a.in_sh * 4 + 48ll * a.in_sw * 4 + 256 >= (1ll << 31)) return VC_EINVAL;
    const int nchunk = a.Cin / 16;
    const int nt = a.Cout >= 128 ? 4 : a.Cout / 32;
    if (a.Cin != nchunk * 16 || a.Cout % 32 || (nt == 4 && a.Cout % 128) || (nt != 1 && nt != 2 && nt != 4)) return VC_EINVAL;
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + 15) / 16;
    a.nblks = a.Cout / (32 * nt);
    if (a.nblks != 1 && k != 3) return VC_EINVAL;
    a.total_blocks = a.tiles_x * a.tiles_y * a.nblks * a.N;
    if (k == 3) {
        if (nchunk == 8 && nt == 4) return launch_conv_dma<DmaCfg<3, 3, 8, 4, 6, 0, false, true>>(st, a);     // 128 -> 128 k
        if (nchunk == 4 && nt == 2) return launch_conv_dma<DmaCfg<3, 3, 4, 2, 3, 0, false, true>>(st, a);     // 64 -> 64
        if (nchunk == 4 && nt == 4) return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 0, false, true>>(st, a);     // 64 -> 128 k
        if (nchunk == 8 && nt == 2) return launch_conv_dma<DmaCfg<3, 3, 8, 2, 3, 0, false, true>>(st, a);     // 128 -> 64
        if (nchunk == 16 && nt == 4) return launch_conv_dma<DmaCfg<3, 3, 16, 4, 6, 0, false, true>>(st, a);   // 256 -> 128 k
        // (32 output channels: a 16-channel chunk of a 3x3 layer is only 2.25 phases of four units -- the next chunk image does not fit
        //  its issue window; those layers stay on the classic instances)
        if (nchunk == 2 && nt == 2) return launch_conv_dma<DmaCfg<3, 3, 2, 2, 3, 0, false, true>>(st, a);     // 32 -> 64
        return VC_EINVAL;
    }
    if (a.out_mode != VC_OUT_PLAIN) return VC_EINVAL;
    if (k == 5) {          // the mask U-Net's 5x5 layers (LHBDC/model/layers.py:202-209)
        if (nchunk == 6 && nt == 1) return launch_conv_dma<DmaCfg<5, 5, 6, 1, 5, 0, false, true>>(st, a);     // 96 -> 32
        if (nchunk == 12 && nt == 2) return launch_conv_dma<DmaCfg<5, 5, 12, 2, 5, 0, false, true>>(st, a);   // 192 -> 64
        if (nchunk == 2 && nt == 2) return launch_conv_dma<DmaCfg<5, 5, 2, 2, 5, 0, false, true>>(st, a);     // 32 -> 64
        return VC_EINVAL;
    }
#ifdef VC_DMA_DIAG      // diagnostic build only (make dma_diag): knock-out variants of the fp32 64 -> 32 instance
    if (nchunk == 4 && nt == 1) {
        const char *e = getenv("VC_DMA_VARIANT");
        switch (e ? atoi(e) : 0) {
        case 1: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 1, false, true>>(st, a);
        case 8: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 8, false, true>>(st, a);
        case 16: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 16, false, true>>(st, a);
        case 32: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 32, false, true>>(st, a);
        case 64: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 64, false, true>>(st, a);
        case 128: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 128, false, true>>(st, a);
        case 25: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 25, false, true>>(st, a);      // MFMAs + barriers only
        case 1024: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 1024, false, true>>(st, a);  // no A DMA
        case 2048: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 2048, false, true>>(st, a);  // no B DMA
        case 256: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 256, false, true>>(st, a);    // DMA issue before the fragment reads
        case 1025: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 1025, false, true>>(st, a);  // no A DMA, no epilogue
        case 17: return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 17, false, true>>(st, a);      // no DMA, no epilogue
        }
    }
#endif
    if (nchunk == 4 && nt == 1) return launch_conv_dma<DmaCfg<7, 7, 4, 1, 5, 0, false, true>>(st, a);     // 64 -> 32
    if (nchunk == 2 && nt == 2) return launch_conv_dma<DmaCfg<7, 7, 2, 2, 5, 0, false, true>>(st, a);     // 32 -> 64
    return VC_EINVAL;
}

int conv_dispatch_dma(hipStream_t st, ConvArgs a, int k, int stride, bool f16)
{
    if (!f16) return conv_dispatch_dma_f32(st, a, k, stride);
    if (stride != 1 || !dma_views_ok(a) || (a.Cin & 31)) return VC_EINVAL;
    const int nchunk = a.Cin / 32;
    // 96 output channels and more: blocks of 128, the last one partly padding (the packing of VC_CFG_N128 / N128B pads the
    // weights and the bias to a multiple of 128 with zeros; the caller names this configuration only on such a packing)
    const int nt = (a.Cout >= 96) ? 4 : (a.Cout == 64 ? 2 : (a.Cout == 32 ? 1 : 0));
    if (!nt || (a.Cout % 4)) return VC_EINVAL;
    if (nt == 4 && (a.Cout % 128) && !a.pack128) return VC_EINVAL;     // a partly padded last block needs the N128 packing
    // per-lane source offsets are 32-bit: the footprint of a tile (KH + 15 rows, COLS <= 40 columns) must stay inside 2 GiB
    if ((long long)(k + 15) * a.in_sh * 2 + 48ll * a.in_sw * 2 + 256 >= (1ll << 31)) return VC_EINVAL;
    a.tiles_x = (a.Wo + 31) / 32;
    a.tiles_y = (a.Ho + 15) / 16;
    a.nblks = (a.Cout + 32 * nt - 1) / (32 * nt);
    if (a.nblks * 32 * nt != a.Cout && a.out_mode != VC_OUT_PLAIN) return VC_EINVAL;
    a.total_blocks = a.tiles_x * a.tiles_y * a.nblks * a.N;
    if (a.tail_wpk) {      // bottleneck block: 3x3 C -> C (activation) -> 1x1 C -> C (+ residual) in one launch, C = 128 or 64
        if (k != 3 || a.Cin != a.Cout || (a.Cout != 128 && a.Cout != 64) || !a.tail_bias || a.chscale || a.out_mode != VC_OUT_PLAIN || a.res_first ||
            (a.act != VC_ACT_NONE && a.act != VC_ACT_RELU && !(a.act == VC_ACT_LRELU && a.slope >= 0.0f && a.slope <= 1.0f)))
            return VC_EINVAL;
        if (a.Cout == 64) return launch_conv_dma<DmaCfg<3, 3, 2, 2, 3, 0, true>>(st, a);
        return launch_conv_dma<DmaCfg<3, 3, 4, 4, 4, 0, true>>(st, a);
    }
    // (a half-precision residual: the fused-tail epilogue above, or the residual-block epilogue of the plain 3x3 instances)
    if (a.res_f16 && (k != 3 || !a.res || a.res_first)) return VC_EINVAL;
    if (k == 3) {
        if (nt == 4 && nchunk == 4) {
#ifdef VC_DMA_DIAG      // diagnostic build only (make dma_diag): knock-out / ring-depth variants of the north-star instance
            const char *e = getenv("VC_DMA_VARIANT");
            const int v = e ? atoi(e) : 0;
            switch (v) {
            case 1: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 1>>(st, a);
            case 2: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 2>>(st, a);
            case 3: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 3>>(st, a);
            case 7: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 7>>(st, a);
            case 11: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 11>>(st, a);
            case 19: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 19>>(st, a);
            case 27: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 27>>(st, a);
            case 32: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 32>>(st, a);
            case 64: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 64>>(st, a);
            case 128: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 128>>(st, a);
            case 129: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 129>>(st, a);
            case 160: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 160>>(st, a);
            case 192: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 192>>(st, a);
            case 256: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 256>>(st, a);
            case 320: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 320>>(st, a);
            case 512: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 512>>(st, a);
            case 768: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 768>>(st, a);
            case 832: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 832>>(st, a);
            case 72: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 72>>(st, a);
            case 68: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 68>>(st, a);
            case 76: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 76>>(st, a);
            case 96: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 96>>(st, a);
            case 65: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 65>>(st, a);
            case 1025: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 1025>>(st, a);
            case 2049: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 2049>>(st, a);
            case 257: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 257>>(st, a);
            case 1281: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 1281>>(st, a);
            case 2305: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 2305>>(st, a);
            case 17: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 17>>(st, a);
            case 4096: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 4096>>(st, a);
            case 8192: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 8192>>(st, a);
            case 16384: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6, 16384>>(st, a);
            case 100: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 9>>(st, a);
            case 101: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 9, 1>>(st, a);
            case 102: return launch_conv_dma<DmaCfg<3, 3, 4, 4, 4>>(st, a);
            }
#endif
            return launch_conv_dma<DmaCfg<3, 3, 4, 4, 6>>(st, a);
        }
        if (nt == 4 && nchunk == 2) return launch_conv_dma<DmaCfg<3, 3, 2, 4, 6>>(st, a);
        if (nt == 4 && nchunk == 3) return launch_conv_dma<DmaCfg<3, 3, 3, 4, 9>>(st, a);      // 96 input channels: 27 phases, ring of 9
        if (nt == 4 && nchunk == 6) return launch_conv_dma<DmaCfg<3, 3, 6, 4, 6>>(st, a);
        if (nt == 4 && nchunk == 8) return launch_conv_dma<DmaCfg<3, 3, 8, 4, 6>>(st, a);
        if (nt == 2 && nchunk == 2) return launch_conv_dma<DmaCfg<3, 3, 2, 2, 3>>(st, a);      // 64 -> 64: 9 phases per tile, ring of 3
    } else if (k == 7) {
        if (nt == 1 && nchunk == 2) return launch_conv_dma<DmaCfg<7, 7, 2, 1, 5>>(st, a);
        if (nt == 2 && nchunk == 1) return launch_conv_dma<DmaCfg<7, 7, 1, 2, 5>>(st, a);
    }
    return VC_EINVAL;
}

#ifdef VC_DMA_DIAG
extern "C" int vc_debug_dma_stamps(unsigned long long *out8)
{
    if (hipDeviceSynchronize() != hipSuccess) return VC_ELAUNCH;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_vc_dma_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return VC_ELAUNCH;
    unsigned long long zero[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_vc_dma_stamps), zero, sizeof(zero)) != hipSuccess) return VC_ELAUNCH;
    return VC_OK;
}
#endif
