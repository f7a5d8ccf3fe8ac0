// Diagnostic build only (`make stamps` -> ../libvc_hip_stamps.so, never shipped): the fp32 3x3 and 7x7 convolution
// instances compiled with -DVC_STAMPS, i.e. with s_memtime stamps around the phases of conv_mfma_kernel (barrier,
// staging, barrier, contraction loop, epilogue) accumulated per wave into g_vc_stamps; tools/stamps.py reads them with
// vc_debug_read_stamps.  One translation unit so that the device-side counter array exists exactly once; every
// other layer comes from the ordinary objects.
// -DVC_STAMPS_F16 (`make stamps16` -> ../libvc_hip_stamps16.so) instruments the fp16-path instances instead.
#ifdef VC_STAMPS_F16
#define VC_TU_F16 1
#else
#define VC_TU_F16 0
#endif
#include "conv_k3.hip"
#include "conv_k7.hip"
#include "conv_api.hip"
