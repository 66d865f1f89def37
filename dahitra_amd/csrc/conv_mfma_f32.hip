// fp32 (parity-mode) instantiations of the MFMA direct convolution (see conv_mfma.hip)
#include "conv_mfma_impl.h"
int dh_conv_launch_f32(const ConvArgs& a, int ks, int stride, hipStream_t st) { return launch_ks<float>(a, ks, stride, st); }
