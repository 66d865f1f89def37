// bf16 instantiations of the MFMA direct convolution (see conv_mfma.hip)
#include "conv_mfma_impl.h"
int dh_conv_launch_bf16(const ConvArgs& a, int ks, int stride, hipStream_t st) { return launch_ks<bf16>(a, ks, stride, st); }
