// split-bf16 ("bf16x3") instantiations of the MFMA direct convolution: fp32 tensors, three bf16 matrix-core products per
// operand pair (see common.h f32x3, conv_mfma.hip)
#include "conv_mfma_impl.h"
int dh_conv_launch_x3(const ConvArgs& a, int ks, int stride, hipStream_t st) { return launch_ks<f32x3>(a, ks, stride, st); }
