// three-way split-bf16 ("f32x6": six bf16 matrix-core products per operand pair, unit roundoff ~2^-23) instantiations of the
// MFMA direct convolution: the forward products of the bf16x3 compute mode (see common.h f32x6, conv_mfma.hip)
#include "conv_mfma_impl.h"
int dh_conv_launch_x6(const ConvArgs& a, int ks, int stride, hipStream_t st) { return launch_ks<f32x6>(a, ks, stride, st); }
