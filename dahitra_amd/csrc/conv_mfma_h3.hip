// fp16-plane split form ("f32h3": three v_mfma_f32_16x16x32_f16 products per operand pair, ~2^-21; operands inside fp16's range)
// of the MFMA direct convolution: the forward products of the bf16x3 compute mode (see common.h f32h3, conv_mfma.hip)
#include "conv_mfma_impl.h"
int dh_conv_launch_h3(const ConvArgs& a, int ks, int stride, hipStream_t st) { return launch_ks<f32h3>(a, ks, stride, st); }
