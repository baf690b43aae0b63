// Shared by the convolution translation units (dense.hip, conv16.hip): geometry of an implicit-GEMM convolution over channels-last
// buffers, epilogue selectors, vector types.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGeom {
    // input buffer [B][H][W][C]; window origin of output (ho,wo) is (ho+ih0, wo+iw0)
    int H, W, C, ih0, iw0;
    int Hout, Wout;           // output positions per utterance; M = B*Hout*Wout
    int seglen, segstride;    // KW*C, W*C
    int Ktot;                 // KH*KW*C
    // output buffer [B][OH][OW][OC]; element (ho,wo,n) at (ho+oh0, wo+ow0, n)
    int OH, OW, OC, oh0, ow0;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_LRELU = 2, EPI_MASK_LRELU_GRAD = 3, EPI_BIAS_EXPTANH = 4 };

