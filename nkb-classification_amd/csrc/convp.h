// Row-balanced DMA-pipelined 3x3 convolution core (convp.hip); entry points are declared in include/nkbhip.h.
#pragma once
#include <hip/hip_runtime.h>
extern "C" int nkb_convp_form_enabled(int form);      // 4: conv1p.hip, 5: stemp.hip, 6: gramr.hip (NKB_CONVP / nkb_convp_config)
extern "C" int nkb_rowres_reserved_cus();             // CUs the family's backward kernels leave to a co-resident collective
extern "C" int nkb_convp_tiles(int dtype, int kind, int N, int H, int W, int Cin, int ldx, int Cout, int ldy, int R, int S, int stride,
                               int pad);
