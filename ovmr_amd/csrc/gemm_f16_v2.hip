// placeholder until the 256x256 8-phase kernel lands
#include "common.h"
int launch_gemm_f16_v2(const GemmArgs&, hipStream_t) { return -100; }
