#include "common.h"
int launch_attention_f16_v1(const half_t*, half_t*, int, int, int, int, hipStream_t) { return -100; }
