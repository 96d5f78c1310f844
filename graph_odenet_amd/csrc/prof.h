#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
// extra = number of additional N x d operand arrays the epilogue reads / writes (pre terms, cotangent terms, Y2)
int  gode_prof_begin(hipStream_t s, int64_t d, int64_t rows, int64_t extra, int kind = 0);   // -1 when profiling is off; kind: GODE_PROF_*
void gode_prof_end(hipStream_t s, int slot);
