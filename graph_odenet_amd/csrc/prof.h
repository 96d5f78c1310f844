#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
int  gode_prof_begin(hipStream_t s, int64_t d, int64_t rows);   // -1 when profiling is off
void gode_prof_end(hipStream_t s, int slot);
