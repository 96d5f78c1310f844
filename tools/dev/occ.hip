#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
  for (int lds : {32768, 65536, 69120, 81920, 98304}) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int nb = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds);
    printf("lds=%d blocks/CU=%d (%s)\n", lds, nb, hipGetErrorString(e));
  }
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("sharedMemPerMultiprocessor=%zu sharedMemPerBlock=%zu maxSharedMemoryPerMultiProcessor=%zu regs/CU=%d\n", pr.sharedMemPerMultiprocessor, pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerMultiprocessor);
}
