#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* o){ extern __shared__ float s[]; s[threadIdx.x]=1; __syncthreads(); o[0]=s[0]; }
int main(){
  hipDeviceProp_t p; hipGetDeviceProperties(&p,0);
  printf("name %s smem/block %zu optin %zu perMP %zu CUs %d clock %d l2 %d\n", p.name, p.sharedMemPerBlock, p.sharedMemPerBlockOptin, p.maxSharedMemoryPerMultiProcessor, p.multiProcessorCount, p.clockRate, p.l2CacheSize);
  int sizes[] = {65536, 98304, 131072, 160*1024, 163840};
  float* d; hipMalloc(&d, 4);
  for (int s : sizes) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, s);
    printf("set %d -> %s\n", s, hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), s, 0, d);
    e = hipGetLastError(); printf(" launch %d -> %s\n", s, hipGetErrorString(e));
    e = hipDeviceSynchronize(); printf(" sync -> %s\n", hipGetErrorString(e));
  }
  return 0;
}
