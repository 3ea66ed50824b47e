// How many 256-thread workgroups of a given static LDS size run on one CU at the same time (gfx950)?  Every workgroup spins for a
// fixed number of cycles; 512 workgroups on 256 CUs finish in ONE spin time if two fit a CU and in two if only one does.
//   hipcc --offload-arch=gfx950 -O3 -o lds_occupancy lds_occupancy.hip && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BYTES, int VG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VG, VG))) void k(unsigned long long spin, unsigned* out) {
  __shared__ unsigned char lds[BYTES];
  lds[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = lds[(blockIdx.x * 7) & 255];
}
template <int BYTES, int VG> void run(const char* tag, unsigned* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned long long spin = 2000000ull;                 // 100 MHz counter? print both
  for (int n : {256, 512, 1024}) {
    k<BYTES, VG><<<n, 256>>>(1000, out); hipDeviceSynchronize();
    hipEventRecord(a); k<BYTES, VG><<<n, 256>>>(spin, out); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s LDS %6d B, %4d workgroups: %.3f ms\n", tag, BYTES, n, ms);
  }
}
int main() {
  unsigned* out; hipMalloc(&out, 4096 * 4);
  run<32768, 2>("32 KB", out);
  run<65536, 2>("64 KB", out);
  run<73728, 2>("72 KB", out);
  run<79048, 2>("79 048 B (group kernel)", out);
  run<81920, 2>("80 KB", out);
  run<81921, 2>("80 KB + 1", out);
  return 0;
}
