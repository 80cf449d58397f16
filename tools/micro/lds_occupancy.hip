// How many 64-thread workgroups with X bytes of dynamic LDS does a gfx950 CU keep resident?  Each workgroup spins for
// a fixed number of cycles; 8 workgroups per CU are launched, so the launch takes ceil(8 / resident) spins.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_occ tools/micro/lds_occupancy.hip && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ float sm[];
__global__ __launch_bounds__(64) void spin(long long cycles, float* out) {
  sm[threadIdx.x] = threadIdx.x;
  long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (out && sm[threadIdx.x] < 0) out[0] = 1;
}
int main() {
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int sizes[] = {4096, 16384, 20480, 21504, 24576, 26160, 26624, 27306, 28672, 32768, 36864, 39472, 40960, 49152, 65536, 81920};
  for (int lds : sizes) {
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 64, lds);
    hipLaunchKernelGGL(spin, dim3(2048), dim3(64), lds, 0, 100000LL, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin, dim3(2048), dim3(64), lds, 0, 100000LL, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("lds %6d B: runtime says %2d blocks/CU; 2048 blocks took %.1f us\n", lds, occ, ms * 1e3);
  }
  return 0;
}
