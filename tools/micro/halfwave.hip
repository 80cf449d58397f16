// microbenchmark: does a wave64 with only 32 (or 16) active lanes issue VALU faster on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int active, int iters) {
  if ((int)threadIdx.x >= active) return;
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) { a = fmaf(a, b, c); }  // dependent chain
  }
  out[blockIdx.x * 64 + threadIdx.x] = a + d;
}
__global__ void indep(float* out, int active, int iters) {
  if ((int)threadIdx.x >= active) return;
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f, c = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c); a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c); }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  float* d; hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int kind = 0; kind < 2; ++kind)
    for (int waves_per_cu = 1; waves_per_cu <= 2; ++waves_per_cu)
      for (int active : {64, 32, 16}) {
        int blocks = 256 * waves_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (kind == 0) hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, active, iters);
          else hipLaunchKernelGGL(indep, dim3(blocks), dim3(64), 0, 0, d, active, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ninst = (double)iters * 64;
        printf("%s blocks=%d active=%d: %.3f ms -> %.2f ns/instr (%.2f cycles @2.4GHz)\n", kind ? "indep8" : "chain ", blocks, active, ms, ms * 1e6 / ninst, ms * 1e6 / ninst * 2.4);
      }
  return 0;
}
