// Issue rate of v_mfma_f32_16x16x4_f32 on gfx950 as the fused MLP kernel uses it: cycles per MFMA per SIMD for 1 / 2 / 4
// wavefronts per SIMD, 1 / 2 / 4 / 8 independent accumulators per wavefront, constant vs changing A / B operands, zero vs
// random data.    hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/micro/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool VARY>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ src, float* __restrict__ out, unsigned long long* clk, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = src[threadIdx.x * 8 + i]; b[i] = src[4096 + threadIdx.x * 8 + i]; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[j % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VARY) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = a[j] * 1.0001f + 1e-6f; b[j] = b[j] * 0.9999f; }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, bool VARY>
void run(const char* name, int waves_per_simd, bool zero) {
  const int threads = 256 * waves_per_simd, grid = 256, iters = 2000;
  float *src, *out;
  unsigned long long* clk;
  hipMalloc(&src, 4096 * 2 * 8 * 4); hipMalloc(&out, grid * threads * 4); hipMalloc(&clk, grid * 16 * 8);
  std::vector<float> h(4096 * 2 * 8);
  for (auto& v : h) v = zero ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, VARY>), dim3(grid), dim3(threads), 0, 0, src, out, clk, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, VARY>), dim3(grid), dim3(threads), 0, 0, src, out, clk, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(grid * 16);
  hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (int g = 0; g < grid; ++g) for (int w = 0; w < 4 * waves_per_simd; ++w) mean += (double)c[g * 16 + w];
  mean /= grid * 4 * waves_per_simd;
  const double mfma_per_simd = (double)iters * 8 * waves_per_simd;
  const double tflops = 2.0 * 16 * 16 * 4 * iters * 8 * (double)grid * 4 * waves_per_simd / (ms * 1e-3) / 1e12;
  printf("%-34s waves/SIMD %d %s: %6.1f ticks per MFMA per SIMD, %7.3f ms, %6.1f TFLOP/s\n", name, waves_per_simd, zero ? "zeros " : "random", mean / mfma_per_simd, ms,
         tflops);
  hipFree(src); hipFree(out); hipFree(clk);
}

int main() {
  for (int zero = 0; zero < 2; ++zero) {
    for (int w : {1, 2, 4}) {
      run<1, false>("1 accumulator, constant operands", w, zero);
      run<2, false>("2 accumulators, constant operands", w, zero);
      run<4, false>("4 accumulators, constant operands", w, zero);
      run<8, false>("8 accumulators, constant operands", w, zero);
      run<2, true>("2 accumulators, changing operands", w, zero);
      run<8, true>("8 accumulators, changing operands", w, zero);
    }
  }
  return 0;
}
