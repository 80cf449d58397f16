// stream_rows.hip - calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the env-step kernel's access pattern
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"; VERDICT r2 missing #8).
//
// The env-step kernel moves its state as coalesced rows of ONE float per lane (a 256-byte row per wavefront access, the row
// offset a compile-time constant: csrc/env_tables.h "wave-tiled SoA") plus 8-byte terrain pairs.  The guide's x2 correction of
// FETCH_SIZE was measured on 16 B / lane streams.  This program copies a buffer of KNOWN size with 4, 8 and 16 bytes per lane,
// every wavefront walking its own tile row by row exactly like the lane program does; run it under
//     rocprofv3 --pmc FETCH_SIZE -- ./stream_rows      and      rocprofv3 --pmc WRITE_SIZE -- ./stream_rows
// and divide the counter of each kernel by the bytes it is known to read / write (printed below): that ratio is the
// correction factor for this pattern (tools/traffic_calibration.py turns the two passes into profiles/traffic.json entries).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream_rows tools/micro/stream_rows.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

// one wavefront per tile of ROWS rows; a row = 64 lanes x VEC floats; consecutive rows of a tile are contiguous (as in the env tiles)
// (TAG only gives the two buffer sizes different kernel names in the profile)
template <int VEC, int ROWS, int TAG>
__global__ __launch_bounds__(64) void copy_rows(const float* __restrict__ in, float* __restrict__ out) {
  typedef float vec __attribute__((ext_vector_type(VEC)));  // VEC = 1: a plain 4-byte load / store per lane
  const size_t tile = (size_t)blockIdx.x * ROWS * 64;
  const vec* src = reinterpret_cast<const vec*>(in) + tile + threadIdx.x;
  vec* dst = reinterpret_cast<vec*>(out) + tile + threadIdx.x;
  vec r[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) r[i] = src[(size_t)i * 64];  // all loads of the tile in flight, like EnvLane::load()
#pragma unroll
  for (int i = 0; i < ROWS; ++i) dst[(size_t)i * 64] = r[i] * 1.0001f;
}
template <int VEC, int TAG>
void run(const char* name, const float* in, float* out, size_t bytes, int reps) {
  constexpr int ROWS = 32;
  const size_t per_block = (size_t)ROWS * 64 * VEC * 4;
  const int blocks = (int)(bytes / per_block);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((copy_rows<VEC, ROWS, TAG>), dim3(blocks), dim3(64), 0, 0, in, out);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((copy_rows<VEC, ROWS, TAG>), dim3(blocks), dim3(64), 0, 0, in, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double moved = (double)blocks * per_block;
  printf("CAL %s vec_bytes=%d bytes_read=%.0f bytes_written=%.0f launches=%d avg_us=%.2f GBps=%.1f\n", name, VEC * 4, moved, moved, reps + 1, ms * 1e3 / reps,
         2.0 * moved / (ms * 1e-3 / reps) / 1e9);
}

int main(int argc, char** argv) {
  // two sizes: the env kernel's own (14 MB in + out: lives in the 256 MiB Infinity Cache between launches) and one well past it
  const size_t sizes[2] = {(size_t)14 << 20, (size_t)1 << 30};
  for (int s = 0; s < 2; ++s) {
    float *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, sizes[s]) != hipSuccess || hipMalloc(&out, sizes[s]) != hipSuccess) return 1;
    hipMemset(in, 0, sizes[s]);
    hipMemset(out, 0, sizes[s]);
    const int reps = s == 0 ? 50 : 5;
    if (s == 0) {
      run<1, 14>("copy_rows<1, 32, 14>", in, out, sizes[s], reps);
      run<2, 14>("copy_rows<2, 32, 14>", in, out, sizes[s], reps);
      run<4, 14>("copy_rows<4, 32, 14>", in, out, sizes[s], reps);
    } else {
      run<1, 1024>("copy_rows<1, 32, 1024>", in, out, sizes[s], reps);
      run<2, 1024>("copy_rows<2, 32, 1024>", in, out, sizes[s], reps);
      run<4, 1024>("copy_rows<4, 32, 1024>", in, out, sizes[s], reps);
    }
    hipDeviceSynchronize();
    hipFree(in);
    hipFree(out);
  }
  return 0;
}
