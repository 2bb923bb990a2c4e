// Register-only MFMA ceiling on gfx950: every wave issues independent v_mfma_f32_16x16x4_f32 (or f64) back to back
// from 18 accumulators (the GEMM kernels' MW x NT = 2 x 9), no LDS, no memory.  One workgroup of `waves` waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32(float* out, int iters, float seed) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
  float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma_f64(double* out, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  d4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 4;
  const int iters = 20000, nacc = 18;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  void* out;
  hipMalloc(&out, (size_t)cus * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int dt = 0; dt < 2; ++dt) {
    for (int rep = 0; rep < 12; ++rep) {
      hipEventRecord(e0);
      if (dt == 0)
        hipLaunchKernelGGL(mfma_f32<18>, dim3(cus), dim3(64 * waves), 0, 0, (float*)out, iters, 1.0f);
      else
        hipLaunchKernelGGL(mfma_f64<18>, dim3(cus), dim3(64 * waves), 0, 0, (double*)out, iters, 1.0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = 2.0 * 16 * 16 * 4 * (double)nacc * iters * waves * cus;
      if (rep == 0 || rep == 5 || rep == 11)
        printf("%s waves/CU=%d rep %2d: %.3f ms  %.1f TFLOP/s\n", dt == 0 ? "f32 16x16x4" : "f64 16x16x4", waves, rep, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
