// Does a back-to-back DEPENDENT v_mfma (same accumulator as SrcC and vDst) cost issue cycles on gfx950?
// The GEMM kernels issue the VEC = 4 k-steps of one fragment into the same accumulator consecutively (chains of 4, or of
// 2 in f64).  Pattern CH = chain length; 18 accumulators per wave, one wave per SIMD, register-only.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_chain.hip -o gpurun_out/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <int CH, class T, class V>
__global__ __launch_bounds__(256) void mfma_chain(T* out, int iters, T seed) {
  constexpr int NACC = 18;
  V acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (V){0, 0, 0, 0};
  T a = seed + threadIdx.x * (T)1e-3, b = seed * (T)0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4 / CH; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if constexpr (sizeof(T) == 4)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          else
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
  }
  V s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// Cross-checks of the f64 ceiling (round 3): the same register-only stream with TWO waves per SIMD (512-thread blocks), and
// the other f64 MFMA shape, v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks per instruction: 512 flop, one f64 result per lane).
template <class T, class V>
__global__ __launch_bounds__(512) void mfma_two_waves(T* out, int iters, T seed) {
  constexpr int NACC = 18;
  V acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (V){0, 0, 0, 0};
  T a = seed + threadIdx.x * (T)1e-3, b = seed * (T)0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (sizeof(T) == 4)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        else
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
  }
  V s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int THREADS>
__global__ __launch_bounds__(THREADS) void mfma_f64_4x4x4(double* out, int iters, double seed) {
  constexpr int NACC = 36;
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
void time_launch(const char* label, double flops, F&& launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 12; ++rep) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 6 && ms < best) best = ms;
  }
  printf("%s: %.3f ms  %.1f TFLOP/s\n", label, best, flops / best / 1e9);
}
template <int CH, class T, class V>
void run(const char* name, int cus, void* out) {
  const int iters = 5000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 12; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_chain<CH, T, V>), dim3(cus), dim3(256), 0, 0, (T*)out, iters, (T)1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 6 && ms < best) best = ms;
  }
  const double flops = 2.0 * 16 * 16 * 4 * 18.0 * 4 * iters * 4 * cus;
  printf("%s chain=%d: %.3f ms  %.1f TFLOP/s\n", name, CH, best, flops / best / 1e9);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  void* out;
  hipMalloc(&out, (size_t)cus * 256 * 8);
  run<1, float, f4>("f32 16x16x4", cus, out);
  run<2, float, f4>("f32 16x16x4", cus, out);
  run<4, float, f4>("f32 16x16x4", cus, out);
  run<1, double, d4>("f64 16x16x4", cus, out);
  run<2, double, d4>("f64 16x16x4", cus, out);
  run<4, double, d4>("f64 16x16x4", cus, out);
  const int iters = 5000;
  hipFree(out);
  hipMalloc(&out, (size_t)cus * 512 * 8);
  time_launch("f64 16x16x4, two waves per SIMD", 2.0 * 16 * 16 * 4 * 18.0 * 4 * iters * 8 * cus, [&] {
    hipLaunchKernelGGL((mfma_two_waves<double, d4>), dim3(cus), dim3(512), 0, 0, (double*)out, iters, 1.0);
  });
  time_launch("f32 16x16x4, two waves per SIMD", 2.0 * 16 * 16 * 4 * 18.0 * 4 * iters * 8 * cus, [&] {
    hipLaunchKernelGGL((mfma_two_waves<float, f4>), dim3(cus), dim3(512), 0, 0, (float*)out, iters, 1.0f);
  });
  time_launch("f64 4x4x4 (4 blocks), one wave per SIMD", 2.0 * 4 * 4 * 4 * 4 * 36.0 * 4 * iters * 4 * cus, [&] {
    hipLaunchKernelGGL((mfma_f64_4x4x4<256>), dim3(cus), dim3(256), 0, 0, (double*)out, iters, 1.0);
  });
  time_launch("f64 4x4x4 (4 blocks), two waves per SIMD", 2.0 * 4 * 4 * 4 * 4 * 36.0 * 4 * iters * 8 * cus, [&] {
    hipLaunchKernelGGL((mfma_f64_4x4x4<512>), dim3(cus), dim3(512), 0, 0, (double*)out, iters, 1.0);
  });
  return 0;
}
