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
  return 0;
}
