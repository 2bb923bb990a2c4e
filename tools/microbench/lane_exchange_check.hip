// Checks the lane-exchange primitives and the sort / merge networks of knn2_kernels.hpp (DPP + v_permlane16/32_swap, gfx950)
// against plain indexing and std::sort.  Build and run on the MI355X:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icorrla_rs_amd/csrc -Iinclude tools/microbench/lane_exchange_check.hip -o /tmp/lxc && /tmp/lxc
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "knn2_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

using namespace corrla::k;

__global__ void xchg_kernel(const int* in, int* out) {  // out[m][lane] for the 11 masks
  const int lane = threadIdx.x;
  const int v = in[lane];
  out[0 * 64 + lane] = k2_xchg<1>(v, lane);
  out[1 * 64 + lane] = k2_xchg<2>(v, lane);
  out[2 * 64 + lane] = k2_xchg<3>(v, lane);
  out[3 * 64 + lane] = k2_xchg<4>(v, lane);
  out[4 * 64 + lane] = k2_xchg<7>(v, lane);
  out[5 * 64 + lane] = k2_xchg<8>(v, lane);
  out[6 * 64 + lane] = k2_xchg<15>(v, lane);
  out[7 * 64 + lane] = k2_xchg<16>(v, lane);
  out[8 * 64 + lane] = k2_xchg<31>(v, lane);
  out[9 * 64 + lane] = k2_xchg<32>(v, lane);
  out[10 * 64 + lane] = k2_xchg<63>(v, lane);
}
__global__ void sort_kernel(const double* d, const int* idx, double* od, int* oi, int trials) {
  const int lane = threadIdx.x;
  for (int t = 0; t < trials; ++t) {
    K2Key a;
    a.d = d[t * 64 + lane];
    a.i = idx[t * 64 + lane];
    k2_sort(a, lane);
    od[t * 64 + lane] = a.d;
    oi[t * 64 + lane] = a.i;
  }
}
// two ascending sequences -> the ascending 128 (what a flush does with the list halves and a sorted batch)
__global__ void merge_kernel(const double* d, const int* idx, double* od, int* oi, int trials) {
  const int lane = threadIdx.x;
  for (int t = 0; t < trials; ++t) {
    K2Key a, b;
    a.d = d[(2 * t) * 64 + lane];
    a.i = idx[(2 * t) * 64 + lane];
    b.d = d[(2 * t + 1) * 64 + lane];
    b.i = idx[(2 * t + 1) * 64 + lane];
    k2_sort(a, lane);
    k2_sort(b, lane);
    const K2Key rb = k2_reverse(b, lane);
    K2Key lo = a, hi = rb;
    if (k2_less(rb, a)) {
      lo = rb;
      hi = a;
    }
    k2_bitonic_merge(lo, lane);
    k2_bitonic_merge(hi, lane);
    od[(2 * t) * 64 + lane] = lo.d;
    oi[(2 * t) * 64 + lane] = lo.i;
    od[(2 * t + 1) * 64 + lane] = hi.d;
    oi[(2 * t + 1) * 64 + lane] = hi.i;
  }
}

int main() {
  const int masks[11] = {1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63};
  int bad = 0;
  {
    std::vector<int> in(64), out(11 * 64);
    for (int i = 0; i < 64; ++i) in[i] = 1000 + 7 * i;
    int *din, *dout;
    CK(hipMalloc(&din, 64 * 4));
    CK(hipMalloc(&dout, 11 * 64 * 4));
    CK(hipMemcpy(din, in.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(xchg_kernel, dim3(1), dim3(64), 0, 0, din, dout);
    CK(hipMemcpy(out.data(), dout, 11 * 64 * 4, hipMemcpyDeviceToHost));
    for (int m = 0; m < 11; ++m)
      for (int l = 0; l < 64; ++l)
        if (out[m * 64 + l] != in[l ^ masks[m]]) {
          if (bad < 10) std::printf("xchg<%d> lane %d: got %d want %d\n", masks[m], l, out[m * 64 + l], in[l ^ masks[m]]);
          ++bad;
        }
  }
  const int trials = 200;
  std::mt19937_64 rng(7);
  std::vector<double> d(trials * 64);
  std::vector<int> idx(trials * 64);
  for (int t = 0; t < trials; ++t)
    for (int l = 0; l < 64; ++l) {
      // many equal distances (ties are ordered by index), some infinities
      const int r = (int)(rng() % 40);
      d[t * 64 + l] = r == 0 ? __builtin_huge_val() : (double)(rng() % (t % 2 ? 8 : 1000000)) * 0.25;
      idx[t * 64 + l] = (int)(rng() % 100000);
    }
  double *dd, *dod;
  int *di, *doi;
  CK(hipMalloc(&dd, trials * 64 * 8));
  CK(hipMalloc(&dod, trials * 64 * 8));
  CK(hipMalloc(&di, trials * 64 * 4));
  CK(hipMalloc(&doi, trials * 64 * 4));
  CK(hipMemcpy(dd, d.data(), trials * 64 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(di, idx.data(), trials * 64 * 4, hipMemcpyHostToDevice));
  std::vector<double> od(trials * 64);
  std::vector<int> oi(trials * 64);
  auto key_less = [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
    return a.first < b.first || (a.first == b.first && a.second < b.second);
  };
  hipLaunchKernelGGL(sort_kernel, dim3(1), dim3(64), 0, 0, dd, di, dod, doi, trials);
  CK(hipMemcpy(od.data(), dod, trials * 64 * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(oi.data(), doi, trials * 64 * 4, hipMemcpyDeviceToHost));
  for (int t = 0; t < trials; ++t) {
    std::vector<std::pair<double, int>> ref(64);
    for (int l = 0; l < 64; ++l) ref[l] = {d[t * 64 + l], idx[t * 64 + l]};
    std::sort(ref.begin(), ref.end(), key_less);
    for (int l = 0; l < 64; ++l)
      if (od[t * 64 + l] != ref[l].first || oi[t * 64 + l] != ref[l].second) {
        if (bad < 10) std::printf("sort trial %d lane %d: got (%g, %d) want (%g, %d)\n", t, l, od[t * 64 + l], oi[t * 64 + l], ref[l].first, ref[l].second);
        ++bad;
      }
  }
  hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(64), 0, 0, dd, di, dod, doi, trials / 2);
  CK(hipMemcpy(od.data(), dod, trials * 64 * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(oi.data(), doi, trials * 64 * 4, hipMemcpyDeviceToHost));
  for (int t = 0; t < trials / 2; ++t) {
    std::vector<std::pair<double, int>> ref(128);
    for (int l = 0; l < 128; ++l) ref[l] = {d[2 * t * 64 + l], idx[2 * t * 64 + l]};
    std::sort(ref.begin(), ref.end(), key_less);
    for (int l = 0; l < 128; ++l)
      if (od[2 * t * 64 + l] != ref[l].first || oi[2 * t * 64 + l] != ref[l].second) {
        if (bad < 10) std::printf("merge trial %d pos %d: got (%g, %d) want (%g, %d)\n", t, l, od[2 * t * 64 + l], oi[2 * t * 64 + l], ref[l].first, ref[l].second);
        ++bad;
      }
  }
  std::printf("lane exchanges, %d sorts, %d merges: %s (%d mismatches)\n", trials, trials / 2, bad ? "FAIL" : "PASS", bad);
  return bad ? 1 : 0;
}
