// The m-sized products of the thin-Q stage on very tall sketches (random_svd.rs:38,57 qr().compute_thin_q(), :103 U = Q U~)
// for l <= 96, f32:
//
//     tall_apply_kernel   out = Y M        (m x l) (l x l2)   -- Y R^-1 of Cholesky-QR, U = Q U~
//     tall_gram_kernel    G   = Y^T Y      (l x l)            -- the Gram matrix of Cholesky-QR
//
// At m = 1.25e6, l = 80 (BASELINE config 4, one of eight shards) these read / write 400 MB each and are HBM work, but
// the general tall-skinny kernels (hip_kernels.hpp) ran them at ~2 TB/s: the l-deep reduction is 2 tiles, of which
// 37 % is zero padding, the skinny operand is restaged for every 64 rows (3.6 bytes of LDS fill per useful byte),
// and the Gram computes a 128 x 80 block of which 80 x 80 / 2 is needed.  Here the small matrix lives in registers as
// MFMA B fragments and Y streams through once:
//
//  * apply: no LDS at all.  A wave owns 64 rows; lane (p, cq) loads the float4 Y[r0 + 4p .. 4p+3][4j + cq] (16 lanes =
//    256 contiguous bytes of a column).  Component e of those loads IS the A fragment of the 16-row tile
//    {r0 + 4p + e}: MFMA does not care which rows form a tile, so the four row-interleaved tiles need no transpose,
//    and the D fragments of the four tiles re-interleave into float4 stores of 16 contiguous rows per lane.
//  * Gram: the MFMA M index must be a COLUMN of Y, i.e. adjacent lanes want addresses l apart; the tile therefore goes
//    through LDS (LDS-DMA, 128 rows x 16 NCT columns, pieces swizzled by column so the fragment reads are conflict-free),
//    every wave takes 32 of the rows and accumulates only the NCT (NCT + 1) / 2 tiles on or above the diagonal.
//    Partial Grams per workgroup go to slabs summed in fixed order (slab_reduce_deep_kernel).
#pragma once
#include "hip_kernels.hpp"

namespace corrla {
namespace k {

struct TallApplyArgs {
  const float* y;   // column-major m x kdim; rows up to round_up(m, 64) readable
  int64_t m, ld_y;
  int kdim;
  const float* mat;  // column-major kdim x n2
  int64_t ld_m;
  int n2;
  float* out;  // column-major m x n2
  int64_t ld_o;
  int out_cols;        // columns of out that may be written
  const float* scale;  // optional device scalar
  int vec_store;       // out is 16-byte aligned with ld_o % 4 == 0
};

// kdim <= 16 KTL, n2 <= 16 NCT.  grid: any (wave-strided over the 64-row blocks), block 256
template <int KTL, int NCT>
__global__ __launch_bounds__(256) void tall_apply_kernel(TallApplyArgs g) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  constexpr int NKS = 4 * KTL;  // MFMA k-steps
  const int lane = threadIdx.x & 63;
  const int fi = lane & 15, kq = lane >> 4;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t nblocks = (g.m + 63) / 64;
  if (gw >= nblocks) return;

  // B fragments: lane (n = fi, kq) of k-step j holds M[4 j + kq][16 ct + n]
  float bf[NKS][NCT];
#pragma unroll
  for (int j = 0; j < NKS; ++j)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int kk = 4 * j + kq, col = 16 * ct + fi;
      bf[j][ct] = (kk < g.kdim && col < g.n2) ? g.mat[(int64_t)col * g.ld_m + kk] : 0.f;
    }
  const float sc = g.scale ? *g.scale : 1.f;

  // lane (p = fi, cq = kq): rows r0 + 4 p .. + 3 of column 4 j + cq
  const float* ybase = g.y + (int64_t)kq * g.ld_y + 4 * fi;
  const int64_t jstride = 4 * g.ld_y;
  auto load = [&](f32x4_t (&v)[NKS], int64_t blk) {
    const float* src = ybase + blk * 64;
#pragma unroll
    for (int j = 0; j < NKS; ++j) {
      if (4 * j + kq < g.kdim)
        v[j] = *(const f32x4_t*)(src + j * jstride);
      else
        v[j] = (f32x4_t){0, 0, 0, 0};
    }
  };
  f32x4_t cur[NKS], nxt[NKS];
  load(cur, gw);
  for (int64_t blk = gw; blk < nblocks; blk += nw) {
    const bool more = blk + nw < nblocks;
    if (more) load(nxt, blk + nw);
    f32x4 acc[4][NCT];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[e][ct] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NKS; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[e][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[j][e], bf[j][ct], acc[e][ct], 0, 0, 0);
    // D fragment of tile e: lane (n = fi, rg = kq), register jp = tile row 4 rg + jp = matrix row r0 + 4 (4 rg + jp) + e
    const int64_t r0 = blk * 64 + 16 * kq;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int col = 16 * ct + fi;
      if (col < g.out_cols) {
        float* dst = g.out + (int64_t)col * g.ld_o + r0;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          const f32x4_t w = {acc[0][ct][jp] * sc, acc[1][ct][jp] * sc, acc[2][ct][jp] * sc, acc[3][ct][jp] * sc};
          const int64_t row = r0 + 4 * jp;
          if (g.vec_store && row + 3 < g.m) {
            *(f32x4_t*)(dst + 4 * jp) = w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (row + e < g.m) dst[4 * jp + e] = w[e];
          }
        }
      }
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < NKS; ++j) cur[j] = nxt[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct TallGramArgs {
  const float* y;  // column-major m x l, leading dimension ld (rows m .. ld-1 are zero)
  int64_t m, ld;
  int l;
  float* slab;  // [workgroup][column][row], leading dimension out_ld
  int64_t slab_stride, out_ld;
  int64_t rows_per_group;  // multiple of kGramRows
  const float* zero;       // >= 16 bytes of zeros
};

constexpr int kGramRows = 128;  // rows per LDS tile: 512 bytes per column
__host__ __device__ constexpr int gram_tile_bytes(int nct) { return 16 * nct * kGramRows * 4; }
__host__ __device__ constexpr int gram_stages(int nct) {
  const int s = (160 * 1024 - 4096) / gram_tile_bytes(nct);
  return s > 4 ? 4 : s;
}
__host__ __device__ constexpr int gram_lds_bytes(int nct) {
  const int ring = gram_stages(nct) * gram_tile_bytes(nct);
  const int red = 3 * (nct * (nct + 1) / 2) * 1024;  // cross-wave sum of the partial tiles (waves 1..3)
  return ring > red ? ring : red;
}

template <int NCT>
__global__ __launch_bounds__(256) void tall_gram_kernel(TallGramArgs g) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  constexpr int TILE = gram_tile_bytes(NCT);
  constexpr int NS = gram_stages(NCT);
  static_assert(NS >= 2, "ring too shallow");
  constexpr int NPAIR = NCT * (NCT + 1) / 2;
  constexpr int NCHUNK = 8 * NCT;  // 1-KiB DMA pieces per tile: two columns each
  static_assert(NCHUNK % 4 == 0, "pieces split evenly over the waves");
  constexpr int DPL = NCHUNK / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fi = lane & 15, kq = lane >> 4;
  const int64_t r_begin = (int64_t)blockIdx.x * g.rows_per_group;
  const int64_t r_end = min(g.ld, r_begin + g.rows_per_group);  // the zero padding rows may be read
  const int ntile = r_begin < r_end ? (int)((r_end - r_begin + kGramRows - 1) / kGramRows) : 0;

  // LDS image of a tile: column c at byte c * 512; its 32 16-byte pieces (4 rows each) sit at physical slot
  //   (q & 16) | ((q ^ c) & 15)   for logical piece q
  // so the 16 lanes of a fragment read (16 columns, one piece) hit 16 different 16-byte bank groups.
  auto stage = [&](int buf, int t) {
    char* rt = smem + buf * TILE;
    const int64_t row0 = r_begin + (int64_t)t * kGramRows;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const int ch = wave + 4 * i;            // columns 2 ch, 2 ch + 1
      const int c = 2 * ch + (lane >> 5);
      const int ps = lane & 31;               // physical slot
      const int q = (ps & 16) | ((ps ^ c) & 15);
      const int64_t row = row0 + 4 * q;
      const float* src = (row < r_end && c < g.l) ? g.y + (int64_t)c * g.ld + row : g.zero;
      glds16(src, rt + ch * 1024);
    }
  };
  for (int t = 0; t < NS - 1 && t < ntile; ++t) stage(t % NS, t);

  f32x4 acc[NPAIR];
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) acc[p] = (f32x4){0, 0, 0, 0};

  for (int t = 0; t < ntile; ++t) {
    if (t + NS - 2 < ntile)
      wait_vmcnt<(NS - 2) * DPL>();
    else
      wait_vmcnt<0>();
    wg_barrier();  // tile t is complete; every wave is done with tile t - 1
    if (t + NS - 1 < ntile) stage((t + NS - 1) % NS, t + NS - 1);
    const char* tb = smem + (t % NS) * TILE;
    // this wave: rows 32 wave .. + 31 of the tile = pieces 8 wave .. + 7; k-step group s (16 rows): pieces 4 s + kq
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = 8 * wave + 4 * s + kq;
      f32x4_t fr[NCT];
#pragma unroll
      for (int a = 0; a < NCT; ++a) {
        const int c = 16 * a + fi;
        fr[a] = *(const f32x4_t*)(tb + c * 512 + (((q & 16) | ((q ^ c) & 15)) << 4));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int p = 0;
#pragma unroll
        for (int a = 0; a < NCT; ++a)
#pragma unroll
          for (int b = a; b < NCT; ++b, ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[a][e], fr[b][e], acc[p], 0, 0, 0);
      }
    }
  }
  // ---- sum the four waves' partial tiles (fixed order) and write this workgroup's slab, both triangles ----
  __syncthreads();  // the ring is free
  float* red = (float*)smem;  // [3][NPAIR][256]
  if (wave > 0) {
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((wave - 1) * NPAIR + p) * 256 + j * 64 + lane] = acc[p][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* dst = g.slab + (int64_t)blockIdx.x * g.slab_stride;
    int p = 0;
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
      for (int b = a; b < NCT; ++b, ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = ((acc[p][j] + red[p * 256 + j * 64 + lane]) + red[(NPAIR + p) * 256 + j * 64 + lane]) +
                          red[(2 * NPAIR + p) * 256 + j * 64 + lane];
          // D: row = 16 a + 4 kq + j (a column index of Y), col = 16 b + fi
          const int gr = 16 * a + 4 * kq + j, gc = 16 * b + fi;
          dst[(int64_t)gc * g.out_ld + gr] = v;
          if (a != b) dst[(int64_t)gr * g.out_ld + gc] = v;
        }
      }
  }
}

}  // namespace k
}  // namespace corrla
