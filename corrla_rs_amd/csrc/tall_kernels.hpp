// The m-sized products of the thin-Q stage on very tall sketches (random_svd.rs:38,57 qr().compute_thin_q(), :103 U = Q U~)
// for l <= 96 (f32) / l <= 64 (f64):
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
//  * apply: no LDS at all.  A wave owns 64 rows (f64: 32); lane (p, cq) loads the 16 bytes Y[r0 + 4p .. 4p+3][4j + cq]
//    (16 lanes = 256 contiguous bytes of a column).  Component e of those loads IS the A fragment of the 16-row tile
//    {r0 + 4p + e}: MFMA does not care which rows form a tile, so the VEC row-interleaved tiles need no transpose,
//    and the D fragments of the tiles re-interleave into 16-byte stores of consecutive rows.
//  * Gram: the MFMA M index must be a COLUMN of Y, i.e. adjacent lanes want addresses l apart; the tile therefore goes
//    through LDS (LDS-DMA, 512 bytes of each of 16 NCT columns, pieces swizzled by column so the fragment reads are conflict-free),
//    every wave takes a quarter of the rows and accumulates only the NCT (NCT + 1) / 2 tiles on or above the diagonal.
//    Partial Grams per workgroup go to slabs summed in fixed order (slab_reduce_deep_kernel).
#pragma once
#include "hip_kernels.hpp"

namespace corrla {
namespace k {

template <class T>
struct TallApplyArgs {
  const T* y;   // column-major m x kdim; rows up to round_up(m, 64) readable
  int64_t m, ld_y;
  int kdim;
  const T* mat;  // column-major kdim x n2
  int64_t ld_m;
  int n2;
  T* out;  // column-major m x n2
  int64_t ld_o;
  int out_cols;    // columns of out that may be written
  const T* scale;  // optional device scalar
  int vec_store;   // out is 16-byte aligned with ld_o * sizeof(T) % 16 == 0
  const int* run_if;  // optional device word: nothing happens when it is 0
};

// kdim <= 16 KTL, n2 <= 16 NCT.  grid: any (wave-strided over the row blocks), block 256
template <class T, int KTL, int NCT>
__global__ __launch_bounds__(256) void tall_apply_kernel(TallApplyArgs<T> g) {
  typedef typename MT<T>::vec_t vec_t;
  typedef typename MT<T>::acc_t acc_t;
  constexpr int VEC = MT<T>::VEC;
  constexpr int RB = 16 * VEC;  // rows per wave block
  constexpr int NKS = 4 * KTL;  // MFMA k-steps
  if (g.run_if && *g.run_if == 0) return;
  const int lane = threadIdx.x & 63;
  const int fi = lane & 15, kq = lane >> 4;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t nblocks = (g.m + RB - 1) / RB;
  if (gw >= nblocks) return;

  // B fragments: lane (n = fi, kq) of k-step j holds M[4 j + kq][16 ct + n]
  T bf[NKS][NCT];
#pragma unroll
  for (int j = 0; j < NKS; ++j)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int kk = 4 * j + kq, col = 16 * ct + fi;
      bf[j][ct] = (kk < g.kdim && col < g.n2) ? g.mat[(int64_t)col * g.ld_m + kk] : (T)0;
    }
  const T sc = g.scale ? *g.scale : (T)1;

  // lane (p = fi, cq = kq): the VEC rows of piece perm(p) of column 4 j + cq.  MFMA tile row i may stand for any matrix
  // row as long as loads and stores agree: f32 uses piece perm(i) = 4 (i % 4) + i / 4, so that the four lanes (rg) that
  // hold D rows 4 rg + jp store the ADJACENT 16-byte pieces 4 jp + rg -- 64 contiguous bytes per column and store
  // instruction instead of four isolated 16-byte pieces (f64 D rows are rg + 4 jp: adjacent already)
  auto piece = [](int i) { return sizeof(T) == 4 ? 4 * (i & 3) + (i >> 2) : i; };
  const T* ybase = g.y + (int64_t)kq * g.ld_y + VEC * piece(fi);
  const int64_t jstride = 4 * g.ld_y;
  auto load = [&](vec_t (&v)[NKS], int64_t blk) {
    const T* src = ybase + blk * RB;
#pragma unroll
    for (int j = 0; j < NKS; ++j) {
      if (4 * j + kq < g.kdim) {
        v[j] = *(const vec_t*)(src + j * jstride);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[j][e] = (T)0;
      }
    }
  };
  vec_t cur[NKS], nxt[NKS];
  load(cur, gw);
  for (int64_t blk = gw; blk < nblocks; blk += nw) {
    const bool more = blk + nw < nblocks;
    if (more) load(nxt, blk + nw);
    acc_t acc[VEC][NCT];
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[e][ct] = (acc_t){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NKS; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[e][ct] = MT<T>::mma(cur[j][e], bf[j][ct], acc[e][ct]);
    // D fragment of tile e: lane (n = fi), register jp = tile row drow(lane, jp) = matrix row r0 + VEC drow + e
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int col = 16 * ct + fi;
      if (col < g.out_cols) {
        T* dst = g.out + (int64_t)col * g.ld_o;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          const int64_t row = blk * RB + VEC * piece(MT<T>::drow(lane, jp));
          vec_t w;
#pragma unroll
          for (int e = 0; e < VEC; ++e) w[e] = acc[e][ct][jp] * sc;
          if (g.vec_store && row + VEC - 1 < g.m) {
            *(vec_t*)(dst + row) = w;
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
              if (row + e < g.m) dst[row + e] = w[e];
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
template <class T>
struct TallGramArgs {
  const T* y;  // column-major m x l, leading dimension ld (rows m .. ld-1 are zero)
  int64_t m, ld;
  int l;
  T* slab;  // [workgroup][column][row], leading dimension out_ld
  int64_t slab_stride, out_ld;
  int64_t rows_per_group;  // multiple of the tile rows
  const T* zero;           // >= 16 bytes of zeros
  const int* run_if;       // optional device word: nothing happens when it is 0
};

// rows per LDS tile: 512 bytes per column
template <class T>
__host__ __device__ constexpr int gram_rows() { return 512 / (int)sizeof(T); }
__host__ __device__ constexpr int gram_tile_bytes(int nct) { return 16 * nct * 512; }
__host__ __device__ constexpr int gram_stages(int nct) {
  const int s = (160 * 1024 - 4096) / gram_tile_bytes(nct);
  return s > 4 ? 4 : s;
}
__host__ __device__ constexpr int gram_lds_bytes(int nct, int esz) {
  const int ring = gram_stages(nct) * gram_tile_bytes(nct);
  const int red = 3 * (nct * (nct + 1) / 2) * 256 * esz;  // cross-wave sum of the partial tiles (waves 1..3)
  return ring > red ? ring : red;
}

template <class T, int NCT>
__global__ __launch_bounds__(256) void tall_gram_kernel(TallGramArgs<T> g) {
  typedef typename MT<T>::vec_t vec_t;
  typedef typename MT<T>::acc_t acc_t;
  constexpr int VEC = MT<T>::VEC;
  constexpr int ROWS = gram_rows<T>();
  constexpr int TILE = gram_tile_bytes(NCT);
  constexpr int NS = gram_stages(NCT);
  static_assert(NS >= 2, "ring too shallow");
  constexpr int NPAIR = NCT * (NCT + 1) / 2;
  constexpr int NCHUNK = 8 * NCT;  // 1-KiB DMA pieces per tile: two columns each
  static_assert(NCHUNK % 4 == 0, "pieces split evenly over the waves");
  constexpr int DPL = NCHUNK / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (g.run_if && *g.run_if == 0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fi = lane & 15, kq = lane >> 4;
  const int64_t r_begin = (int64_t)blockIdx.x * g.rows_per_group;
  const int64_t r_end = min(g.ld, r_begin + g.rows_per_group);  // the zero padding rows may be read
  const int ntile = r_begin < r_end ? (int)((r_end - r_begin + ROWS - 1) / ROWS) : 0;

  // LDS image of a tile: column c at byte c * 512; its 32 16-byte pieces (VEC rows each) sit at physical slot
  //   (q & 16) | ((q ^ c) & 15)   for logical piece q
  // so the 16 lanes of a fragment read (16 columns, one piece) hit 16 different 16-byte bank groups.
  auto stage = [&](int buf, int t) {
    char* rt = smem + buf * TILE;
    const int64_t row0 = r_begin + (int64_t)t * ROWS;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const int ch = wave + 4 * i;            // columns 2 ch, 2 ch + 1
      const int c = 2 * ch + (lane >> 5);
      const int ps = lane & 31;               // physical slot
      const int q = (ps & 16) | ((ps ^ c) & 15);
      const int64_t row = row0 + VEC * q;
      const T* src = (row < r_end && c < g.l) ? g.y + (int64_t)c * g.ld + row : g.zero;
      glds16(src, rt + ch * 1024);
    }
  };
  for (int t = 0; t < NS - 1 && t < ntile; ++t) stage(t % NS, t);

  acc_t acc[NPAIR];
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) acc[p] = (acc_t){0, 0, 0, 0};

  for (int t = 0; t < ntile; ++t) {
    if (t + NS - 2 < ntile)
      wait_vmcnt<(NS - 2) * DPL>();
    else
      wait_vmcnt<0>();
    wg_barrier();  // tile t is complete; every wave is done with tile t - 1
    if (t + NS - 1 < ntile) stage((t + NS - 1) % NS, t + NS - 1);
    const char* tb = smem + (t % NS) * TILE;
    // this wave: pieces 8 wave .. + 7 of every column; k-step group s: pieces 4 s + kq (4 VEC rows of the reduction)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = 8 * wave + 4 * s + kq;
      vec_t fr[NCT];
#pragma unroll
      for (int a = 0; a < NCT; ++a) {
        const int c = 16 * a + fi;
        fr[a] = *(const vec_t*)(tb + c * 512 + (((q & 16) | ((q ^ c) & 15)) << 4));
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        int p = 0;
#pragma unroll
        for (int a = 0; a < NCT; ++a)
#pragma unroll
          for (int b = a; b < NCT; ++b, ++p) acc[p] = MT<T>::mma(fr[a][e], fr[b][e], acc[p]);
      }
    }
  }
  // ---- sum the four waves' partial tiles (fixed order) and write this workgroup's slab, both triangles ----
  __syncthreads();  // the ring is free
  T* red = (T*)smem;  // [3][NPAIR][256]
  if (wave > 0) {
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[((wave - 1) * NPAIR + p) * 256 + j * 64 + lane] = acc[p][j];
  }
  __syncthreads();
  if (wave == 0) {
    T* dst = g.slab + (int64_t)blockIdx.x * g.slab_stride;
    int p = 0;
#pragma unroll
    for (int a = 0; a < NCT; ++a)
#pragma unroll
      for (int b = a; b < NCT; ++b, ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const T v = ((acc[p][j] + red[p * 256 + j * 64 + lane]) + red[(NPAIR + p) * 256 + j * 64 + lane]) +
                      red[(2 * NPAIR + p) * 256 + j * 64 + lane];
          // D: row = 16 a + drow (a column index of Y), col = 16 b + fi
          const int gr = 16 * a + MT<T>::drow(lane, j), gc = 16 * b + fi;
          dst[(int64_t)gc * g.out_ld + gr] = v;
          if (a != b) dst[(int64_t)gr * g.out_ld + gc] = v;
        }
      }
  }
}

}  // namespace k
}  // namespace corrla
