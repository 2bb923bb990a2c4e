// One-sweep power iteration: Zout = A^T (A Z) with A read from HBM ONCE.
//
// The reference schedule (random_svd.rs:35-56) forms Y = A Z and then Z' = A^T Y as two products, i.e. two passes over A
// and one write + one read of the m x l matrix Y per iteration.  For a tall row-major A with few columns (n <= 512) a
// block of 32 rows of A (<= 64 KB) fits in LDS, so both products can be taken from the same staged tile:
//
//     T  = A_b Z            (32 x 16 per workgroup, reduction over n)
//     Z' += A_b^T T         (n x 16 accumulators, reduction over the 32 rows)
//
// A workgroup owns ONE 16-column tile of Z for a contiguous group of rows of A: its slice of Z lives in registers as
// MFMA B-fragments for the whole kernel, its slice of Z' in accumulators, and the only LDS traffic is the A tile
// (LDS-DMA ring, dedicated loader waves as in the tall GEMMs) plus the 2 KB hand-over of T between the two phases.
// The ceil(l / 16) workgroups that share a row group are placed on the same XCD (blockIdx % 8 is the XCD under
// round-robin dispatch -- a speed matter only), so A comes from HBM once and from that XCD's L2 for the others.
// Partial Z' per row group go to slabs that slab_reduce_deep_kernel sums in fixed order (bit-reproducible).
//
// f32 only (the case SURVEY names: BASELINE config 4, 10^7 x 512 f32).  Exact f32 MFMA; the result differs from the
// two-product form only by the rounding of Y to f32 in memory (here T stays in f32 too) and the summation order.
#pragma once
#include "hip_kernels.hpp"

namespace corrla {
namespace k {

constexpr int kAtaRows = 32;  // rows of A per LDS tile

struct AtaArgs {
  const float* a;       // row-major m x n
  int64_t m, n, lda, n_readable;
  const float* z;       // column-major n x L (zero padded: ld >= 64 * NK, column count padded to 16)
  int64_t z_ld;
  float* slab;          // [row group][column][n index], leading dimension out_ld
  int64_t slab_stride, out_ld;
  int64_t rows_per_group;  // multiple of kAtaRows
  int nrowgroups, nct;     // row groups, 16-column tiles
  const float* zero;       // >= 16 bytes of zeros
};

__host__ __device__ constexpr int ata_tile_bytes(int nk) { return kAtaRows * nk * 256; }
__host__ __device__ constexpr int ata_stages(int nk) { return (3 * ata_tile_bytes(nk) + 4096 <= 160 * 1024) ? 3 : 2; }
__host__ __device__ constexpr int ata_lds_bytes(int nk) { return ata_stages(nk) * ata_tile_bytes(nk) + 4096; }

// NK = ceil(n / 64) reduction segments of 256 bytes per row of the tile
template <int NK>
__global__ __launch_bounds__(512) void ata_fused_kernel(AtaArgs g) {
  constexpr int ROWB = NK * 256;            // bytes per tile row
  constexpr int TILE = ata_tile_bytes(NK);
  constexpr int NS = ata_stages(NK);
  constexpr int NCH = kAtaRows * NK / 4;    // 1-KiB DMA pieces per tile (4 segments each)
  static_assert(NCH % 4 == 0, "pieces must split evenly over the 4 loader waves");
  constexpr int DPL = NCH / 4;              // DMA instructions per loader wave per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tbuf = (float*)(smem + NS * TILE);  // [2 K-halves][2 row tiles][16 rows][16 cols]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroup -> (row group, column tile): the nct workgroups of one row group share blockIdx % 8
  const int wg = blockIdx.x, xcd = wg & 7, q = wg >> 3;
  const int cb = q % g.nct;
  const int rg = (q / g.nct) * 8 + xcd;
  if (rg >= g.nrowgroups) return;
  const int64_t r_begin = (int64_t)rg * g.rows_per_group;
  const int64_t r_end = min(g.m, r_begin + g.rows_per_group);
  const int nblk = (int)((r_end - r_begin + kAtaRows - 1) / kAtaRows);
  if (nblk <= 0) return;

  if (wave >= 4) {
    // ---- loader waves: tile i+NS-1 is issued while the MFMA waves work on tile i --------------------------------
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - 4;
    auto stage = [&](int buf, int blk) {
      char* rt = smem + buf * TILE;
      const int64_t row0 = r_begin + (int64_t)blk * kAtaRows;
#pragma unroll
      for (int i = 0; i < DPL; ++i) {
        const int c = lw + 4 * i;                   // piece: segments 4c .. 4c+3 of the linear tile image
        const int sg = 4 * c + (lane >> 4);         // 256-byte segment index = row * NK + segment-in-row
        const int row = sg / NK, sir = sg - row * NK;
        const int slot = (lane & 15) ^ (row & 15);  // physical slot (lane & 15) holds logical slot `slot`
        const int64_t grow = row0 + row;
        const int64_t col = (int64_t)sir * 64 + slot * 4;
        const float* src = (grow < r_end && col < g.n_readable) ? g.a + grow * g.lda + col : g.zero;
        glds16(src, rt + c * 1024);
      }
    };
    for (int t = 0; t < NS - 1 && t < nblk; ++t) stage(t % NS, t);
    for (int i = 0; i < nblk; ++i) {
      if (NS > 2 && i + NS - 2 < nblk)
        wait_vmcnt<(NS - 2) * DPL>();
      else
        wait_vmcnt<0>();
      wg_barrier();  // A: tile i visible; the MFMA waves are done with tile i-1
      if (i + NS - 1 < nblk) stage((i + NS - 1) % NS, i + NS - 1);
      wg_barrier();  // B: (T hand-over of the MFMA waves)
    }
    return;
  }

  // ---- MFMA waves ---------------------------------------------------------------------------------------------------
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const int fi = lane & 15, kq = lane >> 4;
  const int rt = wave & 1, kh = wave >> 1;  // phase 1: row tile, K half
  constexpr int GPH = 2 * NK;               // 16-wide k groups per half (K = 64 NK)
  const int64_t col0 = (int64_t)cb * 16;
  // Z slice as B fragments of phase 1: group G = kh * GPH + gq, MFMA e of the group contracts k = 16 G + 4 kq + e
  f32x4_t zf[GPH];
#pragma unroll
  for (int gq = 0; gq < GPH; ++gq)
    zf[gq] = *(const f32x4_t*)(g.z + (col0 + fi) * g.z_ld + 16 * (kh * GPH + gq) + 4 * kq);
  f32x4 acc[NK];  // phase 2: n tiles NK * wave .. NK * wave + NK - 1
#pragma unroll
  for (int t = 0; t < NK; ++t) acc[t] = (f32x4){0, 0, 0, 0};

  for (int i = 0; i < nblk; ++i) {
    wg_barrier();  // A
    const char* tb = smem + (i % NS) * TILE;
    // phase 1: partial T (16 rows x 16 cols) over this wave's K half
    // (four independent accumulation chains: a dependent 16x16x4 f32 MFMA issues every 40 cycles, an independent one
    // every 32)
    f32x4 tp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) tp[e] = (f32x4){0, 0, 0, 0};
    {
      const int row = 16 * rt + fi;
      const char* rbase = tb + row * ROWB;
#pragma unroll
      for (int gq = 0; gq < GPH; ++gq) {
        const int G = kh * GPH + gq;  // k0 = 16 G + 4 kq: segment G / 4, slot 4 (G % 4) + kq
        const f32x4_t a4 = *(const f32x4_t*)(rbase + (G >> 2) * 256 + ((((G & 3) << 2) + kq) ^ (row & 15)) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) tp[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], zf[gq][e], tp[e], 0, 0, 0);
      }
    }
    const f32x4 tacc = (tp[0] + tp[1]) + (tp[2] + tp[3]);
    // D layout: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < 4; ++j) tbuf[((kh * 2 + rt) * 16 + 4 * kq + j) * 16 + fi] = tacc[j];
    // B: both halves of T are in LDS (the LDS stores must have landed: a bare s_barrier waits for no counter)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // phase 2 B fragments: T[row = 4 s + kq][col = fi], s = 0 .. 7 over the 32 rows
    float tfr[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int row = 4 * s + kq;  // row tile row >> 4, row-in-tile row & 15
      const int o = ((row >> 4) * 16 + (row & 15)) * 16 + fi;
      tfr[s] = tbuf[o] + tbuf[2 * 16 * 16 + o];
    }
    // phase 2: Zacc[n tile][cols] += A_b^T (n x 32) T (32 x 16); A operand = A_b[row = 4 s + kq][n = 16 t + fi]
    // (k-step outermost: consecutive MFMAs go to different accumulators)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int row = 4 * s + kq;
#pragma unroll
      for (int t = 0; t < NK; ++t) {
        const int nt = NK * wave + t;  // n tile: segment nt / 4, slots 4 (nt % 4) + fi / 4, word fi % 4
        const float av = *(const float*)(tb + row * ROWB + (nt >> 2) * 256 + ((((nt & 3) << 2) + (fi >> 2)) ^ (row & 15)) * 16 +
                                         (fi & 3) * 4);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tfr[s], acc[t], 0, 0, 0);
      }
    }
  }
  // ---- partial Z' of this row group: slab[rg][col][n] ----------------------------------------------------------------
  float* dst = g.slab + (int64_t)rg * g.slab_stride;
#pragma unroll
  for (int t = 0; t < NK; ++t) {
    const int nt = NK * wave + t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t nidx = 16 * nt + 4 * kq + j;
      if (nidx < g.n) dst[(col0 + fi) * g.out_ld + nidx] = acc[t][j];
    }
  }
}

}  // namespace k
}  // namespace corrla
