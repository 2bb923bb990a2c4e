// One-sweep power iteration: Zout = A^T (A Z) with A read from HBM ONCE.
//
// The reference schedule (random_svd.rs:35-56) forms Y = A Z and then Z' = A^T Y as two products, i.e. two passes over A
// and one write + one read of the m x l matrix Y per iteration.  For a tall row-major A with few columns (n <= 512) a
// block of 16 rows of A (<= 32 KB) sits in LDS, so both products are taken from the same staged tile:
//
//     T   = A_b Z            (16 x l, reduction over n)
//     Z' += A_b^T T          (n x l accumulators, reduction over the 16 rows)
//
// One workgroup of four waves (one per SIMD, up to 512 VGPRs each) owns a contiguous group of rows of A and ALL l <= 80
// columns: wave w keeps the k-quarter w of Z as MFMA B-fragments (4 NK NCT registers) for phase 1 and the n-quarter w of
// Z' as accumulators (4 NK NCT registers) for phase 2; the four partial T's meet in LDS (double-buffered, one barrier).
// Every byte of A therefore enters a CU once and feeds 2 * 2 * l flops: 3.2 B/clk per CU at l = 80, far below the
// ~10 B/clk a CU can pull from HBM.  (A first version gave each 16-column tile of Z its own workgroup: five CUs then
// pulled the same 64 KB tile -- 16 B/clk each -- and the kernel ran at the LDS-DMA fill rate, 17 % SLOWER than the two
// GEMMs it replaces; profiles/r02_f4_one_sweep.jsonl.)  The waves issue their own LDS-DMA (NK pieces per wave per tile,
// ring of up to 4 tiles); partial Z' per row group go to slabs that slab_reduce_deep_kernel sums in fixed order.
//
// f32 only (the case SURVEY names: BASELINE config 4, 10^7 x 512 f32).  Exact f32 MFMA; the result differs from the
// two-product form only by the rounding of Y to f32 in memory (here T stays in f32 too) and the summation order.
//
// Measured (profiles/r02_f4_one_sweep.jsonl, 1.25e6 x 512 shard, l = 74 -> 80): 2.23 ms per fused pass against
// 1.08 + 0.85 ms for the two GEMMs.  The pass is MFMA-bound, not HBM-bound: 2 * 2 m n l = 205 GFLOP of exact-f32 MFMA
// (16x16x4: 256 flop/clk/CU, 157 TFLOP/s on the chip) is >= 1.30 ms, while the 2.56 GB of A is 0.45 ms of HBM; the two
// GEMMs already run at 95-120 TFLOP/s each, so one sweep can win at most ~15 % of 1.9 ms and at one wave per SIMD (the
// register budget forces that) it exposes its LDS latencies and reaches 58 % of the MFMA peak.  Reading the LDS operands
// of a phase as one batch cost 38 spilled VGPRs and ran slower (2.83 ms).  The flag therefore stays off by default.
#pragma once
#include "hip_kernels.hpp"

namespace corrla {
namespace k {

constexpr int kAtaRows = 16;  // rows of A per LDS tile

struct AtaArgs {
  const float* a;       // row-major m x n
  int64_t m, n, lda, n_readable;
  const float* z;       // column-major n x L (zero padded: ld >= 64 * NK, column count padded to 16)
  int64_t z_ld;
  float* slab;          // [row group][column][n index], leading dimension out_ld
  int64_t slab_stride, out_ld;
  int64_t rows_per_group;  // multiple of kAtaRows
  int nrowgroups;
  const float* zero;       // >= 16 bytes of zeros
};

__host__ __device__ constexpr int ata_tile_bytes(int nk) { return kAtaRows * nk * 256; }
__host__ __device__ constexpr int ata_tpart_bytes(int nct) { return 2 * 4 * nct * 1024; }  // [parity][wave][ct][16 x 16]
__host__ __device__ constexpr int ata_stages(int nk, int nct) {
  const int s = (160 * 1024 - ata_tpart_bytes(nct)) / ata_tile_bytes(nk);
  return s > 4 ? 4 : s;
}
__host__ __device__ constexpr int ata_lds_bytes(int nk, int nct) {
  return ata_stages(nk, nct) * ata_tile_bytes(nk) + ata_tpart_bytes(nct);
}

// NK = reduction segments of 256 bytes per tile row (n padded to 64 NK), NCT = 16-column tiles of Z
template <int NK, int NCT>
__global__ __launch_bounds__(256) void ata_fused_kernel(AtaArgs g) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  constexpr int ROWB = NK * 256;            // bytes per tile row
  constexpr int TILE = ata_tile_bytes(NK);
  constexpr int NS = ata_stages(NK, NCT);
  static_assert(NS >= 2, "ring too shallow");
  constexpr int DPL = NK;                   // 1-KiB DMA pieces per wave per tile (4 NK pieces per tile)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tpart = (float*)(smem + NS * TILE);  // [2][4 waves][NCT][16 rows][16 cols]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = blockIdx.x;
  const int64_t r_begin = (int64_t)rg * g.rows_per_group;
  const int64_t r_end = min(g.m, r_begin + g.rows_per_group);
  const int nblk = (int)((r_end - r_begin + kAtaRows - 1) / kAtaRows);
  if (nblk <= 0) return;
  const int fi = lane & 15, kq = lane >> 4;

  // LDS-DMA of one tile: piece c = wave + 4 i covers the 256-byte segments 4c .. 4c+3 of the linear tile image;
  // physical 16-byte slot (lane & 15) of segment (row, sir) holds logical slot (lane & 15) ^ (row & 15)
  auto stage = [&](int buf, int blk) {
    char* rt = smem + buf * TILE;
    const int64_t row0 = r_begin + (int64_t)blk * kAtaRows;
#pragma unroll
    for (int i = 0; i < DPL; ++i) {
      const int c = wave + 4 * i;
      const int sg = 4 * c + (lane >> 4);
      const int row = sg / NK, sir = sg - row * NK;
      const int slot = (lane & 15) ^ (row & 15);
      const int64_t grow = row0 + row;
      const int64_t col = (int64_t)sir * 64 + slot * 4;
      const float* src = (grow < r_end && col < g.n_readable) ? g.a + grow * g.lda + col : g.zero;
      glds16(src, rt + c * 1024);
    }
  };
  for (int t = 0; t < NS - 1 && t < nblk; ++t) stage(t % NS, t);

  // Z as B fragments of phase 1: group G = wave * NK + gq, MFMA e of the group contracts k = 16 G + 4 kq + e
  f32x4_t zf[NK][NCT];
#pragma unroll
  for (int gq = 0; gq < NK; ++gq)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
    {
      // rows of Z beyond its (zero padded) leading dimension do not exist: they meet zero columns of the tile
      const int k0 = 16 * (wave * NK + gq) + 4 * kq;
      zf[gq][ct] = k0 < g.z_ld ? *(const f32x4_t*)(g.z + (int64_t)(16 * ct + fi) * g.z_ld + k0) : (f32x4_t){0, 0, 0, 0};
    }
  f32x4 acc[NK][NCT];  // phase 2: n tiles NK * wave .. NK * wave + NK - 1, all column tiles
#pragma unroll
  for (int t = 0; t < NK; ++t)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[t][ct] = (f32x4){0, 0, 0, 0};

  for (int i = 0; i < nblk; ++i) {
    // tile i has landed for THIS wave's pieces; the barrier makes every wave's pieces visible
    if (i + NS - 2 < nblk)
      wait_vmcnt<(NS - 2) * DPL>();
    else
      wait_vmcnt<0>();
    wg_barrier();  // A (also: every wave is done with tile i-1 and with the T buffer of block i-2)
    if (i + NS - 1 < nblk) stage((i + NS - 1) % NS, i + NS - 1);
    const char* tb = smem + (i % NS) * TILE;
    float* tp_buf = tpart + (i & 1) * (4 * NCT * 256);
    // ---- phase 1: partial T (16 rows x 16 NCT cols) over this wave's k quarter ----
    f32x4 tp[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) tp[ct] = (f32x4){0, 0, 0, 0};
    {
      const char* rbase = tb + fi * ROWB;  // A operand: row = fi
#pragma unroll
      for (int gq = 0; gq < NK; ++gq) {
        const int G = wave * NK + gq;  // k0 = 16 G + 4 kq: segment G / 4, slot 4 (G % 4) + kq
        const f32x4_t a4 = *(const f32x4_t*)(rbase + (G >> 2) * 256 + ((((G & 3) << 2) + kq) ^ fi) * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) tp[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], zf[gq][ct][e], tp[ct], 0, 0, 0);
      }
    }
    // D layout: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) tp_buf[((wave * NCT + ct) * 16 + 4 * kq + j) * 16 + fi] = tp[ct][j];
    // B: the four partial T's are in LDS (the stores must have landed: a bare s_barrier waits for no counter)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- phase 2: Z'[n tiles of this wave][all cols] += A_b^T T ----
    // B fragments: T[row = 4 s + kq][col = 16 ct + fi] = sum of the four partials
    float tfr[4][NCT];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int o = (ct * 16 + 4 * s + kq) * 16 + fi;
        tfr[s][ct] = (tp_buf[o] + tp_buf[NCT * 256 + o]) + (tp_buf[2 * NCT * 256 + o] + tp_buf[3 * NCT * 256 + o]);
      }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = 4 * s + kq;  // A operand: A_b[row][n = 16 nt + fi]
#pragma unroll
      for (int t = 0; t < NK; ++t) {
        const int nt = NK * wave + t;  // n tile: segment nt / 4, slots 4 (nt % 4) + fi / 4, word fi % 4
        const float av = *(const float*)(tb + row * ROWB + (nt >> 2) * 256 +
                                         ((((nt & 3) << 2) + (fi >> 2)) ^ (row & 15)) * 16 + (fi & 3) * 4);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, tfr[s][ct], acc[t][ct], 0, 0, 0);
      }
    }
  }
  // ---- partial Z' of this row group: slab[rg][col][n] ----
  float* dst = g.slab + (int64_t)rg * g.slab_stride;
#pragma unroll
  for (int t = 0; t < NK; ++t) {
    const int nt = NK * wave + t;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t nidx = 16 * nt + 4 * kq + j;
        if (nidx < g.n) dst[(int64_t)(16 * ct + fi) * g.out_ld + nidx] = acc[t][ct][j];
      }
  }
}

}  // namespace k
}  // namespace corrla
