// SURVEY 8 f4, second half: the tall products of the range finder (random_svd.rs:31, 42-51) on the bf16 matrix
// pipe with an f32 accumulator, f32 operands split on the fly ("bf16x3" / "bf16x6").  OFF BY DEFAULT
// (CORRLA_SKETCH_BF16X3 / CORRLA_SKETCH_BF16X6): the headline path stays exact f32.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate (64 flop/clk/SIMD); v_mfma_f32_16x16x32_bf16 at 16x
// that.  An f32 value is the exact sum of three bf16 pieces (8 significant bits each, round-to-nearest leaves
// |x - hi - mid - lo| <= 2^-27 |x|), and products of bf16 pieces are exact in the f32 accumulator, so
//     x y  =  hi hi' + (hi mid' + mid hi') + (mid mid' + hi lo' + lo hi')  + O(2^-27 |x y|)         ("bf16x6")
// reproduces the f32 product to f32 rounding with 6 MFMAs at 1/16 the cost each (2.7x the exact-f32 rate), and the
// two-piece form  hi hi' + hi lo' + lo hi'  + O(2^-16)  ("bf16x3") doubles that again.  Either way the products stop
// being bound by the matrix pipe and become bound by the stream of A from HBM (69 flop/B at l = 138).
//
// gemm_nn_bf16s:  Out (M x L, col-major) = R (M x K, row-major f32, streamed once) * X (K x L)
// gemm_tn_bf16s:  Out (K x L, col-major) = R^T * X (M x L)
//   R: the big operand in f32, staged by LDS-DMA as it sits in HBM and split in REGISTERS by the MFMA waves (2 x
//      v_cvt_pk_bf16_f32 + shifts / masks / subtractions per pair of elements and level);
//   X: the skinny operand, split ONCE per product by split_planes_kernel into NP bf16 planes whose reduction index is
//      stored in MFMA fragment order, so that a lane's 8 k values of a 16x16x32 step are one 16-byte LDS read.
//
// Workgroup = 8 MFMA waves (two per SIMD: one's LDS reads and split arithmetic run under the other's MFMAs) + 4
// loader waves; it owns 256 outer indices (wave w: 32 of them = 2 MFMA row tiles) and all NT <= 9 column tiles, so R is
// read from HBM once.  Reduction tile: 32 deep = ONE 16x16x32 step per (row tile, column tile, product); LDS stage =
// 32 KiB of R (ring of 3: two tiles of HBM latency in flight) + NP * NT * 1 KiB of X planes (ring of 2, L2-served).
//
// MFMA operand maps (cdna_hip_programming.md section 3): A[row = lane & 15][k = 8 (lane >> 4) + j],
// B[k = 8 (lane >> 4) + j][col = lane & 15], j = 0..7;  D: col = lane & 15, row = 4 (lane >> 4) + reg.  The big operand
// is the A side (a lane's four D registers are four consecutive outer indices of one output column: one 16-byte
// store).  k is a dummy summation index: lane group g = lane >> 4 is fed reduction indices
//     kmap(g, j) = 4 g + j  (j < 4),   16 + 4 g + (j - 4)  (j >= 4)          within each 32-deep tile
// on BOTH sides, because that is what two 16-byte reads of a 128-byte f32 row deliver (slots g and 4 + g).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hip_kernels.hpp"

namespace corrla {
namespace k {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMxWaves = 8;                  // MFMA waves per workgroup
constexpr int kMxLoaders = 4;                // LDS-DMA loader waves
constexpr int kMxRowTiles = 2;               // 16-wide outer tiles per MFMA wave
constexpr int kMxOuter = kMxWaves * kMxRowTiles * 16;  // 256 outer indices per workgroup
constexpr int kMxKT = 32;                    // reduction indices per tile (one 16x16x32 step)
constexpr int kMxBigBytes = kMxOuter * kMxKT * 4;       // 32 KiB
__host__ __device__ constexpr int mx_plane_bytes(int nt) { return nt * 16 * kMxKT * 2; }  // one plane of one tile
// Two LDS rings: the big operand comes from HBM and is prefetched TWO tiles ahead (3 slots of 32 KiB); the planes of the
// skinny operand are re-read by every workgroup, i.e. served by L2, and run one tile ahead (2 slots).  (Round 3's first
// version had one ring of whole tiles: only 2 fit for three planes, and the DMA cost 30 % on top of the DMA-free time.)
constexpr int kMxASlots = 3, kMxBSlots = 2;
// np = 0: the EXACT f32 variant of the same skeleton (v_mfma_f32_16x16x4_f32): the skinny operand is staged as f32, in the
// nn layout of the big image (columns of X = rows of 128 bytes)
__host__ __device__ constexpr int mx_bslot_bytes(int nt, int np) { return np ? np * mx_plane_bytes(nt) : nt * 16 * kMxKT * 4; }
__host__ __device__ constexpr int mx_lds_bytes(int nt, int np) { return kMxASlots * kMxBigBytes + kMxBSlots * mx_bslot_bytes(nt, np) + 1024; }

// reduction index (inside a 32-deep tile) that position p = 8 g + j of a plane row / MFMA fragment holds
__host__ __device__ constexpr int mx_kmap(int p) { return (p & 4) ? 16 + 4 * (p >> 3) + (p & 3) : 4 * (p >> 3) + (p & 3); }

// two f32 -> two bf16 (round to nearest even), packed: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned mx_pack(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// exact f32 subtraction kept scalar: hipcc would pair two of them into v_pk_add_f32, which costs more than the two
// scalar instructions next to MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
__device__ __forceinline__ float mx_sub(float a, float b) {
  float r;
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// x[0..7] -> NP fragments of 8 bf16 (hi, mid, lo): piece p of element e is bf16_rne(x_e - sum of the earlier pieces)
template <int NP>
__device__ __forceinline__ void mx_split8(const float (&x)[8], bf16x8 (&frag)[NP]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = x[e];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    u32x4 w;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const unsigned pk = mx_pack(r[2 * h], r[2 * h + 1]);
      w[h] = pk;
      if (p + 1 < NP) {  // residual for the next piece: exact (the piece shares the leading bits of r)
        r[2 * h] = mx_sub(r[2 * h], __builtin_bit_cast(float, pk << 16));
        r[2 * h + 1] = mx_sub(r[2 * h + 1], __builtin_bit_cast(float, pk & 0xffff0000u));
      }
    }
    frag[p] = __builtin_bit_cast(bf16x8, w);
  }
}

// ---- skinny operand -> NP bf16 planes --------------------------------------------------------------------------------
// x: column-major, ld elements per column, zero padded (rows [rows, ld) and columns up to ncols are readable zeros).
// planes[p][col][q]: ld bf16 per column; q = 32 t + 8 g + j holds piece p of x(32 t + kmap(g, j), col).  One thread
// per 16-byte output slot (8 positions).  run_if: see GemmArgs.
template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int64_t ld, int64_t ncols, __bf16* planes,
                                                           int64_t plane_stride, const int* run_if) {
  if (run_if && *run_if == 0) return;
  const int64_t slots_per_col = ld / 8;
  const int64_t total = slots_per_col * ncols;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = s / slots_per_col, sl = s - col * slots_per_col;
    const int64_t t = sl >> 2;
    const int g = (int)(sl & 3);
    const float* src = x + col * ld + 32 * t;
    const f32x4 lo4 = *(const f32x4*)(src + 4 * g);
    const f32x4 hi4 = *(const f32x4*)(src + 16 + 4 * g);
    const float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    bf16x8 fr[NP];
    mx_split8<NP>(v, fr);
#pragma unroll
    for (int p = 0; p < NP; ++p) *(bf16x8*)(planes + p * plane_stride + col * ld + 8 * sl) = fr[p];
  }
}

struct MxArgs {
  const float* r;        // big operand, row-major f32
  int64_t r_rows, r_cols, r_ld, r_cols_readable;
  const __bf16* planes;  // NP planes of the skinny operand (split_planes_kernel), x_ld bf16 per column
  int64_t x_ld, plane_stride;
  float* out;            // skinny result, column-major f32
  int64_t out_ld, out_cols;
  float* slab;           // partial results when nsplit > 1: slab[z][col][outer]
  int64_t slab_stride;
  const float* scale;    // optional device scalar (nsplit == 1 only)
  const float* zero;     // >= 16 bytes of zeros
  int tiles_total, tiles_per_split, nsplit;
  const int* run_if;
  int vec_store;
  int debug_flags;       // timing-only ablations (wrong results): 2 = no plane DMA, 4 = no big-operand DMA after the first tile
};

// swizzles of the two LDS images (both applied on the DMA's per-lane SOURCE address and again on the read):
//   big image, 128-byte rows (8 slots):  physical slot = logical ^ ((row >> 1) & 7)   -- two rows per 256-byte bank line
//   plane image, 64-byte rows (4 slots): physical slot = logical ^ ((-(row >> 2)) & 3) -- four rows per bank line
// With them the four 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) touch 16
// different 16-byte slots of the bank line (checked by tools/lds_layout_check.py for every lane group).
__host__ __device__ constexpr int mx_big_swz(int row) { return (row >> 1) & 7; }
__host__ __device__ constexpr int mx_plane_swz(int row) { return (-(row >> 2)) & 3; }

template <int NT, int NP>
__device__ __forceinline__ void mx_store(const MxArgs& g, const f32x4 (&acc)[NT], int64_t outer0, int64_t outer_limit, int lane) {
  float* dst;
  float sc = 1.f;
  if (g.nsplit > 1) {
    dst = g.slab + (int64_t)blockIdx.z * g.slab_stride;
  } else {
    dst = g.out;
    if (g.scale) {
      sc = *g.scale;
      asm volatile("" : "+v"(sc));  // retire the load here (hip_kernels.hpp: store_tile)
    }
  }
  const int64_t outer = outer0 + 4 * (lane >> 4);
  if (outer >= outer_limit) return;
  const bool whole = g.vec_store && outer + 3 < outer_limit;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t col = 16 * t + (lane & 15);
    if (g.nsplit > 1 || col < g.out_cols) {
      float* p = dst + col * g.out_ld + outer;
      if (whole) {
        *(f32x4*)p = acc[t] * sc;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (outer + j < outer_limit) p[j] = acc[t][j] * sc;
      }
    }
  }
}

// the products of one (row tile, column tile, 32-deep step): smallest terms first
template <int NP>
__device__ __forceinline__ f32x4 mx_products(const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x4 c) {
  if constexpr (NP == 3) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);  // lo  hi
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);  // hi  lo
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);  // mid mid
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);  // mid hi
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);  // hi  mid
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);  // hi  hi
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);  // lo hi
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);  // hi lo
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);  // hi hi
  }
  return c;
}

// ---------------------------------------------------------------------------------------------------------------------
// TN = false:  Out = R X      grid = (ceil(R_rows / 256), 1, nsplit), reduction over the columns of R
// TN = true :  Out = R^T X    grid = (ceil(R_cols / 256), 1, nsplit), reduction over the rows of R
// Big image of a stage:
//   nn: 256 outer rows x 128 bytes (32 reduction indices of one row of R);  chunk c (1 KiB) = rows 8 c .. 8 c + 7
//   tn: 32 reduction rows x 1 KiB (256 outer columns of one row of R);      chunk c = reduction row c; its 64 slots are
//       swizzled by  physical = logical ^ (4 * (row & 3) ... see mx_tn_swz): a fragment read walks DOWN the rows
// Plane images: NT * 16 columns x 64 bytes per plane;  chunk c = columns 16 c .. 16 c + 15.
// ---------------------------------------------------------------------------------------------------------------------
// tn: element (reduction row kr, outer column oc) at  kr * 1024 + ((oc >> 2) ^ mx_tn_swz(kr)) * 16 + (oc & 3) * 4.
// A fragment read is 8 x ds_read_b32 down the rows kmap(g, j); the two 32-lane halves of such a read are lane groups
// g = {0, 1} and {2, 3}, i.e. rows 4 apart at the same column -> the swizzle must move rows kr and kr + 4 to different
// banks: flip the 64-byte block (16 banks) with bit 2 of the row.
__host__ __device__ constexpr int mx_tn_swz(int kr) { return ((kr >> 2) & 1) << 2; }

template <int NT, int NP, bool TN>
__global__ __launch_bounds__(64 * (kMxWaves + kMxLoaders), 3) void gemm_bf16s_kernel(MxArgs g) {
  constexpr int PLANE = mx_plane_bytes(NT);
  constexpr int BSLOT = mx_bslot_bytes(NT, NP);
  constexpr int BRING = kMxASlots * kMxBigBytes;  // byte offset of the plane ring
  static_assert(mx_lds_bytes(9, 3) <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (g.run_if && *g.run_if == 0) return;  // uniform over the grid
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t outer_first = (int64_t)blockIdx.x * kMxOuter;
  const int t_begin = blockIdx.z * g.tiles_per_split;
  const int t_end = min(t_begin + g.tiles_per_split, g.tiles_total);
  const int nk = t_end - t_begin;

  if (wave >= kMxWaves) {
    // ---- loader waves.  Chunks of 1 KiB dealt round-robin; every loader issues the same number of DMA instructions per
    // tile and ring (the short ones add a dummy into the scratch KiB), so the counted wait is uniform.
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - kMxWaves;
    constexpr int NBIG = kMxBigBytes / 1024;                        // 32 chunks of the big operand per tile
    constexpr int DA = NBIG / kMxLoaders;                           // 8 per loader
    constexpr int NSK = NP ? NP * NT : 2 * NT;                      // plane chunks per tile (f32 image: 2 KiB per column tile)
    constexpr int DB = (NSK + kMxLoaders - 1) / kMxLoaders;         // per loader (padded)
    char* scratch = smem + BRING + kMxBSlots * BSLOT;
    auto stage_big = [&](int slot, int kt) {
      char* st = smem + slot * kMxBigBytes;
      const int64_t k0 = (int64_t)kt * kMxKT;
#pragma unroll
      for (int i = 0; i < DA; ++i) {
        const int c = lw + kMxLoaders * i;
        if ((g.debug_flags & 4) && kt > t_begin + 2) {
          glds16(g.zero, scratch);
          continue;
        }
        const float* src;
        if constexpr (!TN) {
          const int row = 8 * c + (lane >> 3);
          const int ls = (lane & 7) ^ mx_big_swz(row);
          const int64_t grow = outer_first + row, kk = k0 + 4 * ls;
          src = (grow < g.r_rows && kk < g.r_cols_readable) ? g.r + grow * g.r_ld + kk : g.zero;
        } else {
          const int kr = c;  // reduction row of the tile
          const int ls = lane ^ mx_tn_swz(kr);
          const int64_t grow = k0 + kr, oc = outer_first + 4 * ls;
          src = (grow < g.r_rows && oc < g.r_cols_readable) ? g.r + grow * g.r_ld + oc : g.zero;
        }
        glds16(src, st + c * 1024);
      }
    };
    auto stage_planes = [&](int slot, int kt) {
      char* st = smem + BRING + slot * BSLOT;
      const int64_t k0 = (int64_t)kt * kMxKT;
#pragma unroll
      for (int i = 0; i < DB; ++i) {
        const int cc = lw + kMxLoaders * i;
        if (cc >= NSK || ((g.debug_flags & 2) && kt > t_begin + 2)) {
          glds16(g.zero, scratch);  // keeps the per-tile DMA count uniform over the loaders
          continue;
        }
        if constexpr (NP == 0) {
          const int row = 8 * cc + (lane >> 3);  // column of X
          const int ls = (lane & 7) ^ mx_big_swz(row);
          glds16((const float*)g.planes + (int64_t)row * g.x_ld + k0 + 4 * ls, st + cc * 1024);
        } else {
          const int p = cc / NT, ct = cc - p * NT;
          const int row = 16 * ct + (lane >> 2);
          const int ls = (lane & 3) ^ mx_plane_swz(row);
          glds16(g.planes + p * g.plane_stride + (int64_t)row * g.x_ld + k0 + 8 * ls, st + p * PLANE + ct * 1024);
        }
      }
    };
    // issue order (vmcnt retires in order):  B(0) A(0) A(1) | then per tile i, after its barrier:  B(i+1) A(i+2).
    // Before barrier i the planes and the big tile of i must have landed; the ONE younger group, A(i+1), may stay in flight.
    if (nk > 0) {
      stage_planes(0, t_begin);
      stage_big(0, t_begin);
      if (nk > 1) stage_big(1, t_begin + 1);
    }
    for (int i = 0; i < nk; ++i) {
      if (i + 1 < nk)
        wait_vmcnt<DA>();
      else
        wait_vmcnt<0>();
      wg_barrier();  // tile i visible to the MFMA waves; they are done with tile i - 1 (its slots are free)
      if (i + 1 < nk) stage_planes((i + 1) % kMxBSlots, t_begin + i + 1);
      if (i + 2 < nk) stage_big((i + 2) % kMxASlots, t_begin + i + 2);
    }
    return;
  }

  // ---- MFMA waves -------------------------------------------------------------------------------------------------
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[kMxRowTiles][NT];
#pragma unroll
  for (int mw = 0; mw < kMxRowTiles; ++mw)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mw][t] = (f32x4){0, 0, 0, 0};
  // byte offsets inside a stage: one base per image, everything else is an immediate (row tile mw: + 16 rows, column
  // tile t: + 16 plane rows, plane p: + PLANE).  The swizzles only depend on fr: the bases are multiples of 16 rows.
  //   nn: row = 32 wave + 16 mw + fr, slots fg and 4 + fg (the second = the first ^ 64 bytes)
  //   tn: outer column oc = 32 wave + 16 mw + fr; the per-row swizzle is applied below
  const unsigned a_base = !TN ? (unsigned)((32 * wave + fr) * 128 + ((fg ^ mx_big_swz(fr)) << 4))
                              : (unsigned)((((32 * wave + fr) >> 2) << 4) + ((fr & 3) << 2));
  const unsigned b_base = (unsigned)(fr * 64 + ((fg ^ mx_plane_swz(fr)) << 4));  // inside a slot of the plane ring
  int abuf = 0, bbuf = 0;
  for (int i = 0; i < nk; ++i) {
    wg_barrier();  // matches the loaders' barrier: tile i is in LDS
    const char* st = smem + abuf * kMxBigBytes;            // big-operand slot
    const char* sb = smem + BRING + bbuf * BSLOT;          // plane slot
    // Schedule of one tile (everything below is one basic block; round 3 measured the first version -- both splits, then
    // nine groups of [3 LDS reads, wait, 12 MFMAs] -- at 55 % matrix-pipe occupancy with NO DMA at all: the two waves of a
    // SIMD run in step, so both split, then both wait on LDS):
    //   raw reads of both row tiles and the plane fragments of the first T1 column tiles go out together;
    //   split row tile 0; its products over column tiles 0 .. T1-1 (fragments stay in registers) cover the split of row
    //   tile 1 -- VALU and MFMA issue side by side, 2 VALU slots per 16-cycle MFMA; then row tile 1 over the same
    //   fragments; the remaining column tiles read their fragments one tile ahead of their MFMAs.
    constexpr int T1 = NT < 4 ? NT : 4;
    float xr[kMxRowTiles][8];
#pragma unroll
    for (int mw = 0; mw < kMxRowTiles; ++mw) {
      if constexpr (!TN) {
        const f32x4 v0 = *(const f32x4*)(st + (a_base + mw * 2048));
        const f32x4 v1 = *(const f32x4*)(st + ((a_base + mw * 2048) ^ 64u));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xr[mw][j] = v0[j];
          xr[mw][4 + j] = v1[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // reduction row kmap(8 fg + j) = 4 fg + j (j < 4) / 16 + 4 fg + (j - 4): bit 2 of it, which drives the swizzle,
          // is bit 0 of fg for every j
          const unsigned off = (a_base + mw * 64) ^ ((unsigned)(fg & 1) << 6);
          xr[mw][j] = *(const float*)(st + (4 * fg + (j < 4 ? j : 12 + j)) * 1024 + off);
        }
      }
    }
    if constexpr (NP == 0) {
      // exact f32: 8 x v_mfma_f32_16x16x4_f32 per (row tile, column tile); step s multiplies reduction index kmap(g, s) on
      // both sides.  The fragment of the next column tile is read under the MFMAs of this one.
      const unsigned x_base = (unsigned)(fr * 128 + ((fg ^ mx_big_swz(fr)) << 4));
      f32x4 n0 = *(const f32x4*)(sb + x_base), n1 = *(const f32x4*)(sb + (x_base ^ 64u));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 c0 = n0, c1 = n1;
        if (t + 1 < NT) {
          n0 = *(const f32x4*)(sb + (x_base + (t + 1) * 2048));
          n1 = *(const f32x4*)(sb + ((x_base + (t + 1) * 2048) ^ 64u));
        }
#pragma unroll
        for (int mw = 0; mw < kMxRowTiles; ++mw) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc[mw][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[mw][s4], c0[s4], acc[mw][t], 0, 0, 0);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc[mw][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[mw][4 + s4], c1[s4], acc[mw][t], 0, 0, 0);
        }
      }
    } else {
      bf16x8 bk[T1][NP];
  #pragma unroll
      for (int t = 0; t < T1; ++t)
  #pragma unroll
        for (int p = 0; p < NP; ++p) bk[t][p] = *(const bf16x8*)(sb + b_base + t * 1024 + p * PLANE);
      bf16x8 af[kMxRowTiles][NP];
      mx_split8<NP>(xr[0], af[0]);
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int t = 0; t < T1; ++t) acc[0][t] = mx_products<NP>(af[0], bk[t], acc[0][t]);
      mx_split8<NP>(xr[1], af[1]);
      bf16x8 bn[NP];
      if constexpr (T1 < NT) {
  #pragma unroll
        for (int p = 0; p < NP; ++p) bn[p] = *(const bf16x8*)(sb + b_base + T1 * 1024 + p * PLANE);
      }
      // hipcc hoists the whole second split above the first MFMA otherwise: one MFMA, then two of the split's VALU
      // instructions (an MFMA holds the SIMD's vector issue for 8 of its 16 cycles), the plane reads of the next column tile
      // in the last gaps
  #pragma unroll
      for (int i = 0; i < T1 * (NP == 3 ? 6 : 3); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 2 : 3, 0);  // VALU
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);  // DS read
      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
      for (int t = 0; t < T1; ++t) acc[1][t] = mx_products<NP>(af[1], bk[t], acc[1][t]);
  #pragma unroll
      for (int t = T1; t < NT; ++t) {
        bf16x8 bc[NP];
  #pragma unroll
        for (int p = 0; p < NP; ++p) bc[p] = bn[p];
        if (t + 1 < NT) {
  #pragma unroll
          for (int p = 0; p < NP; ++p) bn[p] = *(const bf16x8*)(sb + b_base + (t + 1) * 1024 + p * PLANE);
        }
  #pragma unroll
        for (int mw = 0; mw < kMxRowTiles; ++mw) acc[mw][t] = mx_products<NP>(af[mw], bc, acc[mw][t]);
      }
    }
    abuf = abuf + 1 == kMxASlots ? 0 : abuf + 1;
    bbuf = bbuf + 1 == kMxBSlots ? 0 : bbuf + 1;
  }
  const int64_t limit = TN ? g.r_cols : g.r_rows;
#pragma unroll
  for (int mw = 0; mw < kMxRowTiles; ++mw) mx_store<NT, NP>(g, acc[mw], outer_first + 32 * wave + 16 * mw, limit, lane);
}

}  // namespace k
}  // namespace corrla
