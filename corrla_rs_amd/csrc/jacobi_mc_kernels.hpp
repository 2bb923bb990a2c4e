// Multi-workgroup block Jacobi SVD of the l x l core (random_svd.rs:89) -- takes the core SVD off a single CU.
//
// One-sided (Hestenes) Jacobi is a chain of ~l rounds per sweep whatever the parallelism (every round rotates l/2
// disjoint column pairs), so its time is rounds x time-per-round.  The single-workgroup ring kernel packs all l/2
// rotations of a round into one CU (9 waves at l = 138) and is VALU-issue bound at ~0.75 us per round; a round whose
// rotations are spread over a few CUs with ~1 wave per SIMD runs at the latency of its dependency chain instead
// (LDS read -> dot -> lane reduction -> rotation -> apply -> LDS write -> barrier, ~0.2 us).
//
// Layout: the columns of W (= C on entry) and V (= I) are cut into 2*NP blocks of b columns, in global memory
// (L2-resident).  A sweep is 2*NP - 1 OUTER STEPS; step s pairs the blocks by the round-robin tournament and
// workgroup w of NP owns block pair (P, Q) for the whole step: it loads the 2b columns of W and V into LDS, runs
//   cross rounds  r = 0 .. b-1 : processor i (16 lanes) rotates (P[i], Q[(i + r) mod b])     -- all b^2 cross pairs
//   within rounds (step 0 only): b - 1 tournament rounds inside P and inside Q, b/2 + b/2 rotations each,
// one barrier per round, and stores the columns back.  Workgroups never talk to each other inside a launch: the
// exchange of blocks between steps IS the kernel boundary (~2 us, the same price as an in-kernel hand-off on this
// chip: MI355X_MICROARCH.md, rows 'boundary' / 'handoff-flag'), so there is no spin, no flag and nothing that could
// hang.  Squared column norms are recomputed at the start of every step and updated analytically in between
// (a' = a - t g, b' = b + t g), so a round needs one dot product.
//
// Round 3: the rounds of a step are no longer one ring over the whole block (one workgroup barrier per round) but run
// in SUB-BLOCKS of four columns, one per wave, with a barrier only when the sub-blocks move between waves (see
// "wave-local schedule" in jmc_step_kernel; CORRLA_JMC_LOCAL=0 restores the ring).  Measured at l = 138 f32 (b = 24,
// six waves): 937 shader cycles per round against 1220, 13.5 us per step against 17.2; what is left is the VALU issue
// of ~85 instructions per wave and round (two of the four SIMDs carry two waves), not the barrier.
//
// Columns whose squared norm is below floor2 = l eps^2 (the core is pre-scaled to max |entry| in [1, 2)) are numerically
// zero: every rotation against a large column re-injects rounding noise of their own size into them, so they would
// never test orthogonal and the iteration would never end (a core with sigma_min / sigma_max below eps: 40 sweeps without
// converging); they are left alone, like xGESVJ leaves columns below its underflow-scaled threshold.
//
// Convergence: every step ORs "rotated" / "some |cos| > tol_early" into per-sweep words; the launches of sweep S
// return at once when sweep S-1 had no rotation above tol_early (quadratic convergence: that sweep was the last).
// The host enqueues a fixed number of sweeps and never synchronises; jmc_finish_kernel reports whether the
// iteration had converged (checked together with the Cholesky status records at the end of the call).
#pragma once
#include "hip_kernels.hpp"

namespace corrla {
namespace k {

constexpr int kJmcMaxSweeps = 40;
constexpr float kJmcFastCond = 16.f;   // W-only mode when max |diag| <= 16 min |diag| ...
constexpr float kJmcVerifyCond = 64.f; // ... and accepted when the computed sigma_max <= 64 sigma_min
struct JmcCtl {
  unsigned rot[kJmcMaxSweeps + 1];  // sweep s rotated something
  unsigned big[kJmcMaxSweeps + 1];  // sweep s saw a pair above tol_early
  int sexp;                         // power-of-two prescale exponent (sigma is scaled back on output)
  int bad;                          // non-finite input
  int with_v;                       // 1: the rotations are accumulated into V inside the sweeps (any conditioning);
                                    // 0: W only, the other factor is recovered afterwards as X^-T-free product
                                    //    (jmc_other_factor_kernel), exact to eps * cond(X): well-conditioned cores only
  unsigned long long clk, wall;     // shader-clock / 100 MHz ticks workgroup 0 spent in its round loops (diagnostic)
  unsigned long long rounds;        // rounds workgroup 0 ran
  unsigned long long t_load, t_norm, t_store, t_total, steps;  // 100 MHz ticks of workgroup 0's phases (diagnostic)
  unsigned long long t_max_total;   // sum over steps of the slowest workgroup's time
  unsigned long long t_first, t_last;  // scratch: min start / max end of the current step (per launch)
  unsigned long long t_span;        // sum over steps of (last end - first start) as the workgroups saw it
};

template <class T>
struct JmcVec;
template <>
struct JmcVec<float> {
  typedef float v2 __attribute__((ext_vector_type(2)));
};
template <>
struct JmcVec<double> {
  typedef double v2 __attribute__((ext_vector_type(2)));
};

template <int LANES>
__device__ __forceinline__ float jmc_sum(float x) {
  return group_sum<LANES>(x);
}
template <int LANES>
__device__ __forceinline__ double jmc_sum(double x) {
  static_assert(LANES == 16, "f64 processors are 16 lanes wide");
  auto dpp = [](double v, auto ctrl) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, decltype(ctrl)::value, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, decltype(ctrl)::value, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  };
  x += dpp(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  x += dpp(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  x += dpp(x, std::integral_constant<int, 0x141>{});  // row_half_mirror
  x += dpp(x, std::integral_constant<int, 0x140>{});  // row_mirror
  return x;
}

// rows per column image: NC chunk rows of `lanes` lanes x 2 elements
__host__ __device__ constexpr int jmc_rows(int nc, int lanes) { return nc * 2 * lanes; }
// LDS column pitch (elements).  16 lanes per processor: b64 reads (f32) are serviced per 32-lane half = two
// processors, whose 128-byte segments must fall in different halves of the 256-byte bank row -> pitch = 32 (mod 64);
// b128 reads (f64) are serviced in interleaved 16-lane groups that mix two processors -> their columns must be
// bank-aligned, pitch = 0 (mod 32).  8 lanes per processor (f32 only): a 32-lane half is four processors reading 64
// bytes each from four (mostly consecutive) columns -> pitch = 16 or 48 (mod 64).
__host__ __device__ constexpr int jmc_pitch(int nc, int esz, int lanes) {
  return lanes == 8 ? (jmc_rows(nc, 8) + ((nc % 2 == 0) ? 16 : 0))
                    : (esz == 4 ? (jmc_rows(nc, 16) + ((nc % 2 == 0) ? 32 : 0)) : jmc_rows(nc, 16));
}
__host__ __device__ constexpr size_t jmc_lds_bytes(int nc, int b, int esz, int lanes) {
  return (size_t)2 * (2 * b) * jmc_pitch(nc, esz, lanes) * esz + (size_t)2 * b * esz + 64;
}

// W <- 2^sexp * C (zero padded to rp x ncols_pad, rp = the LDS column pitch so that a block of columns is ONE
// contiguous byte range in global memory and in LDS), V <- I, ctl cleared.  One workgroup.
template <class T>
__global__ __launch_bounds__(1024) void jmc_init_kernel(const T* __restrict__ c, int64_t ldc, int l, T* w, T* v, int rp,
                                                        int ncols_pad, int force_v, JmcCtl* ctl) {
  __shared__ float red[16];
  __shared__ int sbad;
  const int tid = threadIdx.x;
  if (tid == 0) sbad = 0;
  float mx = 0.f;
  int bad = 0;
  // (loads in batches of four, clamped and unconditional: one load per loop iteration is one dependent L2 round trip per
  //  iteration -- this one-workgroup kernel was 20 us of round trips at l = 138)
  for (int idx0 = tid; idx0 < l * l; idx0 += 4 * 1024) {
    T xb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(idx0 + u * 1024, l * l - 1);
      const int j = idx / l, i = idx - j * l;
      xb[u] = c[(int64_t)j * ldc + i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (idx0 + u * 1024 < l * l) {
        const float x = (float)fabs(xb[u]);
        if (!(x < 3.0e38f)) bad = 1;
        mx = fmaxf(mx, x);
      }
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (bad) sbad = 1;
  mx = 0.f;
  for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[i]);
  __syncthreads();
  int sexp = 0;
  if (mx > 0.f && !sbad) {
    (void)frexpf(mx, &sexp);
    sexp = 1 - sexp;  // mx * 2^sexp in [1, 2)
  }
  const int total = rp * ncols_pad;
  for (int idx0 = tid; idx0 < total; idx0 += 4 * 1024) {
    T xb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(idx0 + u * 1024, total - 1);
      const int j = idx / rp, i = idx - j * rp;
      xb[u] = c[(int64_t)min(j, l - 1) * ldc + min(i, l - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = idx0 + u * 1024;
      if (idx < total) {
        const int j = idx / rp, i = idx - j * rp;
        const bool in = i < l && j < l;
        w[idx] = in ? (T)ldexp((double)xb[u], sexp) : (T)0;
        v[idx] = (in && i == j) ? (T)1 : (T)0;
      }
    }
  }
  for (int i = tid; i <= kJmcMaxSweeps; i += 1024) {
    ctl->rot[i] = 0;
    ctl->big[i] = 0;
  }
  // conditioning estimate: the core is a triangular factor (up to rounding), whose diagonal brackets its singular
  // values well enough to choose the mode; the finish kernel verifies the choice against the computed sigma
  float dmin = 3.0e38f, dmax = 0.f;
  for (int i = tid; i < l; i += 1024) {
    const float d = (float)fabs(c[(int64_t)i * ldc + i]);
    dmin = fminf(dmin, d);
    dmax = fmaxf(dmax, d);
  }
  for (int off = 32; off > 0; off >>= 1) {
    dmin = fminf(dmin, __shfl_xor(dmin, off, 64));
    dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
  }
  __shared__ float rmin[16], rmax[16];
  if ((tid & 63) == 0) {
    rmin[tid >> 6] = dmin;
    rmax[tid >> 6] = dmax;
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 0; i < 16; ++i) {
      dmin = fminf(dmin, rmin[i]);
      dmax = fmaxf(dmax, rmax[i]);
    }
    ctl->with_v = (force_v || !(dmin * kJmcFastCond >= dmax) || !(dmax > 0.f)) ? 1 : 0;
    ctl->sexp = sexp;
    ctl->bad = sbad;
    ctl->clk = 0;
    ctl->wall = 0;
    ctl->rounds = 0;
    ctl->t_load = ctl->t_norm = ctl->t_store = ctl->t_total = ctl->steps = 0;
    ctl->t_max_total = ctl->t_span = 0;
    ctl->t_first = ~0ull;
    ctl->t_last = 0;
  }
}

// jacobi_rotation without its early exits: the same rotation arithmetic in a straight line, (cs, sn, t) = (1, 0, 0) when
// the pair is left alone (a column below floor2, |cos| <= tol, or the overflow guard) -- applying that "rotation" is exact.
// The rounds are bound by VALU issue, so the threshold tests are formed without the two reciprocal square roots of
// |g| / sqrt(a b): g^2 > tol^2 a b.  (a, b > floor2 = l eps^2 and the core is pre-scaled to max |entry| in [1, 2), so
// neither product leaves the normal range.)  big: the pair also exceeds tol_early.
__device__ __forceinline__ bool jmc_rotation_flat(float a, float b, float g, float tol, float tol_early, float floor2,
                                                  float& cs, float& sn, float& t, bool& big) {
  const float g2 = g * g, ab = a * b;
  const float zeta = (b - a) * 0.5f * __builtin_amdgcn_rcpf(g);
  const float den = fabsf(zeta) + __builtin_amdgcn_sqrtf(1.f + zeta * zeta);
  const float t_ = copysignf(__builtin_amdgcn_rcpf(den), zeta);
  const float c_ = __builtin_amdgcn_rsqf(1.f + t_ * t_);
  const float s_ = c_ * t_;
  const bool ok = (a > floor2) && (b > floor2) && (g2 > (tol * tol) * ab) && (fabsf(s_) <= 1.f) && (c_ <= 1.f);
  big = ok && (g2 > (tol_early * tol_early) * ab);
  cs = ok ? c_ : 1.f;
  sn = ok ? s_ : 0.f;
  t = ok ? t_ : 0.f;
  return ok;
}
__device__ __forceinline__ bool jmc_rotation_flat(double a, double b, double g, double tol, double tol_early, double floor2,
                                                  double& cs, double& sn, double& t, bool& big) {
  const double g2 = g * g, ab = a * b;
  const double zeta = (b - a) * 0.5 * jr_rcp(g);
  const double w = 1.0 + zeta * zeta;
  const double den = fabs(zeta) + w * jr_rsq(w);
  const double t_ = copysign(jr_rcp(den), zeta);
  const double c_ = jr_rsq(1.0 + t_ * t_);
  const double s_ = c_ * t_;
  const bool ok = (a > floor2) && (b > floor2) && (g2 > (tol * tol) * ab) && (fabs(s_) <= 1.0) && (c_ <= 1.0);
  big = ok && (g2 > (tol_early * tol_early) * ab);
  cs = ok ? c_ : 1.0;
  sn = ok ? s_ : 0.0;
  t = ok ? t_ : 0.0;
  return ok;
}

// One outer step.  grid = NP workgroups, block = b * LANES threads (rounded up to a wave; b <= 32, b even).
// LANES = 16 is what runs; 8 (f32 only: half the waves per block pair, twice the column per lane) measured slower.
template <class T, int NC, int LANES>
__global__ __launch_bounds__(512) void jmc_step_kernel(T* w, T* v, int b, int nblocks, int step, int sweep, int within,
                                                        T tol, T tol_early, T floor2, JmcCtl* ctl, int local) {
  typedef typename JmcVec<T>::v2 v2;
  constexpr int PITCH = jmc_pitch(NC, (int)sizeof(T), LANES);
  // The control words this launch depends on are loaded TOGETHER, unconditionally (clamped index): tested one after the
  // other, each sat in a branch of its own behind an s_waitcnt vmcnt(0) -- three to four dependent L2 round trips at the
  // head of every one of the ~45 launches of a call.
  const int sp = sweep > 0 ? sweep - 1 : 0;
  const unsigned rot_prev = ctl->rot[sp], big_prev = ctl->big[sp];
  const int bad_in = ctl->bad, with_v_in = ctl->with_v;
  // converged in an earlier sweep (the flags were written by earlier launches): nothing to do
  if (sweep > 0 && !(rot_prev && big_prev)) return;
  if (bad_in) return;  // non-finite input: the finish kernel reports it
  if (threadIdx.x == 0 && blockIdx.x == 0 && ctl->t_last != 0) {  // diagnostic: span of the PREVIOUS launch
    ctl->t_span += ctl->t_last - ctl->t_first;
    ctl->t_first = ~0ull;
    ctl->t_last = 0;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* wl = (T*)smem;                        // [2b][PITCH]
  T* vl = wl + (size_t)2 * b * PITCH;      // [2b][PITCH]
  T* nrm = vl + (size_t)2 * b * PITCH;     // [2b] squared column norms
  int* flag = (int*)(nrm + 2 * b);         // [2]; everything in the dynamic region: a static __shared__ would shift
                                           // its base off 16 bytes (cdna_hip_programming.md Guideline 17)
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int proc = tid / LANES, g = tid % LANES;
  const bool act = proc < b;
  int bp, bq;
  tournament_pair(nblocks, step, blockIdx.x, bp, bq);
  if (tid < 2) flag[tid] = 0;
  const bool with_v = with_v_in != 0;  // uniform over the grid
  const long long ts0 = wall_clock64();
  // ---- load the 2b columns of W and V: four contiguous segments (W_P, W_Q, V_P, V_Q), by LDS-DMA ----
  // (a register-staged copy loop serialises one L2 round trip per iteration: 18 of them at l = 266 f64 cost
  // ~25 us per step; the DMA form issues every 1-KiB piece back to back and waits once)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (nthr + 63) >> 6;
  const int seg_bytes = b * PITCH * (int)sizeof(T);
  {
    const char* src[4] = {(const char*)(w + (int64_t)bp * b * PITCH), (const char*)(w + (int64_t)bq * b * PITCH),
                          (const char*)(v + (int64_t)bp * b * PITCH), (const char*)(v + (int64_t)bq * b * PITCH)};
    char* dst[4] = {(char*)wl, (char*)(wl + (size_t)b * PITCH), (char*)vl, (char*)(vl + (size_t)b * PITCH)};
    const int nch = (seg_bytes + 1023) >> 10;
    const int nseg = with_v ? 4 : 2;
#pragma unroll
    for (int sgi = 0; sgi < 4; ++sgi)
      if (sgi < nseg)
      for (int c = wave; c < nch; c += nwaves) {
        const int off = c * 1024 + lane * 16;
        if (off < seg_bytes) glds16(src[sgi] + off, dst[sgi] + c * 1024);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const long long ts1 = wall_clock64();
  // per-lane chunk offsets (elements): chunk c of lane g covers rows (c * LANES + g) * 2 + {0, 1}
  const int lane_off = g * 2;
  // ---- squared norms of the local columns: processor i takes slots i and b + i ----
  if (act) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = proc + h * b;
      const T* col = wl + (size_t)s * PITCH + lane_off;
      T a0 = 0, a1 = 0;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const v2 x = *(const v2*)(col + c * 2 * LANES);
        a0 += x[0] * x[0];
        a1 += x[1] * x[1];
      }
      const T a = jmc_sum<LANES>(a0 + a1);
      if (g == 0) nrm[s] = a;
    }
  }
  __syncthreads();
  int my_rot = 0, my_big = 0;
  auto round = [&](int sx, int sy) {
    T* cx = wl + (size_t)sx * PITCH + lane_off;
    T* cy = wl + (size_t)sy * PITCH + lane_off;
    T* ux = vl + (size_t)sx * PITCH + lane_off;
    T* uy = vl + (size_t)sy * PITCH + lane_off;
    v2 x[NC], y[NC], vx[NC], vy[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      x[c] = *(const v2*)(cx + c * 2 * LANES);
      y[c] = *(const v2*)(cy + c * 2 * LANES);
    }
    if (with_v) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        vx[c] = *(const v2*)(ux + c * 2 * LANES);
        vy[c] = *(const v2*)(uy + c * 2 * LANES);
      }
    }
    T na = nrm[sx], nb = nrm[sy];
    v2 acc = x[0] * y[0];
#pragma unroll
    for (int c = 1; c < NC; ++c) acc += x[c] * y[c];
    const T gg = jmc_sum<LANES>(acc[0] + acc[1]);
    T cs, sn, rel, t;
    if (na > floor2 && nb > floor2 && jacobi_rotation(na, nb, gg, tol, cs, sn, rel, t)) {  // uniform over the 16 lanes of a processor
      my_rot = 1;
      if (rel > tol_early) my_big = 1;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        *(v2*)(cx + c * 2 * LANES) = cs * x[c] - sn * y[c];
        *(v2*)(cy + c * 2 * LANES) = sn * x[c] + cs * y[c];
      }
      if (with_v) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          *(v2*)(ux + c * 2 * LANES) = cs * vx[c] - sn * vy[c];
          *(v2*)(uy + c * 2 * LANES) = sn * vx[c] + cs * vy[c];
        }
      }
      if (g == 0) {
        nrm[sx] = na - t * gg;
        nrm[sy] = nb + t * gg;
      }
    }
  };
  const long long clk0 = clock64(), wall0 = wall_clock64();
  const long long ts2 = wall0;
  // ---- wave-local schedule (local != 0, LANES == 16: a wave = four processors) ----------------------------------
  // The ring schedule below hands every Q column to the next processor once per round, across waves: one workgroup
  // barrier per round (1220 cycles per round at l = 138 f32; this schedule: 922 -- what is left is the issue of the
  // ~85 instructions of a round with two waves on half of the SIMDs, profiles/r03_jacobi_schedules.txt).
  // Here the blocks are cut into SUB-BLOCKS of four columns, one per wave, and a wave meets a sub-block on its own:
  //   inner4: processor i keeps A[i] in registers and rotates it against B[(i + r) mod 4], r = 0..3 -- all 16 pairs of
  //           (A, B); the four B columns cross LDS, but they are touched by THIS wave only, and the LDS operations of
  //           one wave complete in issue order: no barrier, only the data dependence.
  //   cross : super-round s pairs wave k's sub-block of P with sub-block (k + s) mod NSB of Q; barrier per super-round
  //           (NSB = b / 4 barriers instead of b).
  //   within: a round-robin tournament over the sub-blocks of P (first half of the waves) and of Q (second half),
  //           one inner4 per meeting, then the three rounds inside every sub-block (two processors per sub-block).
  // Every pair of columns still meets exactly once per sweep; only the order differs.
  auto inner4 = [&](auto wv, v2 (&x)[NC], v2 (&vx)[NC], T& na, int sb0) {
    constexpr bool WV = decltype(wv)::value;  // V accumulated: the caller branches once, the rounds are straight lines
    const int pi = proc & 3;
    // b is a multiple of four here (the host's geometry): every processor has a column and every sub-block is full, so a
    // round has no predicate at all.  The rotation is computed whether or not the pair needs one and applied as
    // (cs, sn) = (1, 0) when it does not (jmc_rotation_flat); the loads of round r + 1 are issued right behind the stores
    // of round r, ahead of the register half of the update (two register images of the B column, alternating).
    constexpr bool DB = !(WV && sizeof(T) == 8 && NC > 4);  // f64 with V at l > 128: two images of B would spill
    constexpr int NB = DB ? 2 : 1;
    v2 y[NB][NC];
    T nb[NB];
    auto fetch = [&](int r, int buf) {
      const int sy = sb0 + ((pi + r) & 3);
      const T* cy = wl + (size_t)sy * PITCH + lane_off;
      nb[buf] = nrm[sy];
#pragma unroll
      for (int c = 0; c < NC; ++c) y[buf][c] = *(const v2*)(cy + c * 2 * LANES);
    };
    fetch(0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cur = DB ? (r & 1) : 0;
      const int sy = sb0 + ((pi + r) & 3);
      T* cy = wl + (size_t)sy * PITCH + lane_off;
      T* uy = vl + (size_t)sy * PITCH + lane_off;
      v2 acc = x[0] * y[cur][0], acc2 = {0, 0};  // two chains: the dot product sits on the critical path of the round
      if (NC > 1) acc2 = x[1] * y[cur][1];
#pragma unroll
      for (int c = 2; c < NC; c += 2) {
        acc += x[c] * y[cur][c];
        if (c + 1 < NC) acc2 += x[c + 1] * y[cur][c + 1];
      }
      acc += acc2;
      const T gg = jmc_sum<LANES>(acc[0] + acc[1]);
      T cs, sn, t;
      bool big;
      const bool rot = jmc_rotation_flat(na, nb[cur], gg, tol, tol_early, floor2, cs, sn, t, big);
      my_rot |= rot ? 1 : 0;
      my_big |= big ? 1 : 0;
      // late sweeps: most rounds rotate nothing -- a wave none of whose four pairs rotates skips the update (uniform)
      const bool any = __builtin_amdgcn_ballot_w64(rot) != 0;
      if (any) {
#pragma unroll
        for (int c = 0; c < NC; ++c) *(v2*)(cy + c * 2 * LANES) = sn * x[c] + cs * y[cur][c];
        if (g == 0) nrm[sy] = nb[cur] + t * gg;
      }
      // the next inner round reads what other lanes of this wave have just written
      asm volatile("" ::: "memory");
      if (DB && r < 3) fetch(r + 1, cur ^ 1);
      if (any) {
      if (WV) {
        // the V columns only receive the rotation: they are read here, behind the W traffic the next round waits for
        v2 vy[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) vy[c] = *(const v2*)(uy + c * 2 * LANES);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          *(v2*)(uy + c * 2 * LANES) = sn * vx[c] + cs * vy[c];
          vx[c] = cs * vx[c] - sn * vy[c];
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) x[c] = cs * x[c] - sn * y[cur][c];
      na -= t * gg;
      }
      if (!DB && r < 3) fetch(r + 1, 0);
    }
  };
  auto load_col = [&](auto wv, int slot, v2 (&x)[NC], v2 (&vx)[NC], T& na) {
    const T* cx = wl + (size_t)slot * PITCH + lane_off;
    const T* ux = vl + (size_t)slot * PITCH + lane_off;
#pragma unroll
    for (int c = 0; c < NC; ++c) x[c] = *(const v2*)(cx + c * 2 * LANES);
    if (decltype(wv)::value) {
#pragma unroll
      for (int c = 0; c < NC; ++c) vx[c] = *(const v2*)(ux + c * 2 * LANES);
    }
    na = nrm[slot];
  };
  auto store_col = [&](auto wv, int slot, const v2 (&x)[NC], const v2 (&vx)[NC], T na) {
    T* cx = wl + (size_t)slot * PITCH + lane_off;
    T* ux = vl + (size_t)slot * PITCH + lane_off;
#pragma unroll
    for (int c = 0; c < NC; ++c) *(v2*)(cx + c * 2 * LANES) = x[c];
    if (decltype(wv)::value) {
#pragma unroll
      for (int c = 0; c < NC; ++c) *(v2*)(ux + c * 2 * LANES) = vx[c];
    }
    if (g == 0) nrm[slot] = na;
  };
  // the whole schedule once per mode (wv: V accumulated), so that the W-only instance holds no V registers at all
  auto run_local = [&](auto wv) {
    const int nsb = b >> 2;  // sub-blocks per block = waves of the workgroup
    const int pi = proc & 3;
    v2 x[NC], vx[NC];
    T na = 0;
    {
      load_col(wv, proc, x, vx, na);
      for (int s = 0; s < nsb; ++s) {
        int m = wave + s;
        if (m >= nsb) m -= nsb;
        inner4(wv, x, vx, na, b + 4 * m);
        __syncthreads();
      }
      store_col(wv, proc, x, vx, na);
      __syncthreads();
    }
    if (within) {
      const int n_even = nsb + (nsb & 1), halfn = n_even >> 1;
      const int real = (nsb & 1) ? halfn - 1 : halfn;  // meetings per block and round (one pair holds the dummy when nsb is odd)
      const int side = real > 0 ? wave / real : 2, jj = real > 0 ? wave - side * real : 0;
      for (int t_ = 0; t_ < n_even - 1; ++t_) {
        if (side < 2) {
          int pa = -1, pb = -1, cnt = 0;
          for (int j = 0; j < halfn; ++j) {
            int p_, q_;
            tournament_pair(n_even, t_, j, p_, q_);
            if (q_ < nsb) {  // p_ < q_
              if (cnt == jj) {
                pa = p_;
                pb = q_;
              }
              ++cnt;
            }
          }
          if (pa >= 0) {
            const int sa = side * b + 4 * pa + pi;
            load_col(wv, sa, x, vx, na);
            inner4(wv, x, vx, na, side * b + 4 * pb);
            store_col(wv, sa, x, vx, na);
          }
        }
        __syncthreads();
      }
      // inside the sub-blocks: processors 0, 1 of wave k take sub-block k of P, processors 2, 3 that of Q
      {
        const int base = (pi >> 1) * b + 4 * wave;
#pragma unroll
        for (int t_ = 0; t_ < 3; ++t_) {
          int p_, q_;
          tournament_pair(4, t_, pi & 1, p_, q_);
          round(base + p_, base + q_);
          asm volatile("" ::: "memory");
        }
      }
      __syncthreads();
    }
  };
  if (local && LANES == 16 && (b & 3) == 0) {
    if (with_v)
      run_local(std::true_type{});
    else
      run_local(std::false_type{});
  } else {
  // ---- cross rounds: every column of P meets every column of Q ----
  // Processor i keeps column P[i] (and its V column, and its squared norm) in registers for all b rounds: only the Q
  // column of a round crosses LDS.
  {
    T* cx = wl + (size_t)proc * PITCH + lane_off;
    T* ux = vl + (size_t)proc * PITCH + lane_off;
    v2 x[NC], vx[NC];
    T na = 0;
    if (act) {
#pragma unroll
      for (int c = 0; c < NC; ++c) x[c] = *(const v2*)(cx + c * 2 * LANES);
      if (with_v) {
#pragma unroll
        for (int c = 0; c < NC; ++c) vx[c] = *(const v2*)(ux + c * 2 * LANES);
      }
      na = nrm[proc];
    }
    for (int r = 0; r < b; ++r) {
      if (act) {
        int sy = proc + r;
        if (sy >= b) sy -= b;
        sy += b;
        T* cy = wl + (size_t)sy * PITCH + lane_off;
        T* uy = vl + (size_t)sy * PITCH + lane_off;
        v2 y[NC], vy[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) y[c] = *(const v2*)(cy + c * 2 * LANES);
        if (with_v) {
#pragma unroll
          for (int c = 0; c < NC; ++c) vy[c] = *(const v2*)(uy + c * 2 * LANES);
        }
        const T nb = nrm[sy];
        v2 acc = x[0] * y[0];
#pragma unroll
        for (int c = 1; c < NC; ++c) acc += x[c] * y[c];
        const T gg = jmc_sum<LANES>(acc[0] + acc[1]);
        T cs, sn, rel, t;
        if (na > floor2 && nb > floor2 && jacobi_rotation(na, nb, gg, tol, cs, sn, rel, t)) {  // uniform over the lanes of a processor
          my_rot = 1;
          if (rel > tol_early) my_big = 1;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const v2 xn = cs * x[c] - sn * y[c];
            *(v2*)(cy + c * 2 * LANES) = sn * x[c] + cs * y[c];
            x[c] = xn;
          }
          if (with_v) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const v2 xn = cs * vx[c] - sn * vy[c];
              *(v2*)(uy + c * 2 * LANES) = sn * vx[c] + cs * vy[c];
              vx[c] = xn;
            }
          }
          na -= t * gg;
          if (g == 0) nrm[sy] = nb + t * gg;
        }
      }
      __syncthreads();
    }
    if (act) {  // the columns go back to the LDS image (stored to global memory from there)
#pragma unroll
      for (int c = 0; c < NC; ++c) *(v2*)(cx + c * 2 * LANES) = x[c];
      if (with_v) {
#pragma unroll
        for (int c = 0; c < NC; ++c) *(v2*)(ux + c * 2 * LANES) = vx[c];
      }
      if (g == 0) nrm[proc] = na;
    }
    __syncthreads();
  }
  // ---- within rounds (once per sweep): tournaments inside P and inside Q, b even ----
  if (within) {
    // the norms drifted by at most b analytic updates; the pairs inside a block start from fresh ones
    const int half = b >> 1;
    for (int r = 0; r < b - 1; ++r) {
      if (act) {
        const int blk = proc >= half ? 1 : 0;
        const int j = proc - blk * half;
        int p, q;
        tournament_pair(b, r, j, p, q);
        round(blk * b + p, blk * b + q);
      }
      __syncthreads();
    }
  }
  }  // ring schedule
  if (my_rot) flag[0] = 1;
  if (my_big) flag[1] = 1;
  const long long ts3 = wall_clock64();
  const long long clk3 = clock64();
  // ---- store the columns back (batches of 8 vectors per thread in flight) ----
  {
    typedef typename MT<T>::vec_t vec_t;
    const int seg_vecs = seg_bytes / 16;
    vec_t* gdst[4] = {(vec_t*)(w + (int64_t)bp * b * PITCH), (vec_t*)(w + (int64_t)bq * b * PITCH),
                      (vec_t*)(v + (int64_t)bp * b * PITCH), (vec_t*)(v + (int64_t)bq * b * PITCH)};
    const vec_t* lsrc[4] = {(const vec_t*)wl, (const vec_t*)(wl + (size_t)b * PITCH), (const vec_t*)vl,
                            (const vec_t*)(vl + (size_t)b * PITCH)};
    const int nseg = with_v ? 4 : 2;
#pragma unroll
    for (int sgi = 0; sgi < 4; ++sgi)
      if (sgi < nseg)
      for (int base = tid; base < seg_vecs; base += 8 * nthr) {
        vec_t tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * nthr;
          if (idx < seg_vecs) tmp[u] = lsrc[sgi][idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * nthr;
          if (idx < seg_vecs) gdst[sgi][idx] = tmp[u];
        }
      }
  }
  __syncthreads();
  if (tid == 0) {
    if (flag[0]) atomicOr(&ctl->rot[sweep], 1u);
    if (flag[1]) atomicOr(&ctl->big[sweep], 1u);
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      atomicMin(&ctl->t_first, (unsigned long long)ts0);
      atomicMax(&ctl->t_last, (unsigned long long)wall_clock64());
    }
    if (blockIdx.x == 0) {  // diagnostics: only this thread ever touches these words
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the column stores of this wave have left
      const long long ts4 = wall_clock64();
      ctl->clk += (unsigned long long)(clk3 - clk0);
      ctl->wall += (unsigned long long)(ts3 - wall0);
      ctl->rounds += (unsigned long long)(b + (within ? b - 1 : 0));
      ctl->t_load += (unsigned long long)(ts1 - ts0);
      ctl->t_norm += (unsigned long long)(ts2 - ts1);
      ctl->t_store += (unsigned long long)(ts4 - ts3);
      ctl->t_total += (unsigned long long)(ts4 - ts0);
      ctl->steps += 1;
    }
  }
}

// sigma_j = ||w_j||, descending order, outputs (m1 <- V[:, order[:k]], m2 <- W[:, order[:k]] / sigma), and the
// convergence verdict: st->fail = 0 when some enqueued sweep ended the iteration, 1 otherwise (3: non-finite input).
template <class T>
__global__ __launch_bounds__(1024) void jmc_finish_kernel(const T* w, const T* v, int rp, int l, int nsweeps,
                                                          const JmcCtl* ctl, T* m1, int64_t ld1, T* m2, int64_t ld2,
                                                          T* s_out, int k, CholStatus* st) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sigma = (T*)smem;
  int* order = (int*)(sigma + l + 2);
  const int tid = threadIdx.x, group = tid >> 4, gl = tid & 15;
  for (int j = group; j < l; j += 64) {
    T a = 0;
    // (six loads in flight, accumulated in the order of the plain loop)
    for (int i0 = gl; i0 < l; i0 += 6 * 16) {
      T xb[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) xb[u] = w[(int64_t)j * rp + min(i0 + 16 * u, l - 1)];
#pragma unroll
      for (int u = 0; u < 6; ++u)
        if (i0 + 16 * u < l) a += xb[u] * xb[u];
    }
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) a += __shfl_xor(a, msk, 16);
    // a non-finite column (non-finite input: ctl->bad) must not reach the ranking below -- comparisons with NaN are all
    // false, the ranks would collide and `order` would keep uninitialised LDS, i.e. wild column indices
    if (gl == 0) sigma[j] = jacobi_safe_sigma(a);
  }
  __syncthreads();
  for (int j = tid; j < l; j += 1024) {
    const T sj = sigma[j];
    int r = 0;
    for (int i = 0; i < l; ++i) {
      const T si = sigma[i];
      r += (si > sj || (si == sj && i < j)) ? 1 : 0;
    }
    order[r] = j;
  }
  __syncthreads();
  const int sexp = ctl->sexp;
  const bool with_v = ctl->with_v != 0;
  for (int r = group; r < k; r += 64) {
    const int j = order[r];
    const T sj = sigma[j];
    const T inv = sj > (T)0 ? (T)1 / sj : (T)0;
    for (int i0 = gl; i0 < l; i0 += 6 * 16) {
      T wb[6], vb[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = min(i0 + 16 * u, l - 1);
        wb[u] = w[(int64_t)j * rp + i];
        vb[u] = with_v ? v[(int64_t)j * rp + i] : (T)0;
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int i = i0 + 16 * u;
        if (i < l) {
          m2[(int64_t)r * ld2 + i] = wb[u] * inv;
          if (with_v) m1[(int64_t)r * ld1 + i] = vb[u];
        }
      }
    }
    if (gl == 0) s_out[r] = (T)ldexp((double)sj, -sexp);
  }
  if (tid == 0 && st) {
    int conv = 0, used = nsweeps;
    for (int s = nsweeps - 1; s >= 0; --s)
      if (!(ctl->rot[s] && ctl->big[s])) {
        conv = 1;
        used = s + 1;  // sweeps that did work: the first one without a rotation above tol_early, and those before it
      }
    // W-only mode is only valid for a well-conditioned core: verify the estimate that chose it against sigma
    const bool wonly_bad = !with_v && !((float)sigma[order[k - 1]] * kJmcVerifyCond >= (float)sigma[order[0]]);
    // fail: 0 converged, 1 ran out of sweeps, 3 non-finite input, 4 the W-only shortcut was not valid (repeat with V)
    st->fail = ctl->bad ? 3 : (wonly_bad ? 4 : (conv ? 0 : 1));
    st->min_ratio = 1.f;
    st->dev_i = (float)used;  // read back as the hint for how many sweeps the next call of this context enqueues
    st->gmax = 1.f;
    st->clk = 0;
    st->wall = 0;
  }
}

// W-only mode: X = U_X S V_X^T with U_X = W / sigma known (m2) -> V_X = X^T U_X S^-1, i.e.
//   m1[i, r] = sum_j X[j, i] * m2[j, r] / sigma_r      (column i of X dotted with column r of m2)
// exact to eps * cond(X), which the mode selection bounds by kJmcVerifyCond.  One 16-lane group per output.
template <class T>
__global__ __launch_bounds__(256) void jmc_other_factor_kernel(const T* __restrict__ x, int64_t ldx, int l, int k,
                                                              const T* __restrict__ m2, int64_t ld2, const T* __restrict__ s_out,
                                                              const JmcCtl* ctl, T* m1, int64_t ld1) {
  if (ctl->with_v) return;
  const int gid = (int)((blockIdx.x * (unsigned)blockDim.x + threadIdx.x) >> 4), gl = threadIdx.x & 15;
  if (gid >= l * k) return;
  const int r = gid / l, i = gid - r * l;
  const T* xc = x + (int64_t)i * ldx;
  const T* uc = m2 + (int64_t)r * ld2;
  T a = 0;
  for (int j = gl; j < l; j += 16) a += xc[j] * uc[j];
#pragma unroll
  for (int msk = 1; msk < 16; msk <<= 1) a += __shfl_xor(a, msk, 16);
  // s_out carries the unscaled sigma while x is the unscaled core: consistent
  const T sg = s_out[r];
  if (gl == 0) m1[(int64_t)r * ld1 + i] = sg > (T)0 ? a / sg : (T)0;
}

}  // namespace k
}  // namespace corrla
