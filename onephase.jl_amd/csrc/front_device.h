// Device code shared by the per-step kernels of numeric.hip and the dataflow kernel of dataflow.hip: pivot classification and
// counting, the stop flag of the delta loop, the diagonal-block factorisation (diag2_body), the row-parallel triangular
// solve below it (trsm_body) and the constants of the trailing update.
#pragma once
#include "numeric.h"

namespace okkt {

typedef double d4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void classify_pivot(double d, double tol, unsigned& pos, unsigned& neg,
                                                unsigned& zer, unsigned& bad) {
  // julia.jl:72-78: pos = d > tol, neg = d < -tol, zero = rest; NaN/Inf counted apart
  if (isnan(d) || isinf(d)) ++bad;
  else if (d > tol) ++pos;
  else if (d < -tol) ++neg;
  else ++zer;
}

// keep ? v : +0.0 as a bit mask.  A `cond ? load : 0` select lets hipcc sink the load under a branch
// and wait vmcnt(0) for every element (serial memory round trips); the AND keeps the load unconditional.
__device__ __forceinline__ double keep_f64(double v, bool keep) {
  return __longlong_as_double(__double_as_longlong(v) & (keep ? -1ll : 0ll));
}

// offset to add to F + c * f for column c of front s: zero for a pivot column, the front's shift into the shared contribution-block
// region otherwise (numeric.h, DevPlan::cb_shift)
__device__ __forceinline__ int64_t cb_off(const DevPlan& P, int s, int c, int k) { return c >= k ? P.cb_shift[s] : 0; }

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// With want_neg >= 0 (a retry of the delta loop) the running totals decide: more positive pivots than n, more negative
// than m, a zero or a non-finite pivot can never give the inertia (n, m, 0) -> raise the stop flag counters[4]; every
// factorisation kernel launched or started afterwards returns at once (stop_requested_*), the host reads the flag
// with the counts.  A kernel argument decides whether any of this runs: the first attempt pays nothing.
__device__ __forceinline__ void flush_counts(const DevPlan& P, int slot, unsigned pos, unsigned neg, unsigned zer, unsigned bad) {
  // the limits are tested on the slot's own running totals: a part of the true totals, so exceeding them is still proof
  unsigned long long* counters = P.counters + (size_t)slot * kCountStride;
  pos = wave_sum(pos); neg = wave_sum(neg); zer = wave_sum(zer); bad = wave_sum(bad);
  if ((threadIdx.x & 63) == 0) {
    if (P.want_pos < 0 && P.want_neg < 0) {
      // no limits to test: atomics without a return value (the returning form kept the wave waiting for the memory side's answer:
      // 2 us at the end of every diagonal block)
      if (pos) atomicAdd(&counters[0], (unsigned long long)pos);
      if (neg) atomicAdd(&counters[1], (unsigned long long)neg);
      if (zer) atomicAdd(&counters[2], (unsigned long long)zer);
      if (bad) atomicAdd(&counters[3], (unsigned long long)bad);
      return;
    }
    bool fail = false;
    if (pos) { const unsigned long long o = atomicAdd(&counters[0], (unsigned long long)pos); fail |= P.want_pos >= 0 && o + pos > (unsigned long long)P.want_pos; }
    if (neg) { const unsigned long long o = atomicAdd(&counters[1], (unsigned long long)neg); fail |= P.want_neg >= 0 && o + neg > (unsigned long long)P.want_neg; }
    if (zer) { atomicAdd(&counters[2], (unsigned long long)zer); fail |= P.want_neg >= 0; }
    if (bad) { atomicAdd(&counters[3], (unsigned long long)bad); fail |= P.want_neg >= 0; }
    if (fail) atomicExch(&P.counters[4], 1ull);
  }
}

// workgroup-uniform (one load, one barrier) / wave-uniform forms of "has the stop flag been raised?"
__device__ __forceinline__ bool stop_requested_wg(const DevPlan& P) {
  if (P.want_neg < 0) return false;
  __shared__ int s_stop;
  if (threadIdx.x == 0) s_stop = (int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return s_stop != 0;
}
__device__ __forceinline__ bool stop_requested_wave(const DevPlan& P) {
  if (P.want_neg < 0) return false;
  return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&P.counters[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
}

// 1/d from v_rcp_f64 and two Newton steps (about 1 ulp); d = 0 gives inf/NaN like the division would
__device__ __forceinline__ double fast_rcp_f64(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  // wave-uniform broadcast through SGPRs (v_readlane_b32 x 2); src_lane must be wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src_lane);
  hi = __builtin_amdgcn_readlane(hi, src_lane);
  return __hiloint2double(hi, lo);
}

// Bound of the waits inside the dataflow launches: wall-clock time, not a number of polls (advisor, round 4: a launch slowed far below
// normal -- counter collection, several processes on one GPU -- must not turn a legitimate wait into an error).  The time-out word
// counters[5] is raised alone: the host reports "a hand-off timed out", not a pivot-count failure.
constexpr long long kWaitTicks = 300000000LL;      // 3 s of the 100-MHz counter
__device__ __forceinline__ bool wait_expired(int& spins, long long& t0) {
  if ((++spins & 127) != 0) return false;
  const long long now = wall_clock64();
  if (t0 == 0) { t0 = now; return false; }
  return now - t0 > kWaitTicks;
}

constexpr int kIB = 32;  // inner block width of the diagonal-block kernel
constexpr int kTld = 33; // leading dimension of the 32 x 32 scratch blocks (odd: conflict-free column access)

#ifndef OKKT_DIAG_MW
#define OKKT_DIAG_MW 8
#endif
constexpr int kMW = OKKT_DIAG_MW;   // micro-panel width (4 or 8)
constexpr int kPLD = 144;     // leading dimension of the LDS panels: 16-lane groups of an MFMA fragment hit disjoint banks
constexpr int kXld = 33;

// k_big_diag2 (round 3): the same factorisation with the two kinds of work on different waves, software-pipelined.
//   waves 2, 3 hold ALL 36 accumulator tiles (18 each) and do every MFMA; waves 0, 1 are the 128 row threads.
//   Micro-step m:   waves 2, 3: rank-8 update of step m - 1 on the tile column that holds panel m, copy panel m out -> barrier
//                   waves 0, 1: 8 x 8 factor + row solve of panel m      ||      waves 2, 3: rest of the update of step m - 1 -> barrier
//   In k_big_diag the row threads and the MFMA updates alternate and half of the workgroup idles in each; here a micro-step
//   costs max(row work, rest of the previous update) + the short head.  -L / W panels are double-buffered in LDS.  The
//   arithmetic per entry is the same sequence of operations: bitwise the same factor.
// agent-scope (sc1) store: outputs that another workgroup of the SAME launch reads (fused diag + trsm launch below)
__device__ __forceinline__ void st_agent_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// LDS of diag2_body in doubles: raw micro-panel, two buffers each of -L and W, the four 32 x 32 diagonal blocks of L, their inverses, the pivots
#define OKKT_DIAG2_LDS_DOUBLES(MWc) ((size_t)5 * (MWc) * kPLD + (size_t)2 * 4 * 32 * kXld + 128 + 128)
#ifndef OKKT_DIAG2_MW
#define OKKT_DIAG2_MW 8
#endif
constexpr int kDiag2MW = OKKT_DIAG2_MW;     // columns per micro-step of diag2_body (8: round 3; 4 halves the redundant block factorisation of the row threads for twice the barriers)
// tile_lds != nullptr (dataflow.hip, TU + D in one task): the block comes from LDS (column-major, leading dimension tile_ld, written
// by the same workgroup) instead of the front in HBM; it overlaps this function's own LDS areas, hence the barrier behind the loads
// LPROG (round 6, the dataflow worker): the block reports its progress -- `sub_state` counts the 32-column blocks whose L entries and
// pivots are in memory -- so that the task behind it on the critical path (TU(q), dataflow.hip df_tu_lock) works on column block b while
// column block b + 1 is being factored.  The row waves wait for their own stores of column block b at the top of micro-step 4 b + 5,
// in front of the next deferred outputs: the stores are a whole micro-step old by then (and the wait falls into the MFMA waves' head,
// where the row waves idle), so the publication costs the loop nothing.  The inverses of the 32 x 32 diagonal blocks are NOT part of
// the report (they stay in the tail): the follower inverts the blocks itself.  (A form with the inversions inside the loop on a wave
// of their own and a storing wave -- four MFMA waves instead of six -- published complete blocks but took 55 us per diagonal block
// instead of 32: scripts/experiments/r06_diag2_prog_service_waves.patch.)
template <bool AG, int MW = kDiag2MW, int NMM = 4, bool LPROG = false>
__device__ __forceinline__ void diag2_body(const DevPlan& P, int s, int step, int NB, double tol, double* sm, const double* tile_lds = nullptr, int tile_ld = 0, long long* marks = nullptr,
                                           int* sub_state = nullptr) {
  static_assert(MW == 8 || MW == 4, "micro-panels of 4 or 8 columns");
  constexpr int NE = MW / 4;                   // MFMA k-steps (4 columns each) per micro-panel
  constexpr int PER = 16 / MW;                 // micro-panels per 16-column tile
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // opaque: inside a persistent loop (dataflow.hip) nothing derived from the lane id is hoisted out of the body and kept live across the other roles
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = step * NB;
  if (j0 >= k) return;
  const int nb = min(NB, k - j0);
  double* Praw = sm;                           // 8 x kPLD: raw micro-panel (columns as rows of the array)
  double* LpB = Praw + MW * kPLD;             // -L of the micro-panel, two buffers
  double* WpB = LpB + 2 * MW * kPLD;          // W = L * D, two buffers
  double* Ld = WpB + 2 * MW * kPLD;           // 4 diagonal 32 x 32 blocks of L, leading dimension 33
  double* Xs = Ld + 4 * 32 * kXld;             // their inverses
  double* dpiv = Xs + 4 * 32 * kXld;           // the 128 pivots, for the waves that count them behind the loop
  // The 8 x 8 diagonal block of a micro-step is factored by EVERY row thread (188 of the 460 instructions of its micro-step).  Doing it
  // once was tried twice and is slower both ways: handed on column by column through LDS flags (round 4: 50.6 us per diagonal block
  // against 34.5; scripts/experiments/r04_diag2_pipe.patch) and as a whole behind one flag (round 5: wave 0 factors, wave 1 waits and
  // solves with 36 multiply-adds -- 71 us between two diagonal blocks against 63; r05_diag2_factor_once.patch).  The factorisation is a
  // dependent chain of eight reciprocals (about 1 us on a wave that has nothing else to issue); a row thread that repeats it interleaves
  // the chain with its own row's independent work, which is what hides it.  The multipliers as scalar operands through v_readlane
  // from copies of the diagonal rows in the finished lanes of both row waves (r05_diag2_readlane.patch: 72 readlanes instead of the 188
  // instructions) is bitwise equal and time-neutral: the row solve is 0.41 us of a 1.8 us micro-step, the rest is two barriers and
  // four LDS round trips.
  double* F = P.arena + P.front_pos[s];
  // MFMA waves: NMM of them from wave 2 on.  Four in a 384-thread workgroup (k_big_diag2: one per SIMD, the two row waves share two of the
  // SIMDs); six in the 512-thread workers of the dataflow launch, whose waves 6 and 7 would otherwise only meet the barriers -- a wave
  // gets a v_mfma_f64_16x16x4 through at the same rate whether it is alone on its SIMD or shares it with one more, so six waves finish
  // the 36 tiles in two thirds of the time (the rest of the update, not the row phase, was what a micro-step waited for: 9 of 38 us).
  // (Waves 2, 3, 6, 7 -- two SIMDs with two MFMA waves each, the row waves' SIMDs left alone -- was slower than waves 2 .. 5: 39.9 us.)
  static_assert(NMM == 4 || NMM == 6, "36 tiles over four or six MFMA waves");
  constexpr int NT = 36 / NMM;                 // tiles per MFMA wave
  const bool mm = wave >= 2 && wave < 2 + NMM;
  const int mmi = wave >= 2 ? wave - 2 : 0;    // 0 .. NMM - 1 among the MFMA waves
  // tiles of an MFMA wave: t = 4 q + (wave - 2), column-major over the lower triangle of the 8 x 8 tile grid
  int ti_s[NT], tj_s[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int t = NMM * q + mmi;
    const int tj = (t >= 8) + (t >= 15) + (t >= 21) + (t >= 26) + (t >= 30) + (t >= 33) + (t >= 35);
    const int start = tj * 8 - tj * (tj - 1) / 2;
    tj_s[q] = tj;
    ti_s[q] = tj + (t - start);
  }
  d4_t acc[NT];
  if (mm) {
    double raw[NT][4];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int r = 16 * ti_s[q] + l15;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c = 16 * tj_s[q] + 4 * v + l4;
        raw[q][v] = tile_lds ? tile_lds[(size_t)min(c, nb - 1) * tile_ld + min(r, nb - 1)] : F[(size_t)(j0 + min(c, nb - 1)) * f + j0 + min(r, nb - 1)];
      }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int r = 16 * ti_s[q] + l15;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c = 16 * tj_s[q] + 4 * v + l4;
        const double pad = r == c ? 1.0 : 0.0;
        acc[q][v] = (r < nb && c < nb && r >= c) ? raw[q][v] : pad;
      }
    }
  }
  if (tile_lds) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }      // every MFMA wave holds its tiles: the LDS below may be written
  double my_d = 1.0;
  const int nms = (nb + MW - 1) / MW;        // micro-steps
  // marks (task log of the dataflow launch, wave 0 = a row wave): time spent waiting for the head (barrier 1), in the row phase,
  // waiting for the rest of the update (barrier 2), and the start of the tail behind the loop
  long long tk = marks ? wall_clock64() : 0, t_head = 0, t_row = 0, t_rest = 0;
  // row threads: the multipliers and W values of the last micro-step, kept in registers until its outputs are written
  double lr_keep[MW], w_keep[MW];
  int p8_keep = -1;
  auto row_outputs = [&]() {
    // The entries below the diagonal are L; a diagonal row also stores its pivot at its own column (one store at a per-lane address
    // instead of a select per column in every row thread); the columns of a micro-step lie in ONE 32-column block (p8 is a multiple
    // of MW), so whether the row belongs to that diagonal 32 x 32 block is decided once.
    const int p8 = p8_keep;
    if (p8 < 0 || mm || tid < p8 || tid >= 128) return;
    const int r = tid;
    const int i = r - p8;
    double* Fr = F + (size_t)(j0 + p8) * f + j0 + r;
    const bool rv = r < nb;
    const bool inblk = (r >> 5) == (p8 >> 5);
    double* Ldr = Ld + (r >> 5) * 32 * kXld + (r & 31) + (p8 & 31) * kXld;
#pragma unroll
    for (int c = 0; c < MW; ++c) {
      if (i > c && rv && p8 + c < nb) {
        if (AG) st_agent_f64(&Fr[(size_t)c * f], lr_keep[c]); else Fr[(size_t)c * f] = lr_keep[c];
        if (inblk) Ldr[c * kXld] = lr_keep[c];
      }
    }
    if (i < MW) {
#pragma unroll
      for (int c = 0; c < MW; ++c) my_d = i == c ? w_keep[c] : my_d;
      dpiv[r] = my_d;
      if (rv) {
        if (AG) st_agent_f64(&Fr[(size_t)i * f], my_d); else Fr[(size_t)i * f] = my_d;
        if constexpr (LPROG) st_agent_f64(&P.dvals[col0 + j0 + r], my_d);      // the follower needs the pivots with the block, not behind the loop
        Ldr[i * kXld] = my_d;
      }
    }
  };
  for (int ms = 0; ms < nms; ++ms) {
    const int p8 = ms * MW, pp = ms / PER, h = ms % PER;
    const bool report = LPROG && sub_state != nullptr && ms >= 5 && ((ms - 5) & 3) == 0;      // column block (ms - 5) / 4: its last outputs left at the top of the previous micro-step
    if (report && wave < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    row_outputs();      // of micro-step ms - 1, beside the MFMA waves' head of this one
    const double* Lprev = LpB + ((ms + 1) & 1) * MW * kPLD;    // panels of micro-step ms - 1
    const double* Wprev = WpB + ((ms + 1) & 1) * MW * kPLD;
    if (mm) {
      // head: update of step ms - 1 on the tile column of panel ms, then the copy-out of panel ms
      if (ms > 0) {
#pragma unroll
        for (int q = 0; q < NT; ++q)
          if (tj_s[q] == pp) {
            const int rr = 16 * ti_s[q] + l15, cc = 16 * tj_s[q] + l15;
            double av[NE], bv[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { av[e] = Wprev[(4 * e + l4) * kPLD + cc]; bv[e] = Lprev[(4 * e + l4) * kPLD + rr]; }
#pragma unroll
            for (int e = 0; e < NE; ++e) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[e], bv[e], acc[q], 0, 0, 0);
          }
      }
#pragma unroll
      for (int q = 0; q < NT; ++q)
        if (tj_s[q] == pp) {
          const int r = 16 * ti_s[q] + l15;
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            // register h * NE + e of the tile (no dynamic register index: selects)
            double v;
            if constexpr (MW == 8) v = h == 0 ? acc[q][e] : acc[q][2 + e];
            else v = h == 0 ? acc[q][0] : (h == 1 ? acc[q][1] : (h == 2 ? acc[q][2] : acc[q][3]));
            Praw[(4 * e + l4) * kPLD + r] = v;
          }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (marks && tid == 0) { const long long tn = wall_clock64(); t_head += tn - tk; tk = tn; }
    if (report && wave == 0 && lane == 0) __hip_atomic_store(sub_state, ((ms - 5) >> 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // both row waves have waited in front of the barrier
    if (!mm) {
      // row threads: 8 x 8 diagonal LDL^T redundantly in registers, own row solved, the panels -L and W written for the MFMA waves.
      // The entries nobody waits for inside the loop (L to memory, the block's copy for the inverses, the pivots) are written
      // BEHIND the barrier, while the MFMA waves work on the head of the next micro-step (row_outputs, called at the top of the loop).
      double* Lp = LpB + (ms & 1) * MW * kPLD;
      double* Wp = WpB + (ms & 1) * MW * kPLD;
      if (tid >= p8 && tid < 128) {
        const int r = tid;
        double a[MW];
        double A[MW][MW], rd[MW];
#pragma unroll
        for (int c = 0; c < MW; ++c) {
          a[c] = Praw[c * kPLD + r];
#pragma unroll
          for (int i = c; i < MW; ++i) A[i][c] = Praw[c * kPLD + p8 + i];
        }
#pragma unroll
        for (int c = 0; c < MW; ++c) {
          rd[c] = fast_rcp_f64(A[c][c]);
          w_keep[c] = a[c];
          lr_keep[c] = w_keep[c] * rd[c];
#pragma unroll
          for (int i = c + 1; i < MW; ++i) {
            const double lic = A[i][c] * rd[c];
#pragma unroll
            for (int j = c + 1; j <= i; ++j) A[i][j] = __builtin_fma(-lic, A[j][c], A[i][j]);
          }
#pragma unroll
          for (int j = c + 1; j < MW; ++j) a[j] = __builtin_fma(-lr_keep[c], A[j][c], a[j]);
        }
#pragma unroll
        for (int c = 0; c < MW; ++c) {
          Lp[c * kPLD + r] = -lr_keep[c];
          Wp[c * kPLD + r] = w_keep[c];
        }
      }
      p8_keep = p8;
    } else if (ms > 0) {
      // rest of the update of step ms - 1: the tile columns to the right of panel ms's
#pragma unroll
      for (int q = 0; q < NT; ++q)
        if (tj_s[q] > pp) {
          const int rr = 16 * ti_s[q] + l15, cc = 16 * tj_s[q] + l15;
          double av[NE], bv[NE];
#pragma unroll
          for (int e = 0; e < NE; ++e) { av[e] = Wprev[(4 * e + l4) * kPLD + cc]; bv[e] = Lprev[(4 * e + l4) * kPLD + rr]; }
#pragma unroll
          for (int e = 0; e < NE; ++e) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[e], bv[e], acc[q], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (marks && tid == 0) { const long long tn = wall_clock64(); t_row += tn - tk; tk = tn; }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (marks && tid == 0) { const long long tn = wall_clock64(); t_rest += tn - tk; tk = tn; }
  }
  row_outputs();        // of the last micro-step; the inverses below read the block's copy, the pivot counts read dpiv
  if (LPROG && sub_state != nullptr && wave < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the whole of L and D is in memory ...
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (LPROG && sub_state != nullptr && wave == 0 && lane == 0) __hip_atomic_store(sub_state, (nb + 31) / 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ... and reported before the inverses are computed
  if (marks && tid == 0) { marks[0] = t_head; marks[1] = t_row; marks[2] = t_rest; marks[3] = wall_clock64(); }
  (void)my_d;
  if (wave == 4 || wave == 5) {
    // the pivots are stored and counted by two waves that have nothing else to do here (MFMA waves), from the copies the row threads
    // left in LDS: the row waves go straight on to the inverses below (1.2 us of the 38 of a diagonal block)
    unsigned pos = 0, neg = 0, zer = 0, bad = 0;
    const int t = tid - 256;
    if (t < nb) {
      const double d = dpiv[t];
      if (AG) st_agent_f64(&P.dvals[col0 + j0 + t], d); else P.dvals[col0 + j0 + t] = d;
      classify_pivot(d, tol, pos, neg, zer, bad);
    }
    flush_counts(P, 0, pos, neg, zer, bad);
  }
#ifdef OKKT_D_TAIL_MARKS
  if (marks && tid == 0) marks[0] = wall_clock64();
#endif
  // X_bb = inv(L_bb): wave b, one column per lane (Ld is complete behind the last barrier of the loop)
  const int off = wave * 32;
  if (wave < 4 && off < nb) {
    const int w = min(32, nb - off);
    const double* Lb = Ld + wave * 32 * kXld;
    double* Xb = Xs + wave * 32 * kXld;
    {
      // Column c = lane & 31 of the inverse by forward substitution, COLUMN-oriented: once x[p] is final it is applied to every row
      // below it -- independent FMAs -- so the dependent chain is 32 long, not 496 (row-oriented dot products, the first form: 4.8 us
      // of the 38 of a diagonal block, every FMA behind the one before it).  The two halves of the wave share a column: lane c keeps the
      // even rows, lane c + 32 the odd ones (16 values each), x[p] crosses with v_permlane32_swap.  Every entry still receives its
      // terms in ascending p: the same numbers bit for bit.  hipcc would wait for every pair of LDS values in front of the two FMAs that
      // use them (8.8 us): column p + 1 of L is requested before the FMAs of column p.
      const int c = lane & 31, hf = lane >> 5;
      const double* Lh = Lb + hf;                     // row 2 j + hf of column p at Lh[2 j + p * kXld]
      double v[16], la[16], lb[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (2 * j + hf == c) ? 1.0 : 0.0;
      // x[p] from the half that owns row p to both halves
      auto both = [&](double x, int owner) {
        const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto bq = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return owner == 0 ? __hiloint2double((int)bq[0], (int)a[0]) : __hiloint2double((int)bq[1], (int)a[1]);
      };
      // rows of column p below the diagonal: j >= (p + 1) / 2 (p even: j = p / 2 is row p itself in the even half: its multiplier is zero)
#pragma unroll
      for (int j = 0; j < 16; ++j) la[j] = Lh[2 * j];
#pragma unroll
      for (int p = 0; p < 32; p += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = (p + 2) / 2; j < 16; ++j) lb[j] = Lh[2 * j + (p + 1) * kXld];
        {
          // p even: owned by the even half, v[p / 2]
          const double mine = (p < w && c < w && p >= c) ? v[p / 2] : 0.0;
          const double xp = both(mine, 0);
          v[p / 2] = hf == 0 ? xp : __builtin_fma(-la[p / 2], xp, v[p / 2]);      // even half: x[p] itself; odd half: row p + 1
#pragma unroll
          for (int j = p / 2 + 1; j < 16; ++j) v[j] = __builtin_fma(-la[j], xp, v[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (p + 2 < 32) {
#pragma unroll
          for (int j = (p + 2) / 2; j < 16; ++j) la[j] = Lh[2 * j + (p + 2) * kXld];
        }
        {
          // p + 1 odd: owned by the odd half, v[p / 2]; the rows below it: j >= p / 2 + 1 in both halves
          const double mine = (p + 1 < w && c < w && p + 1 >= c) ? v[p / 2] : 0.0;
          const double xp = both(mine, 1);
          v[p / 2] = hf == 1 ? xp : v[p / 2];
#pragma unroll
          for (int j = p / 2 + 1; j < 16; ++j) v[j] = __builtin_fma(-lb[j], xp, v[j]);
        }
      }
      // the masks of the entries that were never anybody's x[p] at the point of use are already in: v[j] IS the masked x[2 j + hf]
#pragma unroll
      for (int j = 0; j < 16; ++j) Xb[(2 * j + hf) + c * kXld] = v[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#ifdef OKKT_D_TAIL_MARKS
    if (marks && tid == 0) marks[1] = wall_clock64();
#endif
    double* Xg = P.invl + P.invl_pos[s] + (size_t)step * NB * NB;
    const int r = lane & 31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = (lane >> 5) + 2 * q;
      if (AG) st_agent_f64(&Xg[(off + r) + (size_t)(off + c) * NB], Xb[r + c * kXld]); else Xg[(off + r) + (size_t)(off + c) * NB] = Xb[r + c * kXld];
    }
#ifdef OKKT_D_TAIL_MARKS
    if (marks && tid == 0) { marks[2] = wall_clock64(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); marks[2] = wall_clock64() * 0 + marks[2]; }
#endif
  }
}

// Rows below the diagonal block: W = A21 * L11^-T by blocked forward substitution over the four 32-column
// blocks, W_i = (A_i - sum_{p<i} W_p L_ip^T) X_ii^T, with only the 32 x 32 inverses X_ii (k_big_diag) -- the full
// 128 x 128 inverse is not needed on the critical path.  One wave owns 16 front rows and all 128 columns, so
// there is no cross-wave dependency.  FP64 MFMA 4x4x4: an accumulator register holds a 4-column x 16-row strip
// (lane <-> column l>>4, row l&15) and is, unchanged, the B operand of the k-step over those 4 columns: W_p
// feeds the later products straight from registers.  L blocks and X_ii are staged once per workgroup in LDS
// and read as broadcast A operands.  L21 = W * D^-1.
// flag != NULL: a workgroup of the fused diag + trsm launch (384 threads: the last two waves only meet the barriers) -- the rows
// of the panel are loaded first, then the front's diagonal block is awaited and staged with agent-scope loads
template <int NBLK, bool AG>
__device__ __forceinline__ void trsm_body(const DevPlan& P, int s, int step, int wcol0, int blk, double* sm, const int* flag, int epoch) {
  constexpr int NB = NBLK * kIB;
  constexpr int NPAIR = NBLK * (NBLK + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;          // sm: NPAIR blocks of 32 x 32, then NB reciprocals
  const bool worker = tid < 256;
  const int col0 = P.sn_col0[s];
  const int k = P.sn_col0[s + 1] - col0;
  const int f = (int)(P.row_ptr[s + 1] - P.row_ptr[s]);
  const int j0 = step * NB;
  if (j0 >= k) return;
  const int nb = min(NB, k - j0);
  const int r0 = j0 + nb + blk * 64;
  if (r0 >= f) return;
  double* F = P.arena + P.front_pos[s];
  double* Wb = P.wbuf + P.wbuf_pos[s] + (size_t)wcol0 * f;   // this panel's slot inside the super-step's W
  const double* X = P.invl + P.invl_pos[s] + (size_t)step * NB * NB;
  double* rdv = sm + NPAIR * kIB * kIB;
  const int row = r0 + (wv & 3) * 16 + (lane & 15);
  const int rowc = min(row, f - 1);
  const int lk = lane >> 4, li = lane & 3;
  double t[NBLK * 8];
  // the rows of the panel do not depend on the diagonal block: in flight before the wait
#pragma unroll
  for (int q = 0; q < NBLK * 8; ++q) {
    const int c = 4 * q + lk;
    t[q] = keep_f64(F[(size_t)(j0 + min(c, nb - 1)) * f + rowc], c < nb && row < f && worker);
  }
  if (AG) {
    if (tid == 0) {
      // bounded like every in-launch wait; a wait that runs into its bound is counted as a non-finite pivot and raises the time-out
      // word (flow_wait does the same): the factorisation fails on its pivot counts instead of solving with a block that never came
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
      if (spins >= (1 << 22)) { atomicAdd(&P.counters[3], 1ull); atomicExch(&P.counters[5], 1ull); }
    }
    __syncthreads();
  }
  // stage block (bi, bp), bp <= bi, at index bi(bi+1)/2 + bp: L_{bi,bp} below the diagonal, X_ii on it
  if (worker) {
    const int e = tid * 4;                 // 4 consecutive rows of one column per thread and block
    const int cc = e / kIB, rr = e - cc * kIB;
#pragma unroll
    for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
      for (int bp = 0; bp <= bi; ++bp) {
        double v[4];
        const int gr = bi * kIB + rr, gc = bp * kIB + cc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double* src = bp == bi ? X + (gr + u) + (size_t)gc * NB
                                       : F + (size_t)(j0 + min(gc, nb - 1)) * f + j0 + min(gr + u, nb - 1);
          v[u] = keep_f64(AG ? ld_agent_f64(src) : *src, gr + u < nb && gc < nb);
        }
        double* dst = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB + e;
#pragma unroll
        for (int u = 0; u < 4; ++u) dst[u] = v[u];
      }
    if (tid < NB) rdv[tid] = tid < nb ? 1.0 / (AG ? ld_agent_f64(&P.dvals[col0 + j0 + tid]) : P.dvals[col0 + j0 + tid]) : 0.0;
  }
  __syncthreads();
  if (!worker) return;
#pragma unroll
  for (int bi = 0; bi < NBLK; ++bi) {
#pragma unroll
    for (int bp = 0; bp < bi; ++bp) {
      const double* Lb = sm + (bi * (bi + 1) / 2 + bp) * kIB * kIB;
#pragma unroll
      for (int gp = 0; gp < 8; ++gp)
#pragma unroll
        for (int g = 0; g < 8; ++g)
          t[bi * 8 + gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Lb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bp * 8 + g], t[bi * 8 + gp], 0, 0, 1 /* neg A */);
    }
    const double* Xb = sm + (bi * (bi + 1) / 2 + bi) * kIB * kIB;
    double wt[8];
#pragma unroll
    for (int gp = 0; gp < 8; ++gp) {
      wt[gp] = 0.0;
#pragma unroll
      for (int g = 0; g <= gp; ++g)   // X_ii is lower triangular
        wt[gp] = __builtin_amdgcn_mfma_f64_4x4x4f64(Xb[(gp * 4 + li) + (g * 4 + lk) * kIB], t[bi * 8 + g], wt[gp], 0, 0, 0);
    }
#pragma unroll
    for (int gp = 0; gp < 8; ++gp) t[bi * 8 + gp] = wt[gp];
  }
  if (row < f) {
#pragma unroll
    for (int q = 0; q < NBLK * 8; ++q) {
      const int c = 4 * q + lk;
      if (c < nb) {
        Wb[(size_t)c * f + row] = t[q];
        F[(size_t)(j0 + c) * f + row] = t[q] * rdv[c];
      }
    }
  }
}

#ifndef OKKT_SYRK_KC
#define OKKT_SYRK_KC 16
#endif
#ifndef OKKT_SYRK_STAGES
#define OKKT_SYRK_STAGES 2
#endif
constexpr int kSyrkKC = OKKT_SYRK_KC;          // k-columns per ring slot
constexpr int kSyrkStages = OKKT_SYRK_STAGES;  // ring slots (KC * stages = 32 keeps two workgroups per CU)
#ifndef OKKT_SYRK_LD
#define OKKT_SYRK_LD 144
#endif
constexpr int kSyrkLd = OKKT_SYRK_LD;   // leading dimension of the LDS panels (doubles)
constexpr size_t syrk_lds_bytes(int stages) { return (size_t)stages * 2 * kSyrkKC * kSyrkLd * sizeof(double); }
typedef __attribute__((address_space(3))) void lds_void_t;

// HEAD selects which tiles of the region that starts at block column tstep a launch updates:
//   kSyrkTrail (the dominant kernel): the lower triangle without its first `csplit` tile columns
//                (csplit = 0: the whole super-step update; csplit = GS: what is left beside the look-ahead columns)
//   kSyrkPanel : only the next panel's own columns [t0, min(t0 + NB, k)) -- what that panel needs before it
//                can be factored (in-group update, K = NB)
//   kSyrkAhead : the first `csplit` tile columns -- all that the next super-step's panels need (look-ahead)
// DBG != 0 are timing-only ablations (wrong results): bit0 no C load, bit1 no MFMA, bit2 no store, bit3 no LDS-DMA.
enum { kSyrkTrail = 0, kSyrkPanel = 1, kSyrkAhead = 2 };
constexpr int kSyrkNW = 8;   // 2 x 4 waves of 64 rows x 32 columns: four waves per SIMD with two workgroups per CU,
                             // so a wave's C-tile load/store hides behind three other waves' MFMAs

}  // namespace okkt
