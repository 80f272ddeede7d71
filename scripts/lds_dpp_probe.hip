// The update task's inner loop (dataflow.hip: df_syrk_tiles) is bound by the CU's LDS pipe: per k-step of four panel columns a wave reads
// 4 row fragments and 8 column fragments (12 x ds_read_b64) for 32 MFMAs.  A column fragment holds only 16 distinct doubles (4 columns x 4 k),
// replicated over the four blocks of v_mfma_f64_4x4x4.  This probe checks the alternative: ONE read fetches 16 columns x 4 k (64 distinct
// doubles, one per lane), and the other three operands are that register ROTATED by 4 / 8 / 12 lanes inside every row of 16 lanes (DPP
// row_ror, VALU, no LDS): block t then multiplies with column group (t -+ m) & 3 -- the same 16 x 16 products, dealt to the accumulators
// in a skewed order that the C tile's load / store addressing follows.  6 LDS reads per k-step instead of 12.
//   MODE 0: the kernel's loop (12 reads);  MODE 1: rotated column fragments (6 reads + 12 v_mov_b32_dpp)
// Verifies C = -W L^T against the host for both and prints cycles per MFMA per SIMD (two waves per SIMD: floor 16).
// build: hipcc -O3 -w --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/lds_dpp_probe.hip -o scripts/_bin/lds_dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
constexpr int kLd = 144, KC = 32;
template <int CTRL>
__device__ __forceinline__ double rot(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int MODE, int DMA>
__global__ __launch_bounds__(512, 1) void k(const double* __restrict__ W, const double* __restrict__ L, double* __restrict__ C, long long* cyc, int reps, const double* src) {
  extern __shared__ double sm[];   // two slots of [W rows: KC x kLd][L columns: KC x kLd]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  for (int e = tid; e < KC * 128; e += 512) {
    const int p = e >> 7, r = e & 127;
    sm[p * kLd + r] = W[(size_t)p * 128 + r];
    sm[(KC + p) * kLd + r] = L[(size_t)p * 128 + r];
    sm[2 * KC * kLd + p * kLd + r] = W[(size_t)p * 128 + r];
    sm[2 * KC * kLd + (KC + p) * kLd + r] = L[(size_t)p * 128 + r];
  }
  __syncthreads();
  double acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  const int quad = (lane >> 2) & 3;
  int grp[4];
  grp[0] = quad;
  grp[1] = __builtin_amdgcn_update_dpp(0, quad, 0x124, 0xf, 0xf, false);
  grp[2] = __builtin_amdgcn_update_dpp(0, quad, 0x128, 0xf, 0xf, false);
  grp[3] = __builtin_amdgcn_update_dpp(0, quad, 0x12C, 0xf, 0xf, false);
  const long long t0 = clock64();
  for (int c = 0; c < reps; ++c) {
    __syncthreads();
    const double* slot = sm + (c & 1) * 2 * KC * kLd;
    if (DMA) {
      double* dst = sm + ((c + 1) & 1) * 2 * KC * kLd;
#pragma unroll
      for (int qq = 0; qq < KC / 8; ++qq) {
        const int prow = qq * 8 + wv;
        __builtin_amdgcn_global_load_lds(src + ((size_t)blockIdx.x * 64 + (c & 31)) * 8192 + prow * 128 + lane * 2, (__attribute__((address_space(3))) void*)(dst + prow * kLd), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + ((size_t)blockIdx.x * 64 + (c & 31)) * 8192 + (KC + prow) * 128 + lane * 2, (__attribute__((address_space(3))) void*)(dst + (KC + prow) * kLd), 16, 0, 0);
      }
    }
    const double* bw = slot + (wv & 1) * 64 + 2 * l15;
    if (MODE == 0) {
      const double* bl = slot + KC * kLd + (wv >> 1) * 32 + (lane & 3);
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        double bv[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) bv[rb] = bw[(kk * 4 + l4) * kLd + (rb & 1) + 32 * (rb >> 1)];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          double av[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) av[qq] = bl[(kk * 4 + l4) * kLd + (half * 4 + qq) * 4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[half * 4 + qq][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[qq], bv[rb], acc[half * 4 + qq][rb], 0, 0, 1);
        }
      }
    } else if (MODE == 3) {
      // MODE 3: as MODE 2 with the LDS reads and their waits written by hand (the compiler waits lgkmcnt(0) for reads it has just issued as soon as
      // LDS-DMA is in the loop): in-order LDS returns, lgkmcnt(2) / lgkmcnt(4) in front of the MFMAs of a step
      typedef double d2v __attribute__((ext_vector_type(2)));
      const unsigned ba = (unsigned)(size_t)(slot + KC * kLd + (wv >> 1) * 32 + (lane & 3) + l4 * kLd) * 1u;      // LDS byte address of the column fragments, k-step 0
      const unsigned bb = (unsigned)(size_t)(bw + l4 * kLd);
      d2v B0[2], B1[2], A0[2], A1[2];      // two buffers each: [0] = values (0,1), [1] = values (2,3)
      auto ldb = [&](int kk, d2v (&b)[2]) {
        const unsigned ad = bb + kk * 4 * kLd * 8;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:256" : "=&v"(b[0]), "=&v"(b[1]) : "v"(ad));
      };
      auto lda = [&](int kk, int half, d2v (&a)[2]) {
        const unsigned ad = ba + kk * 4 * kLd * 8 + half * 128;
        asm volatile("ds_read2_b64 %0, %2 offset1:4\n\tds_read2_b64 %1, %2 offset0:8 offset1:12" : "=&v"(a[0]), "=&v"(a[1]) : "v"(ad));
      };
      ldb(0, B0);
      lda(0, 0, A0);
#pragma unroll
      for (int st = 0; st < 2 * (KC / 4); ++st) {
        const int kk = st >> 1, half = st & 1;
        d2v (&Ac)[2] = (st & 1) ? A1 : A0;
        d2v (&Bc)[2] = (kk & 1) ? B1 : B0;
        if (st + 1 < 2 * (KC / 4)) {
          const int k2 = (st + 1) >> 1, h2 = (st + 1) & 1;
          if (h2 == 0) ldb(k2, (k2 & 1) ? B1 : B0);
          lda(k2, h2, ((st + 1) & 1) ? A1 : A0);
        }
        // everything older than the reads just requested has returned
        if (st + 1 >= 2 * (KC / 4)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Bc[0]), "+v"(Bc[1]));
        else if (((st + 1) & 1) == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Bc[0]), "+v"(Bc[1]));
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Ac[0]), "+v"(Ac[1]), "+v"(Bc[0]), "+v"(Bc[1]));
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) acc[half * 4 + qq][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ac[qq >> 1][qq & 1], Bc[rb >> 1][rb & 1], acc[half * 4 + qq][rb], 0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);      // nothing moves across the step boundary
      }
    } else if (MODE == 2) {
      // MODE 2: the kernel's loop with the operand reads of step s + 1 (a k-step half: 4 column fragments, every other step 4 row fragments too)
      // requested BEFORE the 16 MFMAs of step s, the order pinned with sched_group_barrier (DS reads, then MFMAs)
      const double* bl = slot + KC * kLd + (wv >> 1) * 32 + (lane & 3);
      double bvb[2][4], avb[2][4];
      auto ld_b = [&](int kk, double (&b)[4]) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) b[rb] = bw[(kk * 4 + l4) * kLd + (rb & 1) + 32 * (rb >> 1)];
      };
      auto ld_a = [&](int kk, int half, double (&a)[4]) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) a[qq] = bl[(kk * 4 + l4) * kLd + (half * 4 + qq) * 4];
      };
      ld_b(0, bvb[0]);
      ld_a(0, 0, avb[0]);
#pragma unroll
      for (int st = 0; st < 2 * (KC / 4); ++st) {
        const int kk = st >> 1, half = st & 1;
        if (st + 1 < 2 * (KC / 4)) {
          const int k2 = (st + 1) >> 1, h2 = (st + 1) & 1;
          if (h2 == 0) ld_b(k2, bvb[k2 & 1]);
          ld_a(k2, h2, avb[(st + 1) & 1]);
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) acc[half * 4 + qq][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(avb[st & 1][qq], bvb[kk & 1][rb], acc[half * 4 + qq][rb], 0, 0, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // the DS reads of the next step first ...
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);     // ... then this step's MFMAs
      }
    } else {
      const double* bl = slot + KC * kLd + (wv >> 1) * 32 + l15;
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        double bv[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) bv[rb] = bw[(kk * 4 + l4) * kLd + (rb & 1) + 32 * (rb >> 1)];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          double av[4];
          av[0] = bl[(kk * 4 + l4) * kLd + half * 16];
          av[1] = rot<0x124>(av[0]);
          av[2] = rot<0x128>(av[0]);
          av[3] = rot<0x12C>(av[0]);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[half * 4 + qq][rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[qq], bv[rb], acc[half * 4 + qq][rb], 0, 0, 1);
        }
      }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
  // C tile of the workgroup: 128 x 128, column-major
  double* Cb = C + (size_t)blockIdx.x * 128 * 128;
#pragma unroll
  for (int cg = 0; cg < 8; ++cg) {
    const int half = cg >> 2, m = cg & 3;
    const int col = (wv >> 1) * 32 + (MODE != 1 ? cg * 4 : half * 16 + grp[m] * 4) + l4;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int row = (wv & 1) * 64 + 2 * l15 + (rb & 1) + 32 * (rb >> 1);
      Cb[(size_t)col * 128 + row] = acc[cg][rb];
    }
  }
}
template <int MODE, int DMA>
void run(const char* label, int ncu) {
  std::vector<double> hW(KC * 128), hL(KC * 128);
  for (int i = 0; i < KC * 128; ++i) { hW[i] = std::sin(0.37 * i) ; hL[i] = std::cos(0.11 * i + 1.0); }
  double *W, *L, *C, *src; long long* cyc;
  hipMalloc(&W, KC * 128 * 8); hipMalloc(&L, KC * 128 * 8); hipMalloc(&C, (size_t)ncu * 128 * 128 * 8); hipMalloc(&cyc, ncu * 8 * 8);
  hipMalloc(&src, (size_t)ncu * 64 * 8192 * 8); hipMemset(src, 0, (size_t)ncu * 64 * 8192 * 8);
  hipMemcpy(W, hW.data(), KC * 128 * 8, hipMemcpyHostToDevice); hipMemcpy(L, hL.data(), KC * 128 * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k<MODE, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const size_t lds = (size_t)2 * 2 * KC * kLd * 8;
  // correctness: one pass (no DMA overwrite: src is only read when DMA, and then the second slot is overwritten with zeros AFTER pass 0 reads slot 0)
  hipLaunchKernelGGL((k<MODE, 0>), dim3(1), dim3(512), lds, 0, W, L, C, cyc, 1, src);
  hipDeviceSynchronize();
  std::vector<double> hC(128 * 128); hipMemcpy(hC.data(), C, 128 * 128 * 8, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int c = 0; c < 128; ++c) for (int r = 0; r < 128; ++r) {
    double s = 0; for (int p = 0; p < KC; ++p) s -= hW[p * 128 + r] * hL[p * 128 + c];
    maxerr = std::max(maxerr, std::fabs(s - hC[(size_t)c * 128 + r]));
  }
  const int reps = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, DMA>), dim3(ncu), dim3(512), lds, 0, W, L, C, cyc, reps, src);
  hipDeviceSynchronize();
  std::vector<long long> h(ncu * 8); hipMemcpy(h.data(), cyc, ncu * 8 * 8, hipMemcpyDeviceToHost);
  std::vector<double> per; for (int b = 0; b < ncu; ++b) { long long mx = 0; for (int w = 0; w < 8; ++w) mx = std::max(mx, h[b * 8 + w]); per.push_back((double)mx); }
  std::sort(per.begin(), per.end());
  const double med = per[per.size() / 2];
  // per wave and pass: 8 k-steps x 32 MFMAs = 256; two waves per SIMD
  printf("%-40s max |C - host| %.2e   %.0f cycles per 32-column chunk, %.1f cycles per MFMA per SIMD (%s)\n", label, maxerr, med / reps, med / (reps * 512.0), hipGetErrorString(hipGetLastError()));
  hipFree(W); hipFree(L); hipFree(C); hipFree(cyc); hipFree(src);
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  run<0, 0>("kernel's loop (12 LDS reads), no DMA", ncu);
  run<1, 0>("rotated column fragments, no DMA", ncu);
  run<0, 1>("kernel's loop (12 LDS reads), DMA", ncu);
  run<1, 1>("rotated column fragments, DMA", ncu);
  run<2, 0>("reads of the next step before the MFMAs, no DMA", ncu);
  run<2, 1>("reads of the next step before the MFMAs, DMA", ncu);
  run<3, 0>("... with hand-written reads and waits, no DMA", ncu);
  run<3, 1>("... with hand-written reads and waits, DMA", ncu);
  return 0;
}
