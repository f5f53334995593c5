// Workgroup-cooperative fp64 building blocks for gfx950 (CDNA4): one 512-thread workgroup (8 wave64)
// owns one dense problem.  All matrices are column-major.
//
//   wg_gemm      O = S * X with MFMA v_mfma_f64_16x16x4_f64, operands addressed through separable
//                (row-offset + column-offset) index maps so that tensor contractions over composite
//                indices need no explicit transposes/reshapes in HBM; tiles staged through LDS.
//   wg_qr_r      in-place blocked Householder QR keeping only R (panel width 16; trailing update
//                V, T, C all through MFMA straight from global/L2 with 128-B-line aligned fragments).
//   wg_jacobi_rsv one-sided (Hestenes) Jacobi on a small matrix: right singular vectors + values.
//
// f64 MFMA fragment maps (cdna_hip_programming.md section 3, "f64 MFMA does NOT use these maps"):
//   A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][j = l&15];
//   C/D: reg r of lane l is C[row = (l>>4) + 4r][col = l&15].
#if !defined(WG_THREADS) || !defined(WG_WAVES)
#error "define WG_THREADS / WG_WAVES and include wg_common.h before wg_blocks.h (see kernels.h)"
#endif

namespace wg {

using namespace wgc;

constexpr int GM_MB = 96;    // rows per M block (6 MFMA row tiles)
constexpr int GM_NT = 128;   // columns per N chunk (one 16-col tile per wave)
constexpr int GM_KC = 32;    // K per LDS chunk: 8 k-steps between two barriers, 8 + 8 loads per thread in flight (16 until round 4: the M_t
                             // product of configs[1] waited for its loads at every chunk; 8192 doubles of LDS, inside the engine's QR_BIG_DOUBLES)
constexpr int GM_LDA = 112;  // LDS leading dims: == 16 (mod 32) doubles so that the two 16-lane runs of a
constexpr int GM_LDB = 144;  // ds_read_b64 half-wave land on disjoint banks
constexpr int GM_LDS_DOUBLES = GM_KC * (GM_LDA + GM_LDB);   // 8192 doubles = 64 KiB

template <bool NTX = false, bool NTO = false, class SP, class XP, class OP, class SRO, class SCO, class XRO, class XCO,
          class ORO, class OCO>
__device__ __attribute__((noinline)) void gemm_direct(int M, int N, int K, SP Sp, SRO sro, SCO sco,
                                                     XP Xp, XRO xro, XCO xco, OP Op, ORO oro, OCO oco,
                                                     bool accumulate, int tile0 = -1, int tstride = WG_WAVES);

// O(i,j) (+)= sum_k S(i,k) X(k,j),  i<M, j<N, k<K.
//   S(i,k) = Sp[sro(i) + sco(k)],  X(k,j) = Xp[xro(k) + xco(j)],  O(i,j) = Op[oro(i) + oco(j)].
// `xkfast`: X is contiguous along k (else along j) - only picks the coalesced staging pattern.
// `lds` must provide GM_LDS_DOUBLES doubles.  Ends with a __syncthreads().
template <class SRO, class SCO, class XRO, class XCO, class ORO, class OCO>
__device__ __attribute__((noinline)) void gemm(int M, int N, int K, const gdbl* Sp, SRO sro, SCO sco, const gdbl* Xp, XRO xro,
                     XCO xco, bool xkfast, gdbl* Op, ORO oro, OCO oco, bool accumulate, ldbl* lds) {
#if WG_THREADS != 512
  // the LDS staging maps below are laid out for 512 threads; smaller workgroups take the barrier-free form
  (void)xkfast; (void)lds;
  gemm_direct<false, false>(M, N, K, Sp, sro, sco, Xp, xro, xco, Op, oro, oco, accumulate);
#else
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  ldbl* As = lds;
  ldbl* Bs = lds + GM_KC * GM_LDA;
  for (int mb = 0; mb < M; mb += GM_MB) {
    const int mrows = min(GM_MB, M - mb);
    const int ntm = (mrows + 15) >> 4;
    // staging map for S: i = tid & 127 (only < 96 used), kk = (tid >> 7) + 4e
    const int si = tid & 127;
    const bool s_act = si < ntm * 16;
    const bool s_in = si < mrows;
    const long s_ro = s_in ? (long)sro(mb + si) : 0;
    for (int n0 = 0; n0 < N; n0 += GM_NT) {
      const int ncols = min(GM_NT, N - n0);
      // staging map for X
      constexpr int NE = GM_KC / 4;          // elements per thread and chunk of each operand (512 threads, 128 x GM_KC tile)
      constexpr int JS = 512 / GM_KC;        // xkfast: kk = tid % GM_KC, j = tid / GM_KC + JS e
      long x_co[NE];
      bool x_in[NE];
      if (xkfast) {
#pragma unroll
        for (int e = 0; e < NE; e++) {
          int j = tid / GM_KC + JS * e;
          x_in[e] = j < ncols;
          x_co[e] = x_in[e] ? (long)xco(n0 + j) : 0;
        }
      } else {           // j = tid & 127, kk = (tid >> 7) + 4e
        int j = tid & 127;
        x_in[0] = j < ncols;
        x_co[0] = x_in[0] ? (long)xco(n0 + j) : 0;
      }
      d4 acc[6];
#pragma unroll
      for (int t = 0; t < 6; t++) acc[t] = d4{0, 0, 0, 0};
      const bool w_act = wave * 16 < ncols;
      // Register double buffering: the global loads of chunk k0 + GM_KC are issued before the MFMAs of chunk k0
      // and only written to LDS after them (branch-free clamped loads + selects, so that they stay in flight).
      double sv[NE], xv[NE];
      auto load_chunk = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NE; e++) {
          const int kk = (tid >> 7) + 4 * e;
          const bool ok = s_in && k0 + kk < K;
          const double v = Sp[s_ro + (long)sco(ok ? k0 + kk : 0)];
          sv[e] = ok ? v : 0.0;
        }
        if (xkfast) {
          const int kk = tid % GM_KC;
          const bool kin = k0 + kk < K;
          const long r = (long)xro(kin ? k0 + kk : 0);
#pragma unroll
          for (int e = 0; e < NE; e++) {
            const double v = Xp[r + x_co[e]];
            xv[e] = (kin && x_in[e]) ? v : 0.0;
          }
        } else {
#pragma unroll
          for (int e = 0; e < NE; e++) {
            const int kk = (tid >> 7) + 4 * e;
            const bool ok = x_in[0] && k0 + kk < K;
            const double v = Xp[(long)xro(ok ? k0 + kk : 0) + x_co[0]];
            xv[e] = ok ? v : 0.0;
          }
        }
      };
      load_chunk(0);
      for (int k0 = 0; k0 < K; k0 += GM_KC) {
        // ---- stage the S tile As[kk][i] and the X tile Bs[kk][j] from the registers
        if (s_act) {
#pragma unroll
          for (int e = 0; e < NE; e++) As[((tid >> 7) + 4 * e) * GM_LDA + si] = sv[e];
        }
        if (xkfast) {
#pragma unroll
          for (int e = 0; e < NE; e++) Bs[(tid % GM_KC) * GM_LDB + tid / GM_KC + JS * e] = xv[e];
        } else {
#pragma unroll
          for (int e = 0; e < NE; e++) Bs[((tid >> 7) + 4 * e) * GM_LDB + (tid & 127)] = xv[e];
        }
        __syncthreads();
        if (k0 + GM_KC < K) load_chunk(k0 + GM_KC);
        if (w_act) {
#pragma unroll
          for (int ks = 0; ks < GM_KC / 4; ks++) {
            const int krow = ks * 4 + g;
            const double b = Bs[krow * GM_LDB + wave * 16 + l15];
#pragma unroll
            for (int t = 0; t < 6; t++) {
              if (t < ntm) {
                const double a = As[krow * GM_LDA + t * 16 + l15];
                acc[t] = mfma(a, b, acc[t]);
              }
            }
          }
        }
        __syncthreads();
      }
      // ---- write out
      if (w_act) {
        const int col = wave * 16 + l15;
        if (col < ncols) {
          const long oc = (long)oco(n0 + col);
#pragma unroll
          for (int t = 0; t < 6; t++) {
            if (t < ntm) {
#pragma unroll
              for (int r = 0; r < 4; r++) {
                int row = t * 16 + g + 4 * r;
                if (row < mrows) {
                  gdbl* p = Op + (long)oro(mb + row) + oc;
                  double v = acc[t][r];
                  if (accumulate) v += *p;
                  *p = v;
                }
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
#endif
}

// Barrier-free variant for contractions whose M-side operand is small or cache resident (cores / coupling
// tables in LDS, triangular factors in L2): every wave owns whole 16-column tiles of the output, takes its B
// fragments straight from global memory (8 B per lane per k-step, all k-steps of a tile in flight together)
// and its A fragments from wherever S lives.  No LDS staging, no __syncthreads() inside; waves never wait for
// each other.  Same index-map interface as gemm().  Ends with a __syncthreads().
// SP / XP / OP are address-space typed pointers (ldbl* or gdbl*, see wg_common.h).  NTX / NTO: nontemporal loads of
// X / stores of O - for streams far larger than the caches that would only evict the operands that are reused.
template <bool NTX, bool NTO, class SP, class XP, class OP, class SRO, class SCO, class XRO, class XCO, class ORO, class OCO>
__device__ __attribute__((noinline)) void gemm_direct(int M, int N, int K, SP Sp, SRO sro, SCO sco,
                                                     XP Xp, XRO xro, XCO xco, OP Op, ORO oro, OCO oco,
                                                     bool accumulate, int tile0, int tstride) {
  // tile0 / tstride: first 16-column output tile of this wave and the stride to its next one (defaults: the waves of
  // ONE workgroup share the tiles; the grid-level contractions of the batched sweep pass global wave numbers)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int nks = (K + 3) >> 2;
  const int ntile = (N + 15) >> 4;
  for (int mb = 0; mb < M; mb += GM_MB) {
    const int mrows = min(GM_MB, M - mb);
    const int ntm = (mrows + 15) >> 4;
    long s_ro[6];
    bool s_in[6];
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const int i = t * 16 + l15;
      s_in[t] = t < ntm && i < mrows;
      s_ro[t] = s_in[t] ? (long)sro(mb + i) : 0;
    }
    // Software pipeline over (tile, group of 5 k-steps): the X loads of the next group - of this tile or of the
    // wave's next tile - are issued before the MFMAs of the current one.  (Measured at low occupancy these
    // contractions are latency bound, not bandwidth bound.)
    struct Grp { double b[5]; long so[5]; unsigned kin; };
    auto fetch = [&](long cofs, int ks0, Grp& G) {
      G.kin = 0;
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int k = 4 * (ks0 + u) + g;
        const bool kin = (ks0 + u < nks) && k < K;
        const int kc = kin ? k : 0;
        G.b[u] = NTX ? __builtin_nontemporal_load(Xp + ((long)xro(kc) + cofs)) : Xp[(long)xro(kc) + cofs];   // valid address, masked later
        G.so[u] = (long)sco(kc);
        G.kin |= kin ? (1u << u) : 0u;
      }
    };
    // The tile loop with the number of 16-row tiles of this M block a compile-time constant: as a run-time bound every (tile, k-step)
    // sat in its own exec-masked basic block - address add, ds_read, s_waitcnt lgkmcnt(0), two selects, MFMA: the LDS latency exposed
    // once per MFMA and 4 VALU instructions beside each (tools/valu_audit.py).  The A operand needs no mask at all: a row of the output
    // depends on that row of A only, rows >= mrows are never stored, and past the end of K the B operand is zero (A is read from a
    // clamped, valid address there).
    auto tiles = [&](auto ntm_c) {
      constexpr int NTM = decltype(ntm_c)::value;
      int tl = tile0 < 0 ? wave : tile0;
      if (tl >= ntile) return;
      int j = tl * 16 + l15;
      bool jin = j < N;
      long cofs = jin ? (long)xco(j) : 0;
      Grp cur, nxt;
      fetch(cofs, 0, cur);
      for (;;) {
        d4 acc[NTM];
#pragma unroll
        for (int t = 0; t < NTM; t++) acc[t] = d4{0, 0, 0, 0};
        const int tl2 = tl + tstride;
        const int j2 = tl2 * 16 + l15;
        const bool jin2 = j2 < N;
        long cofs2 = 0;
        for (int ks0 = 0; ks0 < nks; ks0 += 5) {
          if (ks0 + 5 < nks) fetch(cofs, ks0 + 5, nxt);                 // wave-uniform branches
          else if (tl2 < ntile) { cofs2 = jin2 ? (long)xco(j2) : 0; fetch(cofs2, 0, nxt); }
#pragma unroll
          for (int u = 0; u < 5; u++) {
            const bool kin = (cur.kin >> u) & 1u;
            const double bv = (kin && jin) ? cur.b[u] : 0.0;
#pragma unroll
            for (int t = 0; t < NTM; t++) acc[t] = mfma(Sp[s_ro[t] + cur.so[u]], bv, acc[t]);
          }
          cur = nxt;
        }
        if (jin) {
          const long oc = (long)oco(j);
#pragma unroll
          for (int t = 0; t < NTM; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int row = t * 16 + g + 4 * r;
              if (row < mrows) {
                OP p = Op + (long)oro(mb + row) + oc;
                double v = acc[t][r];
                if (accumulate) v += *p;
                if (NTO) __builtin_nontemporal_store(v, p); else *p = v;
              }
            }
          }
        }
        tl = tl2;
        if (tl >= ntile) break;
        j = j2; jin = jin2; cofs = cofs2;
      }
    };
    // K <= 40 (every contraction of configs[1]): the k-dependent parts of the index maps - integer divisions by run-time bond
    // dimensions, 64-bit products: ~100 VALU instructions per group of five k-steps, re-evaluated for every tile - are evaluated
    // ONCE per M block and kept in registers; a tile then costs one 64-bit add per X load and one add per A operand.
    constexpr int NGC = 2;
    if (nks <= 5 * NGC) {
      long xr_[NGC][5], so_[NGC][5];
      unsigned kinm[NGC];
#pragma unroll
      for (int gi = 0; gi < NGC; gi++) {
        kinm[gi] = 0;
#pragma unroll
        for (int u = 0; u < 5; u++) {
          const int k = 4 * (5 * gi + u) + g;
          const bool kin = (5 * gi + u < nks) && k < K;
          const int kc = kin ? k : 0;
          xr_[gi][u] = (long)xro(kc);
          so_[gi][u] = (long)sco(kc);
          kinm[gi] |= kin ? (1u << u) : 0u;
        }
      }
      auto tiles_c = [&](auto ntm_c) {
        constexpr int NTM = decltype(ntm_c)::value;
        int tl = tile0 < 0 ? wave : tile0;
        if (tl >= ntile) return;
        int j = tl * 16 + l15;
        bool jin = j < N;
        long cofs = jin ? (long)xco(j) : 0;
        double cb[5], nb[5];
        auto fetchc = [&](long co, int gi, double (&b)[5]) {          // gi: 0 or 1, resolved at compile time after unrolling
#pragma unroll
          for (int u = 0; u < 5; u++) {
            const long o = (gi == 0 ? xr_[0][u] : xr_[NGC - 1][u]) + co;
            b[u] = NTX ? __builtin_nontemporal_load(Xp + o) : Xp[o];
          }
        };
        fetchc(cofs, 0, cb);
        for (;;) {
          d4 acc[NTM];
#pragma unroll
          for (int t = 0; t < NTM; t++) acc[t] = d4{0, 0, 0, 0};
          const int tl2 = tl + tstride;
          const int j2 = tl2 * 16 + l15;
          const bool jin2 = j2 < N;
          long cofs2 = 0;
#pragma unroll
          for (int gi = 0; gi < NGC; gi++) {
            if (5 * gi >= nks) break;                                   // wave-uniform
            if (gi + 1 < NGC && 5 * (gi + 1) < nks) fetchc(cofs, gi + 1, nb);
            else if (tl2 < ntile) { cofs2 = jin2 ? (long)xco(j2) : 0; fetchc(cofs2, 0, nb); }
#pragma unroll
            for (int u = 0; u < 5; u++) {
              const bool kin = (kinm[gi] >> u) & 1u;
              const double bv = (kin && jin) ? cb[u] : 0.0;
#pragma unroll
              for (int t = 0; t < NTM; t++) acc[t] = mfma(Sp[s_ro[t] + so_[gi][u]], bv, acc[t]);
            }
#pragma unroll
            for (int u = 0; u < 5; u++) cb[u] = nb[u];
          }
          if (jin) {
            const long oc = (long)oco(j);
#pragma unroll
            for (int t = 0; t < NTM; t++) {
#pragma unroll
              for (int r = 0; r < 4; r++) {
                const int row = t * 16 + g + 4 * r;
                if (row < mrows) {
                  OP p = Op + (long)oro(mb + row) + oc;
                  double v = acc[t][r];
                  if (accumulate) v += *p;
                  if (NTO) __builtin_nontemporal_store(v, p); else *p = v;
                }
              }
            }
          }
          tl = tl2;
          if (tl >= ntile) break;
          j = j2; jin = jin2; cofs = cofs2;
        }
      };
      switch (ntm) {
        case 1: tiles_c(std::integral_constant<int, 1>{}); break;
        case 2: tiles_c(std::integral_constant<int, 2>{}); break;
        case 3: tiles_c(std::integral_constant<int, 3>{}); break;
        case 4: tiles_c(std::integral_constant<int, 4>{}); break;
        case 5: tiles_c(std::integral_constant<int, 5>{}); break;
        default: tiles_c(std::integral_constant<int, 6>{}); break;
      }
      continue;
    }
    switch (ntm) {
      case 1: tiles(std::integral_constant<int, 1>{}); break;
      case 2: tiles(std::integral_constant<int, 2>{}); break;
      case 3: tiles(std::integral_constant<int, 3>{}); break;
      case 4: tiles(std::integral_constant<int, 4>{}); break;
      case 5: tiles(std::integral_constant<int, 5>{}); break;
      default: tiles(std::integral_constant<int, 6>{}); break;
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// workgroup reductions through LDS (red must hold WG_WAVES*NV doubles)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// all threads receive the sum; red: >= WG_WAVES doubles; contains 2 barriers
__device__ __forceinline__ double wg_sum(double v, ldbl* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int w = 0; w < WG_WAVES; w++) s += red[w];
  return s;
}
__device__ __forceinline__ double wg_max(double v, ldbl* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < WG_WAVES; w++) s = fmax(s, red[w]);
  return s;
}

// max |x| over a strided 2D region (rows x cols, leading dim ld); all threads get it
template <class AP>
__device__ inline double wg_maxabs(AP A, long ld, int rows, int cols, ldbl* red) {
  double m = 0.0;
  const long tot = (long)rows * cols;
  for (long idx = threadIdx.x; idx < tot; idx += WG_THREADS) {
    int r = (int)(idx % rows);
    long c = idx / rows;
    m = fmax(m, fabs(A[r + ld * c]));
  }
  return wg_max(m, red);
}

// ---------------------------------------------------------------------------------------------
// Blocked Householder QR, R only, in place.
//   Y: column-major, leading dim ld (multiple of 32, >= rows rounded up to 32); the padding rows
//      [rows, rows32) and the padding columns [cols, cols16 + 16) MUST exist and be zero on entry
//      (cols16 = cols rounded up to 16; the extra 16 columns absorb tiles that start unaligned after a
//      short last panel).
//   On exit the upper trapezoid Y[0:min(rows,cols), 0:cols] holds R (everything below is garbage).
//   lds: QR_LDS_DOUBLES doubles.
// Panel (16 columns): LAPACK dgeqr2 + dlarft.  Fast path: the panel lives in registers (4 rows per
// thread, rows <= 2048 below the diagonal block) and each column costs ONE workgroup reduction of 16
// values (|x|^2 and the 15 dot products, taken before scaling); generic path: level 2 in global memory.
// Trailing update C -= V (T^T (V^T C)) per 16-column tile inside one wave, entirely in MFMA registers:
// W0 = V^T C (k-permuted 32-B row fragments = full 128-B lines), W = T^T W0 (accumulator fed back as
// B operand), C^T -= W^T V^T (so that every C access is a full line); global loads are issued four row
// blocks ahead of the MFMAs that consume them.
// ---------------------------------------------------------------------------------------------
constexpr int QR_NB = 16;
constexpr int QR_RS = 4;      // rows per thread of the register panel
constexpr int QR_TC = 3;      // column tiles a wave updates together
constexpr int QR_LDS_BASE = 2 * WG_WAVES * 16 + 16 * 16 + 16 + 2 * 32 + 2 * 16;   // [red x2][Ts][tau][bc]
constexpr int QR_ROWPAR_TILES = 1;   // trailing tiles at or below which a 4-panel update runs row-parallel (one tile at a time)
constexpr bool QR_QUAD = (WG_THREADS == 512);   // aggregate four panels (K = 64 trailing updates) - 512-thread variant only
constexpr int QR_LDS_PAIR = QR_LDS_BASE + 256 + 256 + 16;
#ifndef QR_TRAIL_PER_WAVE
#define QR_TRAIL_PER_WAVE 0      // 1: every wave streams its own copy of the reflector panels (the round-1 trailing update)
#endif
constexpr int QR_LDS_DOUBLES = QR_LDS_PAIR + (QR_QUAD ? 7 * 256 + 32 : 0);

// Cross-lane exchanges without the LDS crossbar (ds_bpermute): gfx950's v_permlane32_swap / v_permlane16_swap do a
// whole butterfly level in one instruction per 32-bit half, DPP row_mirror / row_half_mirror / quad_perm the rest.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// lanes < 32: x(l) + x(l+32);  lanes >= 32: y(l-32) + y(l)
__device__ __forceinline__ double swap_add32(double x, double y) {
  const u32x2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const u32x2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// within every 32 lanes: lanes with bit 4 clear: x(l) + x(l+16);  bit 4 set: y(l-16) + y(l)
__device__ __forceinline__ double swap_add16(double x, double y) {
  const u32x2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const u32x2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// DPP move of a double: 0x140 row_mirror (l <-> 15-l), 0x141 row_half_mirror (l <-> 7-l), 0x4E / 0xB1 quad xor 2 / 1
template <int CTRL>
__device__ __forceinline__ double dpp64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// 16 per-lane partial values -> lanes with (lane & 3) == 0 hold the wave total of value
// idx = ((lane>>5)&1)<<3 | ((lane>>4)&1)<<2 | ((lane>>3)&1)<<1 | ((lane>>2)&1).  Each level pairs the lanes whose
// bit differs (by a mirror below 16 lanes: any pairing works, the later levels sum over all lower bits).
__device__ __forceinline__ double wave_sum16(const double (&v)[16], int lane, int& idx) {
  double a8[8], a4[4], a2[2], a1;
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
  for (int i = 0; i < 8; i++) a8[i] = swap_add32(v[i], v[i + 8]);
#pragma unroll
  for (int i = 0; i < 4; i++) a4[i] = swap_add16(a8[i], a8[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; i++) {
    double keep = b3 ? a4[i + 2] : a4[i], send = b3 ? a4[i] : a4[i + 2];
    a2[i] = keep + dpp64<0x140>(send);
  }
  {
    double keep = b2 ? a2[1] : a2[0], send = b2 ? a2[0] : a2[1];
    a1 = keep + dpp64<0x141>(send);
  }
  a1 += dpp64<0x4E>(a1);
  a1 += dpp64<0xB1>(a1);
  idx = (b5 ? 8 : 0) | (b4 ? 4 : 0) | (b3 ? 2 : 0) | (b2 ? 1 : 0);
  return a1;
}

// 8 per-lane partial values -> lanes with (lane & 7) == 0 hold the wave total of value
// idx = ((lane>>5)&1)<<2 | ((lane>>4)&1)<<1 | ((lane>>3)&1)
__device__ __forceinline__ double wave_sum8(const double (&v)[8], int lane, int& idx) {
  double a4[4], a2[2], a1;
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
  for (int i = 0; i < 4; i++) a4[i] = swap_add32(v[i], v[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; i++) a2[i] = swap_add16(a4[i], a4[i + 2]);
  {
    double keep = b3 ? a2[1] : a2[0], send = b3 ? a2[0] : a2[1];
    a1 = keep + dpp64<0x140>(send);
  }
  a1 += dpp64<0x141>(a1);
  a1 += dpp64<0x4E>(a1);
  a1 += dpp64<0xB1>(a1);
  idx = (b5 ? 4 : 0) | (b4 ? 2 : 0) | (b3 ? 1 : 0);
  return a1;
}

// One column step of the register panel with the column index JJ a compile-time constant: only the
// columns c >= JJ are touched (half the work of a fixed 16-wide step, no register shifting).
template <int JJ>
__device__ __forceinline__ void qr_panel_step(double (&P)[QR_RS][QR_NB], const bool (&rv)[QR_RS], gdbl* Y, long ld,
                                              int j0, ldbl* red, ldbl* tau, ldbl* tot, ldbl* rowb) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  (void)tot;
  ldbl* redj = red + WG_WAVES * 16 * (JJ & 1);   // partials and pivot row are double buffered by column parity:
  ldbl* rowj = rowb + 16 * (JJ & 1);             // ONE barrier per column
  constexpr int NV = QR_NB - JJ;           // values to reduce: |x|^2 and NV-1 dots
  if (NV > 8) {
    double vals[16];
#pragma unroll
    for (int c = 0; c < 16; c++) vals[c] = 0.0;
#pragma unroll
    for (int s = 0; s < QR_RS; s++) {
      const bool below = rv[s] && (s > 0 || tid > JJ);
      const double x = below ? P[s][JJ] : 0.0;
#pragma unroll
      for (int c = JJ; c < QR_NB; c++) vals[c - JJ] += x * P[s][c];
    }
    int idx;
    const double wsum = wave_sum16(vals, lane, idx);
    if ((lane & 3) == 0) redj[wave * 16 + idx] = wsum;
  } else {
    double vals[8];
#pragma unroll
    for (int c = 0; c < 8; c++) vals[c] = 0.0;
#pragma unroll
    for (int s = 0; s < QR_RS; s++) {
      const bool below = rv[s] && (s > 0 || tid > JJ);
      const double x = below ? P[s][JJ] : 0.0;
#pragma unroll
      for (int c = JJ; c < QR_NB; c++) vals[c - JJ] += x * P[s][c];
    }
    int idx;
    const double wsum = wave_sum8(vals, lane, idx);
    if ((lane & 7) == 0) redj[wave * 16 + idx] = wsum;
  }
  if (tid == JJ) {
#pragma unroll
    for (int c = JJ; c < QR_NB; c++) rowj[c - JJ] = P[0][c];
  }
  __syncthreads();
  // every wave sums the partials itself (lane c takes value c) and hands the totals round as wave-uniform values:
  // no second barrier, no LDS round trip
  double sacc = 0.0;
  {
    const int c = lane & 15;
#pragma unroll
    for (int w = 0; w < WG_WAVES; w++) sacc += redj[w * 16 + c];
  }
  double totj[QR_NB];
#pragma unroll
  for (int c = 0; c < NV; c++)
    totj[c] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(sacc), c), __builtin_amdgcn_readlane(__double2loint(sacc), c));
  const double ss = totj[0];
  const double alpha = rowj[0];
  double beta, tj, scale;
  if (ss == 0.0) { beta = alpha; tj = 0.0; scale = 0.0; }
  else {
    // dlarfg with one rsq and one rcp (hardware seeds + Newton steps) instead of a sqrt and two IEEE divisions:
    // every thread of the workgroup evaluates this chain on the critical path of each column
    const double n2 = alpha * alpha + ss;
    double ri = __builtin_amdgcn_rsq(n2);
    ri = ri * (1.5 - 0.5 * n2 * ri * ri);
    ri = ri * (1.5 - 0.5 * n2 * ri * ri);              // 1 / ||x||
    double nrm = n2 * ri;
    nrm = nrm + 0.5 * ri * (n2 - nrm * nrm);           // ||x||
    beta = -copysign(nrm, alpha);
    tj = 1.0 + fabs(alpha) * ri;                       // (beta - alpha) / beta = 1 - alpha / beta
    const double dd = alpha - beta;                    // same sign as alpha, |dd| = |alpha| + ||x||: no cancellation
    double rd = __builtin_amdgcn_rcp(dd);
    rd = rd * (2.0 - dd * rd);
    rd = rd * (2.0 - dd * rd);
    scale = rd;
  }
  if (tid == 0) tau[JJ] = tj;
  double tw[QR_NB];
#pragma unroll
  for (int c = JJ + 1; c < QR_NB; c++) tw[c] = tj * (rowj[c - JJ] + scale * totj[c - JJ]);
#pragma unroll
  for (int s = 0; s < QR_RS; s++) {
    const bool below = rv[s] && (s > 0 || tid > JJ);
    const bool pivot = (s == 0) && (tid == JJ);
    const double v = below ? P[s][JJ] * scale : (pivot ? 1.0 : 0.0);
    if (below) P[s][JJ] = v;
    if (pivot) P[s][JJ] = beta;
#pragma unroll
    for (int c = JJ + 1; c < QR_NB; c++) P[s][c] -= tw[c] * v;
  }
  const long coff = (long)(j0 + JJ) * ld;
#pragma unroll
  for (int s = 0; s < QR_RS; s++)
    if (rv[s]) Y[coff + j0 + tid + WG_THREADS * s] = P[s][JJ];
}

template <int JJ>
__device__ __forceinline__ void qr_panel_steps(double (&P)[QR_RS][QR_NB], const bool (&rv)[QR_RS], gdbl* Y, long ld,
                                               int j0, int nb, ldbl* red, ldbl* tau, ldbl* tot, ldbl* rowb) {
  if constexpr (JJ < QR_NB) {
    if (JJ < nb) qr_panel_step<JJ>(P, rv, Y, ld, j0, red, tau, tot, rowb);
    qr_panel_steps<JJ + 1>(P, rv, Y, ld, j0, nb, red, tau, tot, rowb);
  }
}

// register-resident panel factorisation (rows - j0 <= QR_RS * WG_THREADS); T is built afterwards from the
// Gram matrix of V (qr_gram + qr_T_from_gram).
__device__ __attribute__((noinline)) void qr_panel_regs(gdbl* Y, long ld, int rows, int j0, int nb, ldbl* red,
                                              ldbl* tau, ldbl* bc) {
  const int tid = threadIdx.x;
  double P[QR_RS][QR_NB];
  bool rv[QR_RS];
#pragma unroll
  for (int s = 0; s < QR_RS; s++) {
    const int r = j0 + tid + WG_THREADS * s;
    rv[s] = r < rows;
#pragma unroll
    for (int c = 0; c < QR_NB; c++) P[s][c] = (rv[s] && c < nb) ? Y[(long)(j0 + c) * ld + r] : 0.0;
  }
  ldbl* tot = bc;            // [2][16] wave-summed totals (double buffered by column parity)
  ldbl* rowb = bc + 32;      // [2][16] the pivot row of the panel
  qr_panel_steps<0>(P, rv, Y, ld, j0, nb, red, tau, tot, rowb);
  __syncthreads();
}

// G1[i + 16 j] = sum_r Vx[r][i] Vy[r][j]  and (if TWO) G2[i + 16 j] = sum_r Vx[r][i] Vz[r][j]  over the rows
// from `jrow` (a multiple of 16) by MFMA: every wave takes the row blocks rb = wave, wave+8, ... with the loads
// of the next block in flight; partials are reduced through `big` (>= WG_WAVES*512 doubles of LDS).
// Vx / Vy / Vz are panels stored in place (unit lower-trapezoidal heads at column offsets jx / jy / jz with
// nbx / nby / nbz reflectors).  Results in big[0..255] and big[256..511]; ends with a barrier.
template <bool TWO>
__device__ __forceinline__ void qr_gram(const gdbl* Y, long ld, int rows32, int jrow, int jx, int nbx, int jy,
                                        int nby, int jz, int nbz, ldbl* big) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - jrow) >> 4;
  const gdbl* xcol = Y + (long)(jx + l15) * ld + jrow + 4 * g;
  const gdbl* ycol = Y + (long)(jy + l15) * ld + jrow + 4 * g;
  const gdbl* zcol = Y + (long)(jz + l15) * ld + jrow + 4 * g;
  d4 acc1 = d4{0, 0, 0, 0}, acc2 = d4{0, 0, 0, 0};
  auto head = [&](d4& v, int rb, int jv, int nbv) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rr = jrow + 16 * rb + 4 * g + e - jv;       // relative to the panel's diagonal block
      double a = v[e];
      a = (rr < 16) ? ((rr > l15) ? a : ((rr == l15) ? 1.0 : 0.0)) : a;
      v[e] = (l15 < nbv && rr >= 0 && rb < nrb) ? a : 0.0;
    }
  };
  d4 vx[2], vy[2], vz[2];
  auto load = [&](int rb, d4& a, d4& b, d4& c) {
    const int rbc = min(rb, nrb - 1);
    a = *reinterpret_cast<const gd4*>(xcol + 16 * rbc);
    b = *reinterpret_cast<const gd4*>(ycol + 16 * rbc);
    if (TWO) c = *reinterpret_cast<const gd4*>(zcol + 16 * rbc);
  };
  auto comp = [&](int rb, d4& a, d4& b, d4& c) {
    head(a, rb, jx, nbx); head(b, rb, jy, nby);
    if (TWO) head(c, rb, jz, nbz);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      acc1 = mfma(a[e], b[e], acc1);
      if (TWO) acc2 = mfma(a[e], c[e], acc2);
    }
  };
  if (wave < nrb) load(wave, vx[0], vy[0], vz[0]);
  for (int rb = wave; rb < nrb; rb += 2 * WG_WAVES) {
    load(rb + WG_WAVES, vx[1], vy[1], vz[1]);
    comp(rb, vx[0], vy[0], vz[0]);
    load(rb + 2 * WG_WAVES, vx[0], vy[0], vz[0]);
    comp(rb + WG_WAVES, vx[1], vy[1], vz[1]);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    big[wave * 512 + (g + 4 * r) + 16 * l15] = acc1[r];
    if (TWO) big[wave * 512 + 256 + (g + 4 * r) + 16 * l15] = acc2[r];
  }
  __syncthreads();
  if (WG_WAVES > 1) {
    constexpr int NVAL = TWO ? 512 : 256;
    constexpr int PER = (NVAL + WG_THREADS - 1) / WG_THREADS;
    double s1[PER];
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int idx = tid + e * WG_THREADS;
      s1[e] = 0.0;
      if (idx < NVAL) {
#pragma unroll
        for (int w = 0; w < WG_WAVES; w++) s1[e] += big[w * 512 + idx];
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int idx = tid + e * WG_THREADS;
      if (idx < NVAL) big[idx] = s1[e];
    }
    __syncthreads();
  }
}

// NP Gram blocks of panel x against panels y_0 .. y_{NP-1} in ONE pass over the rows (x is read once instead of
// NP times): G_k[i + 16 j] = sum_r Vx[r][i] Vy_k[r][j], rows from `jrow` (a multiple of 16; x is zero above).
// Results in big[256 k ...]; partials use WG_WAVES * 256 * NP doubles of `big`.  All panels full (16 reflectors).
template <int NP>
__device__ __forceinline__ void qr_gramN(const gdbl* Y, long ld, int rows32, int jrow, int jx, const int (&jy)[NP],
                                         ldbl* big) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - jrow) >> 4;
  const gdbl* xcol = Y + (long)(jx + l15) * ld + jrow + 4 * g;
  const gdbl* ycol[NP];
  d4 acc[NP];
#pragma unroll
  for (int k = 0; k < NP; k++) { ycol[k] = Y + (long)(jy[k] + l15) * ld + jrow + 4 * g; acc[k] = d4{0, 0, 0, 0}; }
  auto head = [&](d4& v, int rb, int jv) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rr = jrow + 16 * rb + 4 * g + e - jv;       // relative to the panel's diagonal block
      double a = v[e];
      a = (rr < 16) ? ((rr > l15) ? a : ((rr == l15) ? 1.0 : 0.0)) : a;
      v[e] = (rr >= 0 && rb < nrb) ? a : 0.0;
    }
  };
  d4 vx[2], vy[2][NP];
  auto load = [&](int rb, d4& a, d4 (&b)[NP]) {
    const int rbc = min(rb, nrb - 1);
    a = *reinterpret_cast<const gd4*>(xcol + 16 * rbc);
#pragma unroll
    for (int k = 0; k < NP; k++) b[k] = *reinterpret_cast<const gd4*>(ycol[k] + 16 * rbc);
  };
  auto comp = [&](int rb, d4& a, d4 (&b)[NP]) {
    head(a, rb, jx);
#pragma unroll
    for (int k = 0; k < NP; k++) {
      head(b[k], rb, jy[k]);
#pragma unroll
      for (int e = 0; e < 4; e++) acc[k] = mfma(a[e], b[k][e], acc[k]);
    }
  };
  if (wave < nrb) load(wave, vx[0], vy[0]);
  for (int rb = wave; rb < nrb; rb += 2 * WG_WAVES) {
    load(rb + WG_WAVES, vx[1], vy[1]);
    comp(rb, vx[0], vy[0]);
    load(rb + 2 * WG_WAVES, vx[0], vy[0]);
    comp(rb + WG_WAVES, vx[1], vy[1]);
  }
  constexpr int NVAL = 256 * NP;
#pragma unroll
  for (int k = 0; k < NP; k++)
#pragma unroll
    for (int r = 0; r < 4; r++) big[wave * NVAL + 256 * k + (g + 4 * r) + 16 * l15] = acc[k][r];
  __syncthreads();
  if (WG_WAVES > 1) {
    constexpr int PER = (NVAL + WG_THREADS - 1) / WG_THREADS;
    double s1[PER];
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int idx = tid + e * WG_THREADS;
      s1[e] = 0.0;
      if (idx < NVAL) {
#pragma unroll
        for (int w2 = 0; w2 < WG_WAVES; w2++) s1[e] += big[w2 * NVAL + idx];
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; e++) {
      const int idx = tid + e * WG_THREADS;
      if (idx < NVAL) big[idx] = s1[e];
    }
    __syncthreads();
  }
}

// T of the block reflector (dlarft) from the Gram matrix G = V^T V in `big`: row i of T depends only on its own
// earlier entries, so lane i builds row i in registers with no synchronisation:
//   T(i,j) = -tau_j sum_{i2=i}^{j-1} T(i,i2) G(i2,j)  (i < j),  T(j,j) = tau_j.
__device__ __forceinline__ void qr_T_from_gram(const ldbl* big, const ldbl* tau, int nb, ldbl* Ts) {
  const int tid = threadIdx.x;
  if (tid < 16) {
    double trow[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      // column j of G and tau_j are loaded unconditionally, before the dependent sum: the loads of the next column
      // overlap the chain of this one.  (The select on i2 >= tid stays: G entries of unused columns need not be finite.)
      double gcol[16];
#pragma unroll
      for (int i2 = 0; i2 < 16; i2++) gcol[i2] = big[i2 + 16 * j];
      const double tj = tau[j];
      double sacc = 0.0;
#pragma unroll
      for (int i2 = 0; i2 < j; i2++) sacc += (i2 >= tid) ? trow[i2] * gcol[i2] : 0.0;
      double v = (j == tid) ? tj : ((j > tid) ? -tj * sacc : 0.0);
      trow[j] = (j < nb) ? v : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) Ts[tid + 16 * j] = trow[j];
  }
  __syncthreads();
}

// generic panel factorisation, panel in global memory (any number of rows)
__device__ __attribute__((noinline)) void qr_panel_global(gdbl* Y, long ld, int rows, int j0, int nb, ldbl* red, ldbl* Ts,
                                       ldbl* tau, ldbl* bc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int jj = 0; jj < nb; jj++) {
    const int col = j0 + jj;
    gdbl* x = Y + (long)col * ld;
    double ss = 0.0;
    for (int r = col + 1 + tid; r < rows; r += WG_THREADS) { double v = x[r]; ss += v * v; }
    ss = wg_sum(ss, red);
    const double alpha = x[col];
    double beta, tj, scale;
    if (ss == 0.0) { beta = alpha; tj = 0.0; scale = 0.0; }
    else {
      beta = -copysign(sqrt(alpha * alpha + ss), alpha);
      tj = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    __syncthreads();   // everyone has read alpha
    for (int r = col + 1 + tid; r < rows; r += WG_THREADS) x[r] *= scale;
    if (tid == 0) { x[col] = beta; tau[jj] = tj; }
    __syncthreads();
    double part[QR_NB];
#pragma unroll
    for (int c = 0; c < QR_NB; c++) part[c] = 0.0;
    for (int r = col + 1 + tid; r < rows; r += WG_THREADS) {
      const double v = x[r];
#pragma unroll
      for (int c = 0; c < QR_NB; c++)
        if (c != jj && c < nb) part[c] += v * Y[(long)(j0 + c) * ld + r];
    }
#pragma unroll
    for (int c = 0; c < QR_NB; c++) part[c] = wave_sum(part[c]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < QR_NB; c++) red[wave * 16 + c] = part[c];
    }
    __syncthreads();
    if (tid < 16) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < WG_WAVES; w++) s += red[w * 16 + tid];
      if (tid < nb && tid != jj) s += Y[(long)(j0 + tid) * ld + col];
      bc[tid] = s;
    }
    __syncthreads();
    for (int r = col + 1 + tid; r < rows; r += WG_THREADS) {
      const double v = x[r];
      for (int c = jj + 1; c < nb; c++) Y[(long)(j0 + c) * ld + r] -= tj * bc[c] * v;
    }
    if (tid > jj && tid < nb) Y[(long)(j0 + tid) * ld + col] -= tj * bc[tid];
    if (tid < jj) {
      double s = 0.0;
      for (int i2 = tid; i2 < jj; i2++) s += Ts[tid + 16 * i2] * bc[i2];
      Ts[tid + 16 * jj] = -tj * s;
    }
    if (tid == 0) Ts[jj + 16 * jj] = tj;
    __syncthreads();
  }
}

// Trailing update of NT adjacent 16-column tiles (first column cb0) by one wave - branch free, so that the
// compiler keeps the global loads in flight behind counted vmcnt waits (any predicated load becomes a branch
// and a vmcnt(0)).  Requires: ld % 32 == 0, rows padded with zeros up to rows32, and the buffer to have
// zero padding columns wherever a tile sticks out of `cols` (they stay zero: W = T^T V^T 0 = 0).
template <int NT>
__device__ __forceinline__ void qr_trail(gdbl* Y, long ld, int rows32, int j0, int nb, int cb0, const ldbl* Ts) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - j0) >> 4;
  const gdbl* vcol = Y + (long)(j0 + l15) * ld + j0 + 4 * g;
  const gdbl* ccol[NT];
  d4 w0[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) { ccol[q] = Y + (long)(cb0 + 16 * q + l15) * ld + j0 + 4 * g; w0[q] = d4{0, 0, 0, 0}; }
  const bool vkeep = l15 < nb;
  // ---- phase A: W0[i][c] = sum_r V[r][i] C[r][c]   (3-stage register ring, loads 2 row blocks ahead)
  {
    d4 sv[3], sc[3][NT];
    auto loadA = [&](int rb, d4& v, d4 (&c)[NT]) {
      const int rbc = min(rb, nrb - 1);
      v = *reinterpret_cast<const gd4*>(vcol + 16 * rbc);
#pragma unroll
      for (int q = 0; q < NT; q++) c[q] = *reinterpret_cast<const gd4*>(ccol[q] + 16 * rbc);
      const bool keep = vkeep && rb < nrb;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int rho = 4 * g + e;
        double x = v[e];
        x = (rb == 0) ? ((rho > l15) ? x : ((rho == l15) ? 1.0 : 0.0)) : x;   // unit lower-trapezoidal head
        v[e] = keep ? x : 0.0;
      }
    };
    auto compA = [&](const d4& v, const d4 (&c)[NT]) {
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int q = 0; q < NT; q++) w0[q] = mfma(v[e], c[q][e], w0[q]);
    };
    loadA(0, sv[0], sc[0]);
    loadA(1, sv[1], sc[1]);
    for (int rb = 0; rb < nrb; rb += 3) {
      loadA(rb + 2, sv[2], sc[2]); compA(sv[0], sc[0]);
      loadA(rb + 3, sv[0], sc[0]); compA(sv[1], sc[1]);
      loadA(rb + 4, sv[1], sc[1]); compA(sv[2], sc[2]);
    }
  }
  // ---- phase B: W = T^T W0   (A[i'][k] = T[k][i'], B k-step s = reg s of W0)
  d4 w[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    w[q] = d4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) w[q] = mfma(Ts[(4 * s + g) + 16 * l15], w0[q][s], w[q]);
  }
  // ---- phase C: C^T[c][r] -= sum_k W[k][c] V[r][k].  Stages of 32 rows aligned to 32: lane l15 owns rows
  //      (2*l15, 2*l15+1) of the stage -> every access is 16 B/lane and a full 128-B line per column.
  {
    const int jb = j0 & ~31;
    const int nst = (rows32 - jb) >> 5;
    for (int st = 0; st < nst; st++) {
      const int row = jb + 32 * st + 2 * l15;
      d2 v[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) {
        const int k = 4 * s2 + g;               // reflector index
        d2 x = *reinterpret_cast<const gd2*>(Y + (long)(j0 + k) * ld + row);
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int rho = row + e - j0;         // row relative to the panel's diagonal block
          double xe = x[e];
          xe = (rho < 16) ? ((rho > k) ? xe : ((rho == k) ? 1.0 : 0.0)) : xe;
          x[e] = (k < nb && rho >= 0) ? xe : 0.0;
        }
        v[s2] = x;
      }
      d2 c[NT][4];
#pragma unroll
      for (int q = 0; q < NT; q++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          c[q][r] = *reinterpret_cast<const gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row);
#pragma unroll
      for (int q = 0; q < NT; q++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          d4 acc = d4{c[q][0][e], c[q][1][e], c[q][2][e], c[q][3][e]};
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) acc = mfma_na(w[q][s2], v[s2][e], acc);
#pragma unroll
          for (int r = 0; r < 4; r++) c[q][r][e] = acc[r];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
          *reinterpret_cast<gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row) = c[q][r];
      }
    }
  }
}

// Trailing update by TWO adjacent panels a (columns j0..j0+15) and b (j0+16..j0+31, nbb reflectors) in one
// pass over the rows (half the HBM traffic of two single-panel passes):
//   (I - Vb Tb^T Vb^T)(I - Va Ta^T Va^T) C = C - Va Wa - Vb Wb,
//   Wa = Ta^T Va^T C,   Wb = Tb^T (Vb^T C - S Wa),   S = Vb^T Va  (16x16, in LDS).
template <int NT>
__device__ __forceinline__ void qr_trail2(gdbl* Y, long ld, int rows32, int j0, int nbb, int cb0,
                                          const ldbl* TsA, const ldbl* TsB, const ldbl* Sm) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - j0) >> 4;
  const gdbl* vacol = Y + (long)(j0 + l15) * ld + j0 + 4 * g;
  const gdbl* vbcol = Y + (long)(j0 + 16 + l15) * ld + j0 + 4 * g;
  const gdbl* ccol[NT];
  d4 wa0[NT], wb0[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    ccol[q] = Y + (long)(cb0 + 16 * q + l15) * ld + j0 + 4 * g;
    wa0[q] = d4{0, 0, 0, 0}; wb0[q] = d4{0, 0, 0, 0};
  }
  const bool bkeep = l15 < nbb;
  {
    d4 sa[3], sb[3], sc[3][NT];
    auto loadA = [&](int rb, d4& va, d4& vb, d4 (&c)[NT]) {
      const int rbc = min(rb, nrb - 1);
      va = *reinterpret_cast<const gd4*>(vacol + 16 * rbc);
      vb = *reinterpret_cast<const gd4*>(vbcol + 16 * rbc);
#pragma unroll
      for (int q = 0; q < NT; q++) c[q] = *reinterpret_cast<const gd4*>(ccol[q] + 16 * rbc);
      const bool in = rb < nrb;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int rho = 4 * g + e;
        double a = va[e], b = vb[e];
        a = (rb == 0) ? ((rho > l15) ? a : ((rho == l15) ? 1.0 : 0.0)) : a;
        b = (rb == 1) ? ((rho > l15) ? b : ((rho == l15) ? 1.0 : 0.0)) : b;
        va[e] = in ? a : 0.0;
        vb[e] = (in && bkeep && rb >= 1) ? b : 0.0;
      }
    };
    auto compA = [&](const d4& va, const d4& vb, const d4 (&c)[NT]) {
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int q = 0; q < NT; q++) { wa0[q] = mfma(va[e], c[q][e], wa0[q]); wb0[q] = mfma(vb[e], c[q][e], wb0[q]); }
    };
    loadA(0, sa[0], sb[0], sc[0]);
    loadA(1, sa[1], sb[1], sc[1]);
    for (int rb = 0; rb < nrb; rb += 3) {
      loadA(rb + 2, sa[2], sb[2], sc[2]); compA(sa[0], sb[0], sc[0]);
      loadA(rb + 3, sa[0], sb[0], sc[0]); compA(sa[1], sb[1], sc[1]);
      loadA(rb + 4, sa[1], sb[1], sc[1]); compA(sa[2], sb[2], sc[2]);
    }
  }
  // phase B
  d4 wa[NT], wb[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    wa[q] = d4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) wa[q] = mfma(TsA[(4 * s + g) + 16 * l15], wa0[q][s], wa[q]);
    d4 t = wb0[q];                          // t = Vb^T C - S Wa
#pragma unroll
    for (int s = 0; s < 4; s++) t = mfma_na(Sm[l15 + 16 * (4 * s + g)], wa[q][s], t);
    wb[q] = d4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) wb[q] = mfma(TsB[(4 * s + g) + 16 * l15], t[s], wb[q]);
  }
  // phase C
  {
    const int jb = j0 & ~31;
    const int nst = (rows32 - jb) >> 5;
    for (int st = 0; st < nst; st++) {
      const int row = jb + 32 * st + 2 * l15;
      d2 va[4], vb[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) {
        const int k = 4 * s2 + g;
        d2 xa = *reinterpret_cast<const gd2*>(Y + (long)(j0 + k) * ld + row);
        d2 xb = *reinterpret_cast<const gd2*>(Y + (long)(j0 + 16 + k) * ld + row);
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int ra = row + e - j0, rbb = row + e - j0 - 16;
          double a = xa[e], b = xb[e];
          a = (ra < 16) ? ((ra > k) ? a : ((ra == k) ? 1.0 : 0.0)) : a;
          b = (rbb < 16) ? ((rbb > k) ? b : ((rbb == k) ? 1.0 : 0.0)) : b;
          xa[e] = (ra >= 0) ? a : 0.0;
          xb[e] = (k < nbb && rbb >= 0) ? b : 0.0;
        }
        va[s2] = xa; vb[s2] = xb;
      }
      d2 c[NT][4];
#pragma unroll
      for (int q = 0; q < NT; q++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          c[q][r] = *reinterpret_cast<const gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row);
#pragma unroll
      for (int q = 0; q < NT; q++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          d4 acc = d4{c[q][0][e], c[q][1][e], c[q][2][e], c[q][3][e]};
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) { acc = mfma_na(wa[q][s2], va[s2][e], acc); acc = mfma_na(wb[q][s2], vb[s2][e], acc); }
#pragma unroll
          for (int r = 0; r < 4; r++) c[q][r][e] = acc[r];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
          *reinterpret_cast<gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row) = c[q][r];
      }
    }
  }
}

// Update ONE 16-column tile (first column cb0) by panel (j0, nb) with all waves working on different rows:
// W0 partials -> LDS, every wave then applies W = T^T W0 to its own 32-row stages.
__device__ __forceinline__ void qr_tile_update_all(gdbl* Y, long ld, int rows32, int j0, int nb, int cb0,
                                                   const ldbl* Ts, ldbl* big) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - j0) >> 4;
  const gdbl* vcol = Y + (long)(j0 + l15) * ld + j0 + 4 * g;
  const gdbl* ccol = Y + (long)(cb0 + l15) * ld + j0 + 4 * g;
  d4 acc = d4{0, 0, 0, 0};
  for (int rb = wave; rb < nrb; rb += WG_WAVES) {
    d4 v = *reinterpret_cast<const gd4*>(vcol + 16 * rb);
    d4 c = *reinterpret_cast<const gd4*>(ccol + 16 * rb);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rho = 4 * g + e;
      double x = v[e];
      x = (rb == 0) ? ((rho > l15) ? x : ((rho == l15) ? 1.0 : 0.0)) : x;
      v[e] = (l15 < nb) ? x : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) acc = mfma(v[e], c[e], acc);
  }
#pragma unroll
  for (int r = 0; r < 4; r++) big[wave * 256 + (g + 4 * r) + 16 * l15] = acc[r];
  __syncthreads();
  // every wave sums the partials into its own W0 registers (C layout: row g+4r, col l15)
  d4 w0 = d4{0, 0, 0, 0};
#pragma unroll
  for (int w = 0; w < WG_WAVES; w++)
#pragma unroll
    for (int r = 0; r < 4; r++) w0[r] += big[w * 256 + (g + 4 * r) + 16 * l15];
  d4 wv = d4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 4; s++) wv = mfma(Ts[(4 * s + g) + 16 * l15], w0[s], wv);
  const int jb = j0 & ~31;
  const int nst = (rows32 - jb) >> 5;
  for (int st = wave; st < nst; st += WG_WAVES) {
    const int row = jb + 32 * st + 2 * l15;
    d2 v[4], c[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
      const int k = 4 * s2 + g;
      d2 x = *reinterpret_cast<const gd2*>(Y + (long)(j0 + k) * ld + row);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int rho = row + e - j0;
        double xe = x[e];
        xe = (rho < 16) ? ((rho > k) ? xe : ((rho == k) ? 1.0 : 0.0)) : xe;
        x[e] = (k < nb && rho >= 0) ? xe : 0.0;
      }
      v[s2] = x;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = *reinterpret_cast<const gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row);
#pragma unroll
    for (int e = 0; e < 2; e++) {
      d4 a = d4{c[0][e], c[1][e], c[2][e], c[3][e]};
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) a = mfma_na(wv[s2], v[s2][e], a);
#pragma unroll
      for (int r = 0; r < 4; r++) c[r][e] = a[r];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) *reinterpret_cast<gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row) = c[r];
  }
  __syncthreads();
}

// Update NT adjacent 16-column tiles (first column cb0) by the panel PAIR a (j0), b (j0+16) with all waves working
// on different rows (the row-parallel counterpart of qr_trail2): partial Va^T C, Vb^T C -> LDS, then every wave
// forms Wa, Wb and applies them to its own 32-row stages.  The V fragments are loaded once for all NT tiles.
// `big`: WG_WAVES * 512 * NT doubles.
template <int NT>
__device__ __forceinline__ void qr_tile_update2_all(gdbl* Y, long ld, int rows32, int j0, int cb0, const ldbl* TsA,
                                                    const ldbl* TsB, const ldbl* Sm, ldbl* big) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - j0) >> 4;
  const gdbl* vacol = Y + (long)(j0 + l15) * ld + j0 + 4 * g;
  const gdbl* vbcol = Y + (long)(j0 + 16 + l15) * ld + j0 + 4 * g;
  d4 acca[NT], accb[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) { acca[q] = d4{0, 0, 0, 0}; accb[q] = d4{0, 0, 0, 0}; }
  for (int rb = wave; rb < nrb; rb += WG_WAVES) {
    d4 va = *reinterpret_cast<const gd4*>(vacol + 16 * rb);
    d4 vb = *reinterpret_cast<const gd4*>(vbcol + 16 * rb);
    d4 c[NT];
#pragma unroll
    for (int q = 0; q < NT; q++) c[q] = *reinterpret_cast<const gd4*>(Y + (long)(cb0 + 16 * q + l15) * ld + j0 + 4 * g + 16 * rb);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rho = 4 * g + e;
      double a = va[e], b = vb[e];
      a = (rb == 0) ? ((rho > l15) ? a : ((rho == l15) ? 1.0 : 0.0)) : a;
      b = (rb == 1) ? ((rho > l15) ? b : ((rho == l15) ? 1.0 : 0.0)) : b;
      va[e] = a;
      vb[e] = (rb >= 1) ? b : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int q = 0; q < NT; q++) { acca[q] = mfma(va[e], c[q][e], acca[q]); accb[q] = mfma(vb[e], c[q][e], accb[q]); }
  }
  constexpr int NVAL = 512 * NT;
#pragma unroll
  for (int q = 0; q < NT; q++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      big[wave * NVAL + 512 * q + (g + 4 * r) + 16 * l15] = acca[q][r];
      big[wave * NVAL + 512 * q + 256 + (g + 4 * r) + 16 * l15] = accb[q][r];
    }
  __syncthreads();
  d4 wa[NT], wb[NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    d4 wa0 = d4{0, 0, 0, 0}, wb0 = d4{0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < WG_WAVES; w++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        wa0[r] += big[w * NVAL + 512 * q + (g + 4 * r) + 16 * l15];
        wb0[r] += big[w * NVAL + 512 * q + 256 + (g + 4 * r) + 16 * l15];
      }
    wa[q] = d4{0, 0, 0, 0}; wb[q] = d4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) wa[q] = mfma(TsA[(4 * s + g) + 16 * l15], wa0[s], wa[q]);
    d4 t = wb0;
#pragma unroll
    for (int s = 0; s < 4; s++) t = mfma_na(Sm[l15 + 16 * (4 * s + g)], wa[q][s], t);
#pragma unroll
    for (int s = 0; s < 4; s++) wb[q] = mfma(TsB[(4 * s + g) + 16 * l15], t[s], wb[q]);
  }
  const int jb = j0 & ~31;
  const int nst = (rows32 - jb) >> 5;
  for (int st = wave; st < nst; st += WG_WAVES) {
    const int row = jb + 32 * st + 2 * l15;
    d2 va[4], vb[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) {
      const int k = 4 * s2 + g;
      d2 xa = *reinterpret_cast<const gd2*>(Y + (long)(j0 + k) * ld + row);
      d2 xb = *reinterpret_cast<const gd2*>(Y + (long)(j0 + 16 + k) * ld + row);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int ra = row + e - j0, rbb = row + e - j0 - 16;
        double a = xa[e], b = xb[e];
        a = (ra < 16) ? ((ra > k) ? a : ((ra == k) ? 1.0 : 0.0)) : a;
        b = (rbb < 16) ? ((rbb > k) ? b : ((rbb == k) ? 1.0 : 0.0)) : b;
        xa[e] = (ra >= 0) ? a : 0.0;
        xb[e] = (rbb >= 0) ? b : 0.0;
      }
      va[s2] = xa; vb[s2] = xb;
    }
#pragma unroll
    for (int q = 0; q < NT; q++) {
      d2 c[4];
#pragma unroll
      for (int r = 0; r < 4; r++) c[r] = *reinterpret_cast<const gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        d4 acc = d4{c[0][e], c[1][e], c[2][e], c[3][e]};
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) { acc = mfma_na(wa[q][s2], va[s2][e], acc); acc = mfma_na(wb[q][s2], vb[s2][e], acc); }
#pragma unroll
        for (int r = 0; r < 4; r++) c[r][e] = acc[r];
      }
#pragma unroll
      for (int r = 0; r < 4; r++) *reinterpret_cast<gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row) = c[r];
    }
  }
  __syncthreads();
}

// Trailing update by FOUR adjacent full panels p = 0..3 (columns j0 + 16p) in one pass over the rows - a quarter of
// the HBM traffic of four single-panel passes:
//   C <- C - sum_p V_p W_p,   W_p = T_p^T (V_p^T C - sum_{r<p} S_pr W_r),   S_pr = V_p^T V_r  (16x16 blocks in LDS).
// Tq[p] = T_p; Sq = {S10, S20, S21, S30, S31, S32}.
template <int NT>
__device__ __forceinline__ void qr_trail4(gdbl* Y, long ld, int rows32, int j0, int cb0, const ldbl* const (&Tq)[4],
                                          const ldbl* const (&Sq)[6]) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - j0) >> 4;
  const gdbl* vcol[4];
#pragma unroll
  for (int p = 0; p < 4; p++) vcol[p] = Y + (long)(j0 + 16 * p + l15) * ld + j0 + 4 * g;
  const gdbl* ccol[NT];
  d4 w0[4][NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    ccol[q] = Y + (long)(cb0 + 16 * q + l15) * ld + j0 + 4 * g;
#pragma unroll
    for (int p = 0; p < 4; p++) w0[p][q] = d4{0, 0, 0, 0};
  }
  // ---- phase A: W0_p = V_p^T C  (2-stage register ring: the next row block is in flight behind the MFMAs)
  {
    d4 sv[2][4], sc[2][NT];
    auto loadA = [&](int rb, d4 (&v)[4], d4 (&c)[NT]) {
      const int rbc = min(rb, nrb - 1);
#pragma unroll
      for (int p = 0; p < 4; p++) v[p] = *reinterpret_cast<const gd4*>(vcol[p] + 16 * rbc);
#pragma unroll
      for (int q = 0; q < NT; q++) c[q] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(ccol[q] + 16 * rbc));
      const bool in = rb < nrb;
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int rho = 4 * g + e;
          double a = v[p][e];
          a = (rb == p) ? ((rho > l15) ? a : ((rho == l15) ? 1.0 : 0.0)) : a;   // unit lower-trapezoidal head
          v[p][e] = (in && rb >= p) ? a : 0.0;
        }
    };
    auto compA = [&](const d4 (&v)[4], const d4 (&c)[NT]) {
#pragma unroll
      for (int e = 0; e < 4; e++)
#pragma unroll
        for (int q = 0; q < NT; q++)
#pragma unroll
          for (int p = 0; p < 4; p++) w0[p][q] = mfma(v[p][e], c[q][e], w0[p][q]);
    };
    loadA(0, sv[0], sc[0]);
    for (int rb = 0; rb < nrb; rb += 2) {
      loadA(rb + 1, sv[1], sc[1]); compA(sv[0], sc[0]);
      loadA(rb + 2, sv[0], sc[0]); compA(sv[1], sc[1]);
    }
  }
  // ---- phase B: the W recurrence, tile by tile, in registers (T and S fragments from LDS)
  d4 w[4][NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      d4 t = w0[p][q];
#pragma unroll
      for (int r = 0; r < p; r++) {
        const ldbl* S = Sq[p * (p - 1) / 2 + r];
#pragma unroll
        for (int s = 0; s < 4; s++) t = mfma_na(S[l15 + 16 * (4 * s + g)], w[r][q][s], t);
      }
      d4 o = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(Tq[p][(4 * s + g) + 16 * l15], t[s], o);
      w[p][q] = o;
    }
  }
  // ---- phase C: C^T -= sum_p W_p^T V_p^T over 32-row stages (every access a full 128-B line per column)
  {
    const int jb = j0 & ~31;
    const int nst = (rows32 - jb) >> 5;
    for (int st = 0; st < nst; st++) {
      const int row = jb + 32 * st + 2 * l15;
      d2 v[4][4];
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          const int k = 4 * s2 + g;
          d2 x = *reinterpret_cast<const gd2*>(Y + (long)(j0 + 16 * p + k) * ld + row);
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const int rp = row + e - j0 - 16 * p;          // row relative to panel p's diagonal block
            double a = x[e];
            a = (rp < 16) ? ((rp > k) ? a : ((rp == k) ? 1.0 : 0.0)) : a;
            x[e] = (rp >= 0) ? a : 0.0;
          }
          v[p][s2] = x;
        }
      d2 c[NT][4];
#pragma unroll
      for (int q = 0; q < NT; q++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          c[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row));
#pragma unroll
      for (int q = 0; q < NT; q++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          d4 acc = d4{c[q][0][e], c[q][1][e], c[q][2][e], c[q][3][e]};
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
            for (int p = 0; p < 4; p++) acc = mfma_na(w[p][q][s2], v[p][s2][e], acc);
#pragma unroll
          for (int r = 0; r < 4; r++) c[q][r][e] = acc[r];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
          __builtin_nontemporal_store(c[q][r], reinterpret_cast<gd2*>(Y + (long)(cb0 + 16 * q + g + 4 * r) * ld + row));
      }
    }
  }
}

// Row-parallel counterpart of qr_trail4<1>: ONE 16-column tile updated by the four panels at j0 with all waves
// working on different rows (partials of V_p^T C through LDS).  Used for the last tiles of a factorisation, when
// there are fewer tiles than waves and the tile-per-wave split would leave most of the workgroup idle.
// `big`: WG_WAVES * 1024 doubles.  noinline on purpose: inlined into qr_r (which sits at the 256-register limit) one
// build of this function was observed to return wrong results although its source had not changed.
__device__ __attribute__((noinline)) void qr_tile_update4_all(gdbl* Y, long ld, int rows32, int j0, int cb0,
                                                    const ldbl* const (&Tq)[4], const ldbl* const (&Sq)[6], ldbl* big) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int nrb = (rows32 - j0) >> 4;
  const gdbl* vcol[4];
#pragma unroll
  for (int p = 0; p < 4; p++) vcol[p] = Y + (long)(j0 + 16 * p + l15) * ld + j0 + 4 * g;
  const gdbl* ccol = Y + (long)(cb0 + l15) * ld + j0 + 4 * g;
  d4 acc[4];
#pragma unroll
  for (int p = 0; p < 4; p++) acc[p] = d4{0, 0, 0, 0};
  for (int rb = wave; rb < nrb; rb += WG_WAVES) {
    d4 v[4];
#pragma unroll
    for (int p = 0; p < 4; p++) v[p] = *reinterpret_cast<const gd4*>(vcol[p] + 16 * rb);
    const d4 c = *reinterpret_cast<const gd4*>(ccol + 16 * rb);
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int rho = 4 * g + e;
        double a = v[p][e];
        a = (rb == p) ? ((rho > l15) ? a : ((rho == l15) ? 1.0 : 0.0)) : a;
        v[p][e] = (rb >= p) ? a : 0.0;
      }
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
      for (int p = 0; p < 4; p++) acc[p] = mfma(v[p][e], c[e], acc[p]);
  }
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int r = 0; r < 4; r++) big[wave * 1024 + 256 * p + (g + 4 * r) + 16 * l15] = acc[p][r];
  __syncthreads();
  d4 w[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    d4 t = d4{0, 0, 0, 0};
#pragma unroll
    for (int w2 = 0; w2 < WG_WAVES; w2++)
#pragma unroll
      for (int r = 0; r < 4; r++) t[r] += big[w2 * 1024 + 256 * p + (g + 4 * r) + 16 * l15];
#pragma unroll
    for (int r = 0; r < p; r++) {
      const ldbl* S = Sq[p * (p - 1) / 2 + r];
#pragma unroll
      for (int s = 0; s < 4; s++) t = mfma_na(S[l15 + 16 * (4 * s + g)], w[r][s], t);
    }
    d4 o = d4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) o = mfma(Tq[p][(4 * s + g) + 16 * l15], t[s], o);
    w[p] = o;
  }
  const int jb = j0 & ~31;
  const int nst = (rows32 - jb) >> 5;
  for (int st = wave; st < nst; st += WG_WAVES) {
    const int row = jb + 32 * st + 2 * l15;
    d2 v[4][4];
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) {
        const int k = 4 * s2 + g;
        d2 x = *reinterpret_cast<const gd2*>(Y + (long)(j0 + 16 * p + k) * ld + row);
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int rp = row + e - j0 - 16 * p;
          double a = x[e];
          a = (rp < 16) ? ((rp > k) ? a : ((rp == k) ? 1.0 : 0.0)) : a;
          x[e] = (rp >= 0) ? a : 0.0;
        }
        v[p][s2] = x;
      }
    d2 c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = *reinterpret_cast<const gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row);
#pragma unroll
    for (int e = 0; e < 2; e++) {
      d4 a4 = d4{c[0][e], c[1][e], c[2][e], c[3][e]};
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
        for (int p = 0; p < 4; p++) a4 = mfma_na(w[p][s2], v[p][s2][e], a4);
#pragma unroll
      for (int r = 0; r < 4; r++) c[r][e] = a4[r];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) *reinterpret_cast<gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row) = c[r];
  }
  __syncthreads();
}

// Trailing update by the FOUR panels at j0 of ALL tiles [cstart, cstart + 16 ntl) with the reflector panels shared
// through LDS: the eight waves of the workgroup walk the rows together, 32 at a time; every 32-row stage of the four
// panels (64 columns, heads already in unit-lower-trapezoidal form) is loaded from HBM ONCE by the whole workgroup and
// read by every wave as LDS fragments, instead of each wave streaming its own copy of V through the vector L1 (which
// made phase C L1-bound: 16 KB of V per wave and stage against 16 KB of C).  Every wave owns up to NT = 2 tiles per pass
// (contiguous runs, as wave_tiles); C goes straight from HBM to registers and back (nontemporal).
//   Vs: 2 * QR_VM_STAGE doubles of LDS (double buffer; QR_BIG_DOUBLES covers it).  Column stride == 4 banks (mod 64) per
//   column, so the phase A reads (16 lanes = 16 columns, 32 B each) and the phase C reads (16 lanes = 32 consecutive
//   rows of one column) are both conflict free.  One barrier per stage.
constexpr int QR_VS_LD = 34;                   // 32-row stages (batched kernels of v2_kernels.h)
constexpr int QR_VS_STAGE = 64 * QR_VS_LD;
constexpr int QR_BIG_DOUBLES = (WG_WAVES * 1024 > 2 * 64 * 66) ? WG_WAVES * 1024 : 2 * 64 * 66;   // `big` of qr_r
typedef __attribute__((address_space(3))) d2 ld2;
// One pass of qr_trail4_coop for a wave that owns NTL (0, 1 or 2: compile time, so that the stage loops are straight-line
// code the scheduler can pipeline - with a run-time tile count every group of MFMAs sat in its own basic block behind an
// s_waitcnt lgkmcnt(0)) tiles at columns cq0, cq1.  Every wave executes the same barriers whatever its NTL.
// The reflector panels are staged 64 rows at a time (QR_VM_*): ONE barrier per 64 rows; C is still fetched 32 rows
// ahead.  Every barrier makes all eight waves wait for the slowest C fetch: with 256 workgroups streaming, halving their
// number took the pass from 0.66 to 0.57 ms (tools/probes/trail_probe.hip).
constexpr int QR_VM_LD = 66;                   // == 4 banks (mod 64) per column, as QR_VS_LD
constexpr int QR_VM_STAGE = 64 * QR_VM_LD;
template <int NTL>
__device__ __forceinline__ void qr_trail4_coop_pass(gdbl* Y, long ld, int j0, int nst, int cq0, int cq1,
                                                    const ldbl* const (&Tq)[4], const ldbl* const (&Sq)[6], ldbl* Vs) {
  constexpr int NR = NTL > 0 ? NTL : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int nms = (nst + 1) >> 1;                         // 64-row stages; the second half of the last one may be empty
  // staging map: thread -> (column sc of the 64, rows 4*sr .. 4*sr+3 of each 32-row half)
  const int sc = tid >> 3, sr = tid & 7;
  const gdbl* vsrc = Y + (long)(j0 + sc) * ld + j0 + 4 * sr;
  const int spanel = sc >> 4, scol = sc & 15;
  auto stage_load = [&](int m, int h) -> d4 {
    const int s = min(2 * min(m, nms - 1) + h, nst - 1);
    return *reinterpret_cast<const gd4*>(vsrc + 32 * s);
  };
  auto stage_store = [&](int m, int h, d4 v) {
    ldbl* dst = Vs + (m & 1) * QR_VM_STAGE + sc * QR_VM_LD + 32 * h + 4 * sr;
    if (m == 0) {                                                    // only the first 64 rows hold the panels' diagonal blocks
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int rp = 32 * h + 4 * sr + e - 16 * spanel;            // row relative to this panel's diagonal block
        double a = v[e];
        a = (rp < 16) ? ((rp > scol) ? a : ((rp == scol) ? 1.0 : 0.0)) : a;
        v[e] = (rp >= 0) ? a : 0.0;
      }
    }
    // rows past the matrix: zero reflector rows.  A uniform test: inside the stage loops the store is two LDS writes and no
    // VALU instruction (the per-element masks above were ~40 VALU instructions per half stage beside 64 MFMAs).
    if (2 * m + h >= nst) v = d4{0, 0, 0, 0};
    *reinterpret_cast<ld2*>(dst) = d2{v[0], v[1]};
    *reinterpret_cast<ld2*>(dst + 2) = d2{v[2], v[3]};
  };
  const int cq[2] = {cq0, cq1};
  // ------------------------------------------------------------ phase A: W0_p = V_p^T C
  d4 w0[4][NR];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < NR; q++) w0[p][q] = d4{0, 0, 0, 0};
  {
    const gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + l15) * ld + j0 + 4 * g;
    {
      const d4 v0 = stage_load(0, 0), v1 = stage_load(0, 1);
      __syncthreads();                                    // the previous users of Vs are done
      stage_store(0, 0, v0); stage_store(0, 1, v1);
    }
    d4 cc[NR][2];                                         // [tile][row block of the 32 rows], 32 rows ahead
    if (NTL > 0) {
#pragma unroll
      for (int q = 0; q < NR; q++) {
        cc[q][0] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q]));
        cc[q][1] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 16));
      }
    }
    for (int m = 0; m < nms; m++) {
#pragma unroll 1
      for (int h = 0; h < 2; h++) {
        const d4 vn = stage_load(m + 1, h);               // half h of the next stage: stored when this half is done
        const int sn = min(2 * m + h + 1, nst - 1);
        d4 cn[NR][2];
        if (NTL > 0) {
#pragma unroll
          for (int q = 0; q < NR; q++) {
            cn[q][0] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn));
            cn[q][1] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn + 16));
          }
        }
        if (h == 0) lds_barrier();                        // stage m is in Vs[m & 1]; the prefetches stay in flight
        if (NTL > 0) {
          const ldbl* vb = Vs + (m & 1) * QR_VM_STAGE + 32 * h;
#pragma unroll
          for (int rb = 0; rb < 2; rb++) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
              const ldbl* vp = vb + (16 * p + l15) * QR_VM_LD + 16 * rb + 4 * g;
              const d2 va = *reinterpret_cast<const ld2*>(vp), vbb = *reinterpret_cast<const ld2*>(vp + 2);
              const double v4[4] = {va[0], va[1], vbb[0], vbb[1]};
#pragma unroll
              for (int e = 0; e < 4; e++)
#pragma unroll
                for (int q = 0; q < NR; q++) w0[p][q] = mfma(v4[e], cc[q][rb][e], w0[p][q]);
            }
          }
#pragma unroll
          for (int q = 0; q < NR; q++) { cc[q][0] = cn[q][0]; cc[q][1] = cn[q][1]; }
        }
        stage_store(m + 1, h, vn);                        // the other buffer: nobody reads it before the next barrier
      }
    }
  }
  // ------------------------------------------------------------ phase B: the W recurrence (registers)
  d4 w[4][NR];
  if (NTL > 0) {
#pragma unroll
    for (int q = 0; q < NR; q++) {
#pragma unroll
      for (int p = 0; p < 4; p++) {
        d4 t = w0[p][q];
#pragma unroll
        for (int r = 0; r < p; r++) {
          const ldbl* S = Sq[p * (p - 1) / 2 + r];
#pragma unroll
          for (int s = 0; s < 4; s++) t = mfma_na(S[l15 + 16 * (4 * s + g)], w[r][q][s], t);
        }
        d4 o = d4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; s++) o = mfma(Tq[p][(4 * s + g) + 16 * l15], t[s], o);
        w[p][q] = o;
      }
    }
  }
  // ------------------------------------------------------------ phase C: C^T -= sum_p W_p^T V_p^T
  {
    {
      const d4 v0 = stage_load(0, 0), v1 = stage_load(0, 1);
      __syncthreads();                                    // every wave has left phase A's last stage
      stage_store(0, 0, v0); stage_store(0, 1, v1);
    }
    gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + g) * ld + j0 + 2 * l15;
    d2 cc[NR][4];
    if (NTL > 0) {
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) cc[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(cp[q] + (long)(4 * r) * ld));
    }
    for (int m = 0; m < nms; m++) {
#pragma unroll 1
      for (int h = 0; h < 2; h++) {
        const d4 vn = stage_load(m + 1, h);
        const int s = 2 * m + h;
        const int sn = min(s + 1, nst - 1);
        d2 cn[NR][4];
        if (NTL > 0) {
#pragma unroll
          for (int q = 0; q < NR; q++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              cn[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(cp[q] + (long)(4 * r) * ld + 32 * sn));
        }
        if (h == 0) lds_barrier();
        if (NTL > 0) {
          if (s < nst) {
            const ldbl* vb = Vs + (m & 1) * QR_VM_STAGE + 32 * h;
            d4 acc[NR][2];                                // [tile][e]
#pragma unroll
            for (int q = 0; q < NR; q++)
#pragma unroll
              for (int e = 0; e < 2; e++) acc[q][e] = d4{cc[q][0][e], cc[q][1][e], cc[q][2][e], cc[q][3][e]};
#pragma unroll
            for (int p = 0; p < 4; p++)
#pragma unroll
              for (int s2 = 0; s2 < 4; s2++) {
                const d2 v = *reinterpret_cast<const ld2*>(vb + (16 * p + 4 * s2 + g) * QR_VM_LD + 2 * l15);
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                  for (int q = 0; q < NR; q++) acc[q][e] = mfma_na(w[p][q][s2], v[e], acc[q][e]);
              }
#pragma unroll
            for (int q = 0; q < NR; q++)
#pragma unroll
              for (int r = 0; r < 4; r++)
                __builtin_nontemporal_store(d2{acc[q][0][r], acc[q][1][r]}, reinterpret_cast<gd2*>(cp[q] + (long)(4 * r) * ld + 32 * s));
          }
#pragma unroll
          for (int q = 0; q < NR; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) cc[q][r] = cn[q][r];
        }
        stage_store(m + 1, h, vn);
      }
    }
  }
}

__device__ __attribute__((noinline)) void qr_trail4_coop(gdbl* Y, long ld, int rows32, int j0, int cstart, int ntl,
                                                         const ldbl* const (&Tq)[4], const ldbl* const (&Sq)[6], ldbl* Vs) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nst = (rows32 - j0) >> 5;                     // 32-row stages (j0 is a multiple of 64)
  // this wave's tiles: contiguous run (as wave_tiles)
  const int tbase = ntl / WG_WAVES, trem = ntl % WG_WAVES;
  const int tcnt = tbase + (wave < trem ? 1 : 0);
  const int tstart = wave * tbase + min(wave, trem);
  const int npass = (tbase + (trem ? 1 : 0) + 1) >> 1;
  for (int pass = 0; pass < npass; pass++) {
    const int t0 = 2 * pass;
    const int nt = __builtin_amdgcn_readfirstlane(max(0, min(2, tcnt - t0)));   // tiles of this wave in this pass
    const int cb0 = cstart + 16 * (tstart + t0);
    if (nt == 2) qr_trail4_coop_pass<2>(Y, ld, j0, nst, cb0, cb0 + 16, Tq, Sq, Vs);
    else if (nt == 1) qr_trail4_coop_pass<1>(Y, ld, j0, nst, cb0, cb0, Tq, Sq, Vs);
    else qr_trail4_coop_pass<0>(Y, ld, j0, nst, cstart, cstart, Tq, Sq, Vs);
  }
  __syncthreads();
}

// `big`: >= WG_WAVES*1024 doubles of LDS scratch when QR_QUAD (else WG_WAVES*512); may alias the gemm tile buffers
__device__ __attribute__((noinline)) void qr_r(gdbl* Y, long ld, int rows, int cols, ldbl* lds, ldbl* big, Prof* pr = nullptr,
                     unsigned long long* plast = nullptr, int ph_panel = 0, int ph_trail = 0, bool force_generic = false) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  ldbl* red = lds;                      // [2][WG_WAVES*16]
  ldbl* Ts = lds + 2 * WG_WAVES * 16;   // [16*16] T, column-major Ts[i + 16*j]
  ldbl* tau = Ts + 256;                 // [16]
  ldbl* bc = tau + 16;                  // broadcast scratch [64 + 32]
  const int kmax = min(rows, cols);
  const bool fine = pr && ph_panel < 0;       // self-test: finer phase split in slots 14..18
  const int rows16 = (rows + 15) & ~15;
  const int rows32 = (rows + 31) & ~31;      // ld >= rows32, rows [rows, rows32) zero

  ldbl* TsB = lds + QR_LDS_BASE;        // second T and the cross Gram S of a panel pair
  ldbl* Sm = TsB + 256;
  ldbl* tauB = Sm + 256;
  // quad aggregation (QR_QUAD): T of panels c, d, the six cross Grams, their taus
  ldbl* TsC = lds + QR_LDS_PAIR;
  ldbl* TsD = TsC + 256;
  ldbl* Sdc = TsD + 256;
  ldbl* Sca = Sdc + 256;
  ldbl* Scb = Sca + 256;
  ldbl* Sda = Scb + 256;
  ldbl* Sdb = Sda + 256;
  ldbl* tauC = Sdb + 256;
  ldbl* tauD = tauC + 16;

  // factor the full-width register panel at jp: V in place, T -> Tp (taus -> taup)
  auto factor_panel = [&](int jp, int nbp, ldbl* Tp, ldbl* taup) {
    for (int i = tid; i < 256; i += WG_THREADS) Tp[i] = 0.0;
    if (tid < 16) taup[tid] = 0.0;
    __syncthreads();
    qr_panel_regs(Y, ld, rows, jp, nbp, red, taup, bc);
    if (fine) prof_mark(pr, *plast, 14);
  };
  // panels (jp, jp+16), both full and register resident: T1, T2 and S21 = V2^T V1
  auto factor_pair = [&](int jp, ldbl* T1, ldbl* tau1, ldbl* T2, ldbl* tau2, ldbl* S21) {
    factor_panel(jp, 16, T1, tau1);
    qr_gram<false>(Y, ld, rows32, jp, jp, 16, jp, 16, jp, 16, big);
    if (fine) prof_mark(pr, *plast, 15);
    qr_T_from_gram(big, tau1, 16, T1);
    if (pr) prof_mark(pr, *plast, fine ? 16 : ph_panel);
    qr_tile_update_all(Y, ld, rows32, jp, 16, jp + 16, T1, big);
    if (pr) prof_mark(pr, *plast, fine ? 17 : ph_trail);
    factor_panel(jp + 16, 16, T2, tau2);
    qr_gram<true>(Y, ld, rows32, jp, jp + 16, 16, jp + 16, 16, jp, 16, big);   // V2^T V2 and S21 = V2^T V1 in one pass
    for (int i = tid; i < 256; i += WG_THREADS) S21[i] = big[256 + i];
    if (fine) prof_mark(pr, *plast, 15);
    qr_T_from_gram(big, tau2, 16, T2);
    if (pr) prof_mark(pr, *plast, fine ? 16 : ph_panel);
  };
  // second pair (c, d) of a quad whose first pair sits at ja: as factor_pair, with the cross Grams against the
  // first pair taken in the same pass over the rows as each panel's own Gram
  auto factor_pair2 = [&](int ja) {
    const int jc = ja + 32, jd = ja + 48;
    factor_panel(jc, 16, TsC, tauC);
    { const int jy[3] = {jc, ja, ja + 16}; qr_gramN<3>(Y, ld, rows32, jc, jc, jy, big); }      // Gcc, Sca, Scb
    for (int i = tid; i < 256; i += WG_THREADS) { Sca[i] = big[256 + i]; Scb[i] = big[512 + i]; }
    if (fine) prof_mark(pr, *plast, 15);
    qr_T_from_gram(big, tauC, 16, TsC);
    if (pr) prof_mark(pr, *plast, fine ? 16 : ph_panel);
    qr_tile_update_all(Y, ld, rows32, jc, 16, jd, TsC, big);
    if (pr) prof_mark(pr, *plast, fine ? 17 : ph_trail);
    factor_panel(jd, 16, TsD, tauD);
    { const int jy[4] = {jd, jc, ja, ja + 16}; qr_gramN<4>(Y, ld, rows32, jd, jd, jy, big); }  // Gdd, Sdc, Sda, Sdb
    for (int i = tid; i < 256; i += WG_THREADS) { Sdc[i] = big[256 + i]; Sda[i] = big[512 + i]; Sdb[i] = big[768 + i]; }
    if (fine) prof_mark(pr, *plast, 15);
    qr_T_from_gram(big, tauD, 16, TsD);
    if (pr) prof_mark(pr, *plast, fine ? 16 : ph_panel);
  };
  // tiles [cstart, cols) are shared out to the waves in contiguous runs
  auto wave_tiles = [&](int cstart, int& tstart, int& tcnt) {
    const int ntl = (cols - cstart + 15) / 16;   // may be 0
    const int tbase = ntl / WG_WAVES, trem = ntl % WG_WAVES;
    tcnt = tbase + (wave < trem ? 1 : 0);
    tstart = wave * tbase + min(wave, trem);
  };

  for (int j0 = 0; j0 < kmax;) {
    const int nb = min(QR_NB, kmax - j0);
    const bool fast = !force_generic && rows - j0 <= QR_RS * WG_THREADS;
    // four / two full register panels with a trailing matrix behind them are aggregated
    const bool quad = QR_QUAD && fast && j0 + 64 <= kmax && j0 + 64 < cols;
    const bool pair = !quad && fast && j0 + 32 <= kmax && j0 + 32 < cols;
    if (quad) {
      factor_pair(j0, Ts, tau, TsB, tauB, Sm);
      qr_tile_update2_all<2>(Y, ld, rows32, j0, j0 + 32, Ts, TsB, Sm, big);
      if (pr) prof_mark(pr, *plast, fine ? 17 : ph_trail);
      factor_pair2(j0);
      const ldbl* const Tq[4] = {Ts, TsB, TsC, TsD};
      const ldbl* const Sq[6] = {Sm, Sca, Scb, Sda, Sdb, Sdc};
      const int ntl4 = (cols - (j0 + 64) + 15) / 16;
      if (WG_WAVES > 1 && ntl4 <= QR_ROWPAR_TILES) {
        // few tiles left: one tile at a time, rows shared out to all waves
        for (int tl = 0; tl < ntl4; tl++) qr_tile_update4_all(Y, ld, rows32, j0, j0 + 64 + 16 * tl, Tq, Sq, big);
      } else if (WG_WAVES == 8 && !QR_TRAIL_PER_WAVE) {
        // reflector panels shared through LDS, all waves in step (see qr_trail4_coop).  A tile count of 4k + 1 would
        // leave three SIMDs idle for a whole tile: the odd tile is updated row-parallel instead.
        const int rem = ((ntl4 & 3) == 1 && ntl4 > 4) ? 1 : 0;
        qr_trail4_coop(Y, ld, rows32, j0, j0 + 64, ntl4 - rem, Tq, Sq, big);
        if (rem) qr_tile_update4_all(Y, ld, rows32, j0, j0 + 64 + 16 * (ntl4 - 1), Tq, Sq, big);
      } else {
        int tstart, tcnt;
        wave_tiles(j0 + 64, tstart, tcnt);
        for (int tg = 0; tg < tcnt; tg += 2) {
          const int cb0 = j0 + 64 + (tstart + tg) * 16;
          if (tcnt - tg >= 2) qr_trail4<2>(Y, ld, rows32, j0, cb0, Tq, Sq);
          else qr_trail4<1>(Y, ld, rows32, j0, cb0, Tq, Sq);
        }
      }
      __syncthreads();
      if (pr) prof_mark(pr, *plast, fine ? 18 : ph_trail);
      j0 += 64;
      continue;
    }
    if (pair) factor_pair(j0, Ts, tau, TsB, tauB, Sm);
    else {
      for (int i = tid; i < 256; i += WG_THREADS) Ts[i] = 0.0;
      if (tid < 16) tau[tid] = 0.0;
      __syncthreads();
      if (fast) {
        qr_panel_regs(Y, ld, rows, j0, nb, red, tau, bc);
        if (fine) prof_mark(pr, *plast, 14);
        qr_gram<false>(Y, ld, rows32, j0, j0, nb, j0, nb, j0, nb, big);
        if (fine) prof_mark(pr, *plast, 15);
        qr_T_from_gram(big, tau, nb, Ts);
      } else qr_panel_global(Y, ld, rows, j0, nb, red, Ts, tau, bc);
      if (pr) prof_mark(pr, *plast, fine ? 16 : ph_panel);
    }
    // ------------------------------------------------------------------ trailing update
    const int cstart = pair ? j0 + 32 : j0 + nb;
    int tstart, tcnt;
    wave_tiles(cstart, tstart, tcnt);
    for (int tg = 0; tg < tcnt; tg += QR_TC) {
      const int nt = min(QR_TC, tcnt - tg);
      const int cb0 = cstart + (tstart + tg) * 16;
      if (pair) {
        if (nt == 1) qr_trail2<1>(Y, ld, rows32, j0, 16, cb0, Ts, TsB, Sm);
        else if (nt == 2) qr_trail2<2>(Y, ld, rows32, j0, 16, cb0, Ts, TsB, Sm);
        else qr_trail2<3>(Y, ld, rows32, j0, 16, cb0, Ts, TsB, Sm);
      } else {
        if (nt == 1) qr_trail<1>(Y, ld, rows32, j0, nb, cb0, Ts);
        else if (nt == 2) qr_trail<2>(Y, ld, rows32, j0, nb, cb0, Ts);
        else qr_trail<3>(Y, ld, rows32, j0, nb, cb0, Ts);
      }
    }
    __syncthreads();
    if (pr) prof_mark(pr, *plast, fine ? 18 : ph_trail);
    j0 += pair ? 32 : 16;
  }
  (void)g; (void)l15; (void)lane;
}

// ---------------------------------------------------------------------------------------------
// One-sided Jacobi (Hestenes): A (m x n, ld lda) -> A V = U Sigma; V (n x n, ld ldv) accumulated from I
// (V == nullptr: not accumulated - the rotated columns sigma_j u_j then carry the LEFT singular vectors).
// Column norms of the final A are the singular values (unsorted).  Up to 8 lanes per column pair (fewer when a round has more pairs than the workgroup has 8-lane groups).
// Deflation: a column whose norm falls below 1e-14 ||A||_F is numerically null; it only carries rounding
// noise, rotating it never converges and never matters, and it can never grow back - after every sweep the
// null columns leave the tournament, so later sweeps run over the active columns only.
// `red`: >= 16 doubles; `act`: >= n ints of LDS.  Returns the number of sweeps, or -1 if not converged.
// lda / ldv should be odd (LDS bank spreading between the column pairs of different lane groups).
// ---------------------------------------------------------------------------------------------
// Jacobi rotation that orthogonalises two columns with squared norms al, be and inner product ga:
// t = tan(theta) = sgn(d) 2 ga / (|d| + sqrt(d^2 + 4 ga^2)), d = be - al  (smaller root); c = 1/sqrt(1 + t^2), s = c t:
// one sqrt, one reciprocal, one rsqrt - hardware seeds refined by Newton steps (the plain IEEE division / sqrt
// sequences dominate the cost of a rotation otherwise)
__device__ __forceinline__ void jac_cs(double al, double be, double ga, double& c, double& s) {
  const double d = be - al, g2 = 2.0 * ga;
  const double h2 = d * d + g2 * g2;
  double rs = __builtin_amdgcn_rsq(h2);
  rs = rs * (1.5 - 0.5 * h2 * rs * rs);
  double hy = h2 * rs;                                   // sqrt(h2)
  hy = hy + 0.5 * rs * (h2 - hy * hy);
  const double den = fabs(d) + hy;
  double rd = __builtin_amdgcn_rcp(den);
  rd = rd * (2.0 - den * rd);
  rd = rd * (2.0 - den * rd);
  const double t = copysign(g2, d * ga) * rd;
  const double o2 = 1.0 + t * t;
  c = __builtin_amdgcn_rsq(o2);
  c = c * (1.5 - 0.5 * o2 * c * c);
  c = c * (1.5 - 0.5 * o2 * c * c);
  s = c * t;
}

// One column pair handled by 8 lanes when the column length is exactly 8 R: no predicates, no address selects, the
// columns stay in registers between the dot products and the rotation.  Returns cos^2 of the angle if it rotated.
template <int R, class AP>
__device__ __forceinline__ double jac_pair8(AP ap, AP aq, int sub, double tol, double nul) {
  double x[R], y[R];
#pragma unroll
  for (int i = 0; i < R; i++) { x[i] = ap[sub + 8 * i]; y[i] = aq[sub + 8 * i]; }
  double al = 0, be = 0, ga = 0;
#pragma unroll
  for (int i = 0; i < R; i++) { al += x[i] * x[i]; be += y[i] * y[i]; ga += x[i] * y[i]; }
  al += dpp64<0x141>(al); be += dpp64<0x141>(be); ga += dpp64<0x141>(ga);
  al += dpp64<0x4E>(al); be += dpp64<0x4E>(be); ga += dpp64<0x4E>(ga);
  al += dpp64<0xB1>(al); be += dpp64<0xB1>(be); ga += dpp64<0xB1>(ga);
  if (!(ga * ga > (tol * tol) * (al * be) && al > nul && be > nul)) return 0.0;
  double c, s;
  jac_cs(al, be, ga, c, s);
#pragma unroll
  for (int i = 0; i < R; i++) { ap[sub + 8 * i] = c * x[i] - s * y[i]; aq[sub + 8 * i] = s * x[i] + c * y[i]; }
  return ga * ga / (al * be);
}

template <class T> struct jac_in_hbm { static constexpr bool value = false; };
template <> struct jac_in_hbm<gdbl*> { static constexpr bool value = true; };

// One column pair of a matrix in HBM handled by 8 lanes with 16-byte accesses: lane `sub` holds the row pairs
// 2 (sub + 8 j), +1, so the eight lanes of a pair read / write one whole 128-byte line per instruction (the columns are
// 128-byte aligned: lda a multiple of 16).  A round of the HBM-resident Jacobi is bound by the number of lines a CU's L1 can
// serve (a 160-row column is ten lines, every round touches all columns two to three times); with one row per lane and
// access every line was requested twice.  Same rotations as the scalar path; the dot products sum the rows in another order.
__device__ __forceinline__ double jac_pair_hbm16(gdbl* ap, gdbl* aq, int sub, int m, double tol, double nul) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) d2_t gd2_t;
  constexpr int JP = 6;                        // row pairs per lane held in registers at a time (12 rows, as the scalar path)
  const int npl = (m + 15) >> 4;               // row pairs per lane
  double al = 0, be = 0, ga = 0;
  d2_t x[JP], y[JP];
  auto loadc = [&](int j0) {
#pragma unroll
    for (int j = 0; j < JP; j++) {
      const int r = 2 * (sub + 8 * (j0 + j));
      const bool ok0 = r < m, ok1 = r + 1 < m;
      const int rc = ok1 ? r : 2 * sub;        // an in-range, aligned pair (m >= 16 on this path)
      const d2_t xv = *reinterpret_cast<const gd2_t*>(ap + rc), yv = *reinterpret_cast<const gd2_t*>(aq + rc);
      x[j] = d2_t{ok0 && ok1 ? xv[0] : 0.0, ok1 ? xv[1] : 0.0};
      y[j] = d2_t{ok0 && ok1 ? yv[0] : 0.0, ok1 ? yv[1] : 0.0};
    }
  };
  for (int j0 = 0; j0 < npl; j0 += JP) {
    loadc(j0);
#pragma unroll
    for (int j = 0; j < JP; j++) {
      al += x[j][0] * x[j][0]; be += y[j][0] * y[j][0]; ga += x[j][0] * y[j][0];
      al += x[j][1] * x[j][1]; be += y[j][1] * y[j][1]; ga += x[j][1] * y[j][1];
    }
  }
  al += dpp64<0x141>(al); be += dpp64<0x141>(be); ga += dpp64<0x141>(ga);
  al += dpp64<0x4E>(al); be += dpp64<0x4E>(be); ga += dpp64<0x4E>(ga);
  al += dpp64<0xB1>(al); be += dpp64<0xB1>(be); ga += dpp64<0xB1>(ga);
  if (!(ga * ga > (tol * tol) * (al * be) && al > nul && be > nul)) return 0.0;
  double c, s;
  jac_cs(al, be, ga, c, s);
  for (int j0 = 0; j0 < npl; j0 += JP) {
    if (npl > JP) loadc(j0);
#pragma unroll
    for (int j = 0; j < JP; j++) {
      const int r = 2 * (sub + 8 * (j0 + j));
      if (r + 1 < m) {
        *reinterpret_cast<gd2_t*>(ap + r) = d2_t{c * x[j][0] - s * y[j][0], c * x[j][1] - s * y[j][1]};
        *reinterpret_cast<gd2_t*>(aq + r) = d2_t{s * x[j][0] + c * y[j][0], s * x[j][1] + c * y[j][1]};
      }
    }
  }
  return ga * ga / (al * be);
}

template <class AP>     // AP = ldbl* (matrix in LDS) or gdbl* (in HBM): typed so that the inner loops are ds_* / global_*
__device__ __attribute__((noinline)) int jacobi_rsv(AP A, int lda, int m, int n, double* V, int ldv, ldbl* red,
                                                    __attribute__((address_space(3))) int* act, int maxsweeps) {
  const int tid = threadIdx.x;
  if (V) {
    for (int idx = tid; idx < n * n; idx += WG_THREADS) {
      int r = idx % n, c = idx / n;
      V[r + (long)ldv * c] = (r == c) ? 1.0 : 0.0;
    }
  }
  for (int c = tid; c < n; c += WG_THREADS) act[c] = c;
  __syncthreads();
  if (n < 2) return 0;
  double fro2 = 0.0;
  for (int idx = tid; idx < m * n; idx += WG_THREADS) { double v = A[(idx % m) + (long)lda * (idx / m)]; fro2 += v * v; }
  fro2 = wg_sum(fro2, red);
  const double nul = 1e-28 * fro2;
  const double tol = 1e-15;
  constexpr int JC = 12;                // rows per lane held in registers at a time
  int nact = n;
  int sweep = 0;
  bool conv = false;
  for (; sweep < maxsweeps; sweep++) {
    const int ne = (nact + 1) & ~1;       // even number of "players"
    const int npairs = ne / 2;
    // lanes per column pair (at most 8): enough that a lane's share of a column fits the JC registers (the
    // rotation then reuses the loaded values), and more if the workgroup still has a lane group for every pair
    int lg = 3;
    while (lg > 0 && (npairs << lg) > WG_THREADS) lg--;
    while (lg < 3 && ((m + (1 << lg) - 1) >> lg) > JC) lg++;
    const int LP = 1 << lg;
    const int sub = tid & (LP - 1);       // lane inside the pair group
    const int grp = tid >> lg;
    const int rpl = (m + LP - 1) >> lg;   // rows per lane
    const bool exact8 = !V && lg == 3 && m == 8 * rpl && (rpl == 2 || rpl == 4 || rpl == 6 || rpl == 8 || rpl == 10 || rpl == 12);
    const bool hbm16 = jac_in_hbm<AP>::value && !V && lg == 3 && (m & 1) == 0 && m >= 16 && (lda & 15) == 0;
    double worst = 0.0;                   // largest cos^2 of the angle between two columns met in this sweep
    for (int round = 0; round < ne - 1; round++) {
      for (int pb = 0; pb < npairs; pb += WG_THREADS >> lg) {
        const int pi = pb + grp;
        if (pi < npairs) {
          // round-robin tournament: player ne-1 fixed, others rotate
          int p, q;                       // round < ne-1 and pi < ne/2: one conditional subtraction replaces %
          if (pi == 0) { p = ne - 1; q = round; }
          else {
            p = round + pi; p -= (p >= ne - 1) ? (ne - 1) : 0;
            q = round + (ne - 1) - pi; q -= (q >= ne - 1) ? (ne - 1) : 0;
          }
          if (p > q) { int t_ = p; p = q; q = t_; }
          if (q < nact) {
            const int cp = act[p], cq = act[q];
            AP ap = A + (long)lda * cp;
            AP aq = A + (long)lda * cq;
            if (hbm16) {           // matrix in HBM, even column length, aligned columns: whole lines per access
              if constexpr (jac_in_hbm<AP>::value) worst = fmax(worst, jac_pair_hbm16(ap, aq, sub, m, tol, nul));
              continue;
            }
            if (exact8) {          // the common shapes: 8 lanes per pair, column length a multiple of 8 (wave-uniform)
              double w2 = 0.0;
              switch (rpl) {
                case 2: w2 = jac_pair8<2>(ap, aq, sub, tol, nul); break;
                case 4: w2 = jac_pair8<4>(ap, aq, sub, tol, nul); break;
                case 6: w2 = jac_pair8<6>(ap, aq, sub, tol, nul); break;
                case 8: w2 = jac_pair8<8>(ap, aq, sub, tol, nul); break;
                case 10: w2 = jac_pair8<10>(ap, aq, sub, tol, nul); break;
                default: w2 = jac_pair8<12>(ap, aq, sub, tol, nul); break;
              }
              worst = fmax(worst, w2);
              continue;
            }
            // rows r = sub + LP*i of the two columns, JC per lane at a time with all loads in flight together
            // (a single wave has no second wave to hide the LDS latency); when the whole column fits the JC
            // registers the rotation reuses them instead of reading the columns again
            double al = 0, be = 0, ga = 0;
            double x[JC], y[JC];
            auto loadc = [&](int i0) {
#pragma unroll
              for (int i = 0; i < JC; i++) {
                const int r = sub + ((i0 + i) << lg);
                const bool ok = r < m;
                const int rc = ok ? r : sub;
                const double xv = ap[rc], yv = aq[rc];
                x[i] = ok ? xv : 0.0; y[i] = ok ? yv : 0.0;
              }
            };
            for (int i0 = 0; i0 < rpl; i0 += JC) {
              loadc(i0);
#pragma unroll
              for (int i = 0; i < JC; i++) { al += x[i] * x[i]; be += y[i] * y[i]; ga += x[i] * y[i]; }
            }
            // sum over the LP lanes of the pair group (LP = 1, 2, 4, 8; groups are aligned): DPP moves, no LDS crossbar
            if (lg >= 3) { al += dpp64<0x141>(al); be += dpp64<0x141>(be); ga += dpp64<0x141>(ga); }
            if (lg >= 2) { al += dpp64<0x4E>(al); be += dpp64<0x4E>(be); ga += dpp64<0x4E>(ga); }
            if (lg >= 1) { al += dpp64<0xB1>(al); be += dpp64<0xB1>(be); ga += dpp64<0xB1>(ga); }
            if (ga * ga > (tol * tol) * (al * be) && al > nul && be > nul) {
              worst = fmax(worst, ga * ga / (al * be));
              double c, s;
              jac_cs(al, be, ga, c, s);
              for (int i0 = 0; i0 < rpl; i0 += JC) {
                if (rpl > JC) loadc(i0);
#pragma unroll
                for (int i = 0; i < JC; i++) {
                  const int r = sub + ((i0 + i) << lg);
                  if (r < m) { ap[r] = c * x[i] - s * y[i]; aq[r] = s * x[i] + c * y[i]; }
                }
              }
              if (V) {
                double* vp = V + (long)ldv * cp;
                double* vq = V + (long)ldv * cq;
                for (int r = sub; r < n; r += LP) { double x = vp[r], y = vq[r]; vp[r] = c * x - s * y; vq[r] = s * x + c * y; }
              }
            }
          }
        }
      }
      __syncthreads();
    }
    // converged if nobody rotated in this sweep - or if every rotation of this sweep was by less than 1e-8:
    // cyclic Jacobi converges quadratically, what is left is then below 1e-16 and the sweep that would only
    // verify it (a tenth of the whole cost) is skipped
    const double any = wg_max(worst, red);
    if (any < 1e-16) { conv = true; break; }
    // deflate: one thread per active column measures it, thread 0 compacts the list
    for (int i = tid; i < nact; i += WG_THREADS) {     // every thread only touches its own entries
      const int mycol = act[i];
      AP ac = A + (long)lda * mycol;
      double s2 = 0.0;
      for (int r = 0; r < m; r++) s2 += ac[r] * ac[r];
      act[i] = (s2 > nul) ? mycol : -1;
    }
    __syncthreads();
    if (tid == 0) {
      int w = 0;
      for (int i = 0; i < nact; i++) { const int c = act[i]; if (c >= 0) act[w++] = c; }
      red[15] = (double)w;
    }
    __syncthreads();
    nact = (int)red[15];
    __syncthreads();
    if (nact < 2) { conv = true; break; }
  }
  __syncthreads();
  return conv ? sweep + 1 : -1;
}

}  // namespace wg
