// Host side of the batched gauge sweep (v2_kernels.h): plans every time step of a lock-step batch on the host (all
// dimensions follow from the bond tables of the operands), uploads the per-problem descriptors and issues the launches.
#include "wg_common.h"
#define WG_THREADS 512
#define WG_WAVES 8
namespace v512 {
#include "wg_blocks.h"
}
#undef WG_THREADS
#undef WG_WAVES
#include "engine_types.h"
#include "v2_kernels.h"
#include "cq_kernels.h"
#include "ctx.h"
#include "v2_engine.h"
#include <mutex>

namespace {

inline int r16i(int x) { return (x + 15) & ~15; }
inline int r32i(int x) { return (x + 31) & ~31; }

struct QrDims { int rows, cols, kmax; };

// columns per block of the two-level Jacobi for factors of m rows, n columns: a block pair (2 nb columns, odd leading
// dimension) fits 150 KB of LDS, about eight blocks per factor (four workgroups per problem, seven launches per sweep), at
// most 32 columns (an inner sweep is 2 nb - 1 rounds with a barrier each); 0: does not fit (m > 1200)
inline int jac_block_nb(int m, int n) {
  static const int forced = [] { const char* e = getenv("MPBP_JACOBI_NB"); return e ? atoi(e) : 0; }();
  const int fit = (int)((150 * 1024 / 8 - 32) / (2 * (int64_t)(m | 1)));
  if (fit < 4) return 0;
  int nb = forced > 0 ? forced : std::max(8, (n + 7) / 8);
  nb = std::min(std::min(nb, 32), fit);
  return std::max(nb, 2);
}

// ------------------------------------------------------------------------------------------------------------------
// Look-ahead for the latency-bound case (one or two LARGE problems per launch: configs[3] hubs, configs[4]): the panel
// factorisation of block k+1 (17 cooperative column steps + Gram + T per 16 columns, ~25 launches per 64 columns, a
// handful of workgroups each) runs BESIDE the trailing update of block k instead of after it.  Two internal streams with
// disjoint CU masks (hipExtStreamCreateWithCUMask): the panel chain owns LA_RESERVED CUs - so the cooperative kernel's
// workgroups are co-resident by construction and its VALU-bound column steps do not share SIMDs with the trailing
// update's MFMA streams (which would slow them 3x, profiles/r03_dp_pipe_probe.txt) - the trailing update the rest.
//   stream B:  panel chain of block k   record(b)   wait(a: part 2 of block k-1)   part 1 of block k   ...block k+1
//   stream A:  wait(b)   part 2 of block k   record(a)
// part 1 = the next block's four panel tiles, updated in the fine-grained form of the in-block updates (quarter row
// chunks, ~90 us; the (8 tiles x row chunk) form takes ~340 us whatever the tile count: its time is one workgroup's pass
// over its 2048 rows), part 2 = all other tiles in the (8 tiles x row chunk) form.  Part 1 of block k needs part 2 of
// block k-1 (which brought those columns up to block k-1); block k+1's chain writes the second copy of the per-problem
// scratch (shift_auxlay) while part 2 of block k reads the first.
// ------------------------------------------------------------------------------------------------------------------
static const int LA_RESERVED = [] { const char* e = getenv("MPBP_LA_RESERVED"); const int v = e ? atoi(e) : 64; return (v >= 16 && v <= 128) ? v : 64; }();      // CUs of the panel stream
constexpr int LA_MAX_WGS = 24;       // look-ahead only while the batch is latency bound: row-chunk workgroups of all its problems (6400 x 1600 x 16 with 64 of them is throughput bound and loses 20 % to the reserved CUs)
struct LookAhead {
  hipStream_t sa = nullptr, sb = nullptr;
  hipEvent_t e_in = nullptr, e_a[2] = {nullptr, nullptr}, e_b = nullptr, e_out_a = nullptr, e_out_b = nullptr;
  bool ok = false, tried = false;
};
// Per-device state shared by every context (and self test) of the process on that device: the two CU-masked streams and
// six events of the look-ahead, and the "function attributes are set" flag.  Reference counted: mpbp_create /
// mpbp_selftest_qr_batched* acquire, mpbp_destroy / the end of the self test release, and the LAST release destroys the
// streams and events - so nothing of ours is alive when static destructors (and a profiler's finaliser) run at exit
// (round-3 review item 5).  The mutex makes creation safe for contexts of several host threads on one device; the streams
// themselves are used by one qr_batch at a time (include/mpbp_hip.h: contexts on one device share them - calls into
// different contexts of one device must not overlap in time).
struct DeviceShared { std::mutex mu; int refs = 0; bool attrs = false; LookAhead la; };
constexpr int MAX_DEV = 16;
static DeviceShared g_dev[MAX_DEV];

static void la_destroy(LookAhead& l) {
  if (l.sa) { (void)hipStreamSynchronize(l.sa); (void)hipStreamDestroy(l.sa); }
  if (l.sb) { (void)hipStreamSynchronize(l.sb); (void)hipStreamDestroy(l.sb); }
  for (hipEvent_t e : {l.e_in, l.e_a[0], l.e_a[1], l.e_b, l.e_out_a, l.e_out_b}) if (e) (void)hipEventDestroy(e);
  l = LookAhead{};
}
static LookAhead* lookahead_streams() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  DeviceShared& D = g_dev[dev];
  std::lock_guard<std::mutex> lk(D.mu);
  if (D.refs <= 0) return nullptr;             // nobody holds the device: there would be no one to release the streams
  LookAhead& l = D.la;
  if (!l.tried) {
    l.tried = true;
    if (getenv("MPBP_DEBUG_NO_LOOKAHEAD")) return nullptr;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return nullptr;
    const int ncu = pr.multiProcessorCount, words = (ncu + 31) / 32;
    if (ncu < 2 * LA_RESERVED) return nullptr;
    std::vector<uint32_t> mb(words, 0u), ma(words, 0u);
    for (int i = 0; i < ncu; i++) (i < LA_RESERVED ? mb : ma)[i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(&l.sa, words, ma.data()) != hipSuccess) { (void)hipGetLastError(); l.sa = nullptr; return nullptr; }
    if (hipExtStreamCreateWithCUMask(&l.sb, words, mb.data()) != hipSuccess) { (void)hipGetLastError(); l.sb = nullptr; la_destroy(l); l.tried = true; return nullptr; }
    bool ev = true;
    for (hipEvent_t* e : {&l.e_in, &l.e_a[0], &l.e_a[1], &l.e_b, &l.e_out_a, &l.e_out_b}) ev = ev && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
    l.ok = ev;
  }
  return l.ok ? &l : nullptr;
}
// dynamic-LDS limits of the batched QR's kernels: an attribute belongs to the (function, device) pair, so it is set once
// per device (it used to be set 6-7 times per qr_batch call, ~400 intercepted API calls per gauge sweep)
static void set_func_attrs_once() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) dev = -1;
  if (dev >= 0) { std::lock_guard<std::mutex> lk(g_dev[dev].mu); if (g_dev[dev].attrs) return; g_dev[dev].attrs = true; }
  hipFuncSetAttribute((const void*)v2::k_fpanel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v2::fpanel_lds_bytes(3));
  hipFuncSetAttribute((const void*)cq::k_cq_upd<256>, hipFuncAttributeMaxDynamicSharedMemorySize, cq::UPD_LDS_DOUBLES * 8);
  hipFuncSetAttribute((const void*)cq::k_cq_updfac, hipFuncAttributeMaxDynamicSharedMemorySize, cq::UPD_LDS_DOUBLES * 8);
  hipFuncSetAttribute((const void*)v2::k_jac_block, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
}

// Launch sequence of one R-only QR over a batch whose dimensions `dims` are known on the host.
// d_probs: device array of v2::QrProb (same order as dims).  force_tall: column-step panels even when they would fit.
// aux2: every problem's scratch holds TWO AuxLay copies (auxd doubles apart) - required for the look-ahead.
// (Tried and measured slower, round 3: cutting a many-problem batch into four groups that run this sequence on their own
// streams, so that one group's panel chain runs beside another's trailing update - 6400 x 1600 x 16: 49.3 against 28.3 ms,
// 7200 x 900 x 128: 82.5 against 73.6 ms; without disjoint CUs the chains' VALU-bound kernels share SIMDs with MFMA streams
// and run at a third of their speed, profiles/r03_dp_pipe_probe.txt.)
int qr_batch(hipStream_t st, const v2::QrProb* d_probs, const std::vector<QrDims>& dims, const v2::AuxLay& lay,
             bool force_tall, int* coop_err = nullptr, int coop_max_wgs = 128, int64_t aux2 = 0, int ncu = 256, int* path = nullptr) {
  const int P = (int)dims.size();
  if (path) *path = 0;             // which form ran (self tests): 1 communication-avoiding, 2 look-ahead, 0 launch per panel
  if (P == 0) return 0;
  int kmax_max = 0, kmax_min = 1 << 30, rows32_max = 0, cols_max = 0;
  for (const QrDims& d : dims) {
    kmax_max = std::max(kmax_max, d.kmax); kmax_min = std::min(kmax_min, d.kmax);
    rows32_max = std::max(rows32_max, r32i(d.rows)); cols_max = std::max(cols_max, d.cols);
  }
  const int nchunk = (rows32_max + v2::CH - 1) / v2::CH;
  if (nchunk > lay.nchunk) return -1;
  set_func_attrs_once();
  static const bool no_coop = getenv("MPBP_DEBUG_NO_COOP_PANEL") != nullptr;

  // ---- the panel chain of block jb (4 panels: in-block update, column steps, Gram, T) on stream s with scratch layout L
  auto panel_chain = [&](hipStream_t s, const v2::AuxLay& L, int jb, int npmax, bool tall, int coop_wgs) {
    for (int p = 0; p < npmax; p++) {
      const int jp = jb + 16 * p;
      if (p > 0) {
        if (!tall) {
          switch (p) {
            case 1: hipLaunchKernelGGL(v2::k_inblock<1>, dim3(P), dim3(512), 0, s, d_probs, L, jb); break;
            case 2: hipLaunchKernelGGL(v2::k_inblock<2>, dim3(P), dim3(512), 0, s, d_probs, L, jb); break;
            default: hipLaunchKernelGGL(v2::k_inblock<3>, dim3(P), dim3(512), 0, s, d_probs, L, jb); break;
          }
        } else {
          // few problems: a quarter chunk per workgroup (tw = 0, grid.x = 4); many: a chunk per workgroup (tw = 1)
          const int twi = ((int64_t)nchunk * P <= 128) ? 0 : 1;
          const dim3 g(twi == 0 ? 4 : 1, nchunk, P);
          switch (p) {
            case 1: hipLaunchKernelGGL(v2::k_trailW<1>, g, dim3(256), 0, s, d_probs, L, jb, 1, twi, 0, 0);
                    hipLaunchKernelGGL(v2::k_trailU<1>, g, dim3(256), 0, s, d_probs, L, jb, 1, twi, 0, 0); break;
            case 2: hipLaunchKernelGGL(v2::k_trailW<2>, g, dim3(256), 0, s, d_probs, L, jb, 1, twi, 0, 0);
                    hipLaunchKernelGGL(v2::k_trailU<2>, g, dim3(256), 0, s, d_probs, L, jb, 1, twi, 0, 0); break;
            default: hipLaunchKernelGGL(v2::k_trailW<3>, g, dim3(256), 0, s, d_probs, L, jb, 1, twi, 0, 0);
                     hipLaunchKernelGGL(v2::k_trailU<3>, g, dim3(256), 0, s, d_probs, L, jb, 1, twi, 0, 0); break;
          }
        }
      }
      if (tall) {
        // every row-chunk workgroup resident at once: one launch with arrival counters; else one launch per column
        // (the cooperative kernel needs all the row-chunk workgroups of a problem resident together: a batch too large
        // for that goes through it in groups of problems, as long as that takes fewer launches than the 17 column steps)
        const int pb = (nchunk > 0) ? coop_wgs / nchunk : 0;          // problems per cooperative launch
        if (coop_err && !no_coop && pb >= 1 && (P + pb - 1) / pb <= 8) {
          for (int p0 = 0; p0 < P; p0 += pb)
            hipLaunchKernelGGL(v2::k_colsteps_coop, dim3(nchunk, std::min(pb, P - p0)), dim3(512), 0, s, d_probs + p0, L, jp, p, coop_err);
        } else
          for (int jj = 0; jj <= 16; jj++)
            hipLaunchKernelGGL(v2::k_colstep, dim3(nchunk, P), dim3(512), 0, s, d_probs, L, jp, jj, p);
        hipLaunchKernelGGL(v2::k_gram, dim3(nchunk * v2::GSUB, P), dim3(512), 0, s, d_probs, L, jb, p);
        hipLaunchKernelGGL(v2::k_build_T, dim3(P), dim3(256), 0, s, d_probs, L, jb, p);
      } else {
        hipLaunchKernelGGL(v2::k_fpanel, dim3(P), dim3(512), v2::fpanel_lds_bytes(p), s, d_probs, L, jb, p);
      }
    }
  };
  static const int coop_nt = [] { const char* e = getenv("MPBP_COOP_NT"); return e ? atoi(e) : 0; }();
  // tiles [t0, t1) of the 4-panel trailing update in the (8 tiles x row chunk) form
  auto trail_coop = [&](hipStream_t s, const v2::AuxLay& L, int jb, int t0, int t1) {
    if (t1 <= t0) return;
    const bool nt1 = coop_nt != 2;   // one tile per wave measured 8-20 % faster at every size tried (two workgroups per CU)
    if (nt1) {
      const dim3 gc((t1 - t0 + 7) / 8, nchunk, P);
      hipLaunchKernelGGL(v2::k_trailW_coop<1>, gc, dim3(512), 0, s, d_probs, L, jb, t0, t1);
      hipLaunchKernelGGL(v2::k_trailU_coop<1>, gc, dim3(512), 0, s, d_probs, L, jb, t0, t1);
    } else {
      const dim3 gc((t1 - t0 + 15) / 16, nchunk, P);
      hipLaunchKernelGGL(v2::k_trailW_coop<2>, gc, dim3(512), 0, s, d_probs, L, jb, t0, t1);
      hipLaunchKernelGGL(v2::k_trailU_coop<2>, gc, dim3(512), 0, s, d_probs, L, jb, t0, t1);
    }
  };

  // ---- look-ahead: uniform large problems only (every block has four full panels for every problem, the cooperative
  //      panel kernel fits the reserved CUs, the trailing matrix is wide enough to have a part 2)
  const v2::AuxLay lay2[2] = {lay, v2::second_auxlay(lay, aux2)};

  // ---- communication-avoiding form (cq_kernels.h): tall problems, every one with rows >= cols; the node slots live in
  //      the per-problem scratch behind the fixed headers of BOTH copies (which sit together at the front: lay.part on =
  //      the partial products / Grams / W0 of the other form, first and second copy back to back), so a CAQR call never
  //      touches the arrival counters a later look-ahead call on the same scratch relies on (round-3 advisor)
  {
    static const bool no_cq = getenv("MPBP_DEBUG_NO_CAQR") != nullptr;
    static const int cq_min_rows = [] { const char* e = getenv("MPBP_CQ_MIN_ROWS"); return e ? atoi(e) : v2::CH; }();
    bool tall_all = true;
    for (const QrDims& d : dims) tall_all = tall_all && d.rows >= d.cols;
    int64_t slots = 0;
    for (int n = (rows32_max + 255) / 256;; n = (n + 3) / 4) { slots += n; if (n == 1) break; }
    const int64_t ws_off = lay.part;
    if (!no_cq && !force_tall && aux2 > 0 && tall_all && rows32_max > cq_min_rows && ws_off + slots * cq::IMG_DOUBLES <= 2 * aux2) {
      // One stream, one launch after the other.  Measured and dropped (round 3): (a) the upper-level factorisations of a
      // single problem on a CU-masked stream beside the updates - the 20-40 us per event hand-over and the CUs taken from
      // the updates cost what the overlap gained (16384 x 4096: 43.9 against 44.4 ms at the time); (b) the batch cut into
      // 2 / 4 groups of problems on their own streams, so that one group's narrow launches fill CUs beside another's wide
      // ones: 6400 x 1600 x 16 26.9 -> 28.0 / 34.5 ms, 7200 x 900 x 128 69.9 -> 67.1 / 71.6 ms.
      // Per block: F_0, then one launch per level with the update U_l and the next level's factorisation F_{l+1} (k_cq_updfac).
      static const bool no_fuse = getenv("MPBP_DEBUG_CQ_NOFUSE") != nullptr;
      const int cols16_max = r16i(cols_max);
      // tiles per workgroup: one tile per wave and four-wave workgroups for the small upper levels; else the number of tile
      // groups with the fewest (rounds over the CUs) x (time of a workgroup: a fixed part + 6.2 us per tile, measured with the
      // chip full) - it decides how the last round is filled.  The fixed part: ~15 us of image load + the first tile's wait;
      // swept 4 ... 50 us on four shapes (round 4, with the rewritten tile update): flat within 1 % from 8 to 50 on the
      // many-problem shapes, 6400 x 1600 x 16 22.0 -> 21.4 ms and 16384 x 4096 26.9 -> 26.7 ms at 30
      static const double img_us = [] { const char* e = getenv("MPBP_CQ_IMG_US"); return e ? atof(e) : 30.0; }();
      auto tile_groups = [&](int ntl, int n, int& tpg, int& nthr) {
        const int64_t tiles = (int64_t)ntl * n * P;
        // four-wave workgroups always: the fused update + factor launch needs that shape, and the two-waves-per-SIMD build of the
        // update alone (cq_kernels.h, compute_tile) wins only at >= 128 problems (profiles/r04_cq_upd_probe.txt)
        tpg = 8; nthr = 256;
        if (tiles <= 4 * ncu) { tpg = 4; return; }
        double best = 1e30;
        for (int g = 1; g <= (ntl + 7) / 8; g++) {
          const int t = (ntl + g - 1) / g;
          if (t > 64) continue;
          const int64_t wgs = (int64_t)((ntl + t - 1) / t) * n * P;
          const double cost = (double)((wgs + ncu - 1) / ncu) * (img_us + 6.2 * t);
          if (cost < best) { best = cost; tpg = t; }
        }
      };
      for (int jb = 0; jb < kmax_max; jb += 64) {
        const int ntl = cols16_max > jb + 64 ? (cols16_max - jb - 64) / 16 : 0;
        int nl[12], nlev = 0;
        for (int n = (rows32_max - jb + 255) / 256; nlev < 12; n = (n + 3) / 4) { nl[nlev++] = n; if (n == 1) break; }
        int slot = 0;
        // (Round 3 had a second build of this kernel at two waves per SIMD for launches with more nodes than CUs.  Since the
        //  column steps are straight-line code the one-per-CU build is as fast per CU - 7200 x 900 x 128: 63.5 against 63.8 ms,
        //  21600 x 900 x 16: 28.1 / 28.2, 6400 x 1600 x 16: 22.5 / 22.4 - and the other one carried 1072 spills: removed.)
        hipLaunchKernelGGL(cq::k_cq_fac2, dim3(nl[0], P), dim3(256), cq::FAC_LDS_DOUBLES * 8, st, d_probs, ws_off, jb, 0, 0, 0);
        for (int level = 0; level < nlev; level++) {
          const int n = nl[level];
          const bool more = level + 1 < nlev;
          int tpg = 8, nthr = 256;
          if (ntl > 0) tile_groups(ntl, n, tpg, nthr);
          const int ntg = ntl > 0 ? (ntl + tpg - 1) / tpg : 0;
          if (ntl > 0 && more && !no_fuse) {
            const int64_t wgs = (int64_t)P * nl[level + 1] + (int64_t)P * n * ntg;
            hipLaunchKernelGGL(cq::k_cq_updfac, dim3((unsigned)wgs), dim3(256), cq::UPD_LDS_DOUBLES * 8, st, d_probs, P, ws_off, jb, level, slot, n, ntg,
                               tpg, slot + n, nl[level + 1]);
          } else {
            if (ntl > 0) {
              hipLaunchKernelGGL(cq::k_cq_upd<256>, dim3(ntg, n, P), dim3(256), cq::UPD_LDS_DOUBLES * 8, st, d_probs, ws_off, jb, level, slot, tpg, 0);
            }
            if (more) {
              hipLaunchKernelGGL(cq::k_cq_fac2, dim3(nl[level + 1], P), dim3(256), cq::FAC_LDS_DOUBLES * 8, st, d_probs, ws_off, jb, level + 1, slot + n, 0);
            }
          }
          slot += n;
        }
      }
      if (path) *path = 1;
      return hipGetLastError() == hipSuccess ? 0 : -2;
    }
  }
  // (the streams are created on first use - only when a batch has the look-ahead's shape, after the tree form declined)
  const bool la_cand = aux2 > 0 && coop_err && !no_coop && !getenv("MPBP_DEBUG_NO_COOP_TRAIL") &&
                       (int64_t)nchunk * P <= LA_MAX_WGS && rows32_max > 2 * v2::CH && kmax_min == kmax_max;
  LookAhead* la = la_cand ? lookahead_streams() : nullptr;
  const bool la_shape = la != nullptr;
  bool la_on = false;
  int la_par = 0;
  auto la_leave = [&]() {
    if (!la_on) return;
    hipEventRecord(la->e_out_a, la->sa); hipEventRecord(la->e_out_b, la->sb);
    hipStreamWaitEvent(st, la->e_out_a, 0); hipStreamWaitEvent(st, la->e_out_b, 0);
    la_on = false;
  };

  for (int jb = 0; jb < kmax_max; jb += 64) {
    const int npmax = std::min(4, (kmax_max - jb + 15) / 16);
    // register panels / one workgroup per problem for the in-block updates while the rows below the diagonal fit
    const bool tall = force_tall || rows32_max - jb > v2::CH;
    const int ntile4 = cols_max > jb + 64 ? (cols_max - jb - 64 + 15) / 16 : 0;
    if (la_shape && tall && npmax == 4 && kmax_min - jb >= 64 && ntile4 > 8 && rows32_max - jb > 2 * v2::CH) {
      const bool first = !la_on;
      if (path) *path = 2;
      if (first) {
        hipEventRecord(la->e_in, st);
        hipStreamWaitEvent(la->sa, la->e_in, 0); hipStreamWaitEvent(la->sb, la->e_in, 0);
        la_on = true; la_par = 0;
      }
      const v2::AuxLay& L = lay2[la_par];
      panel_chain(la->sb, L, jb, 4, true, LA_RESERVED);
      hipEventRecord(la->e_b, la->sb);
      // part 2 beside the next block's chain
      hipStreamWaitEvent(la->sa, la->e_b, 0);
      trail_coop(la->sa, L, jb, 4, ntile4);
      hipEventRecord(la->e_a[la_par], la->sa);
      // part 1 (the next block's panel tiles) behind part 2 of the previous block.  (Tiles 1..3 on a second stream of the
      // same CUs beside the next panel's column steps: measured no faster, 41.9 against 41.0 ms.)
      if (!first) hipStreamWaitEvent(la->sb, la->e_a[la_par ^ 1], 0);
      hipLaunchKernelGGL(v2::k_trailW<4>, dim3(16, nchunk, P), dim3(256), 0, la->sb, d_probs, L, jb, 0, 0, 0, 0);
      hipLaunchKernelGGL(v2::k_trailU<4>, dim3(16, nchunk, P), dim3(256), 0, la->sb, d_probs, L, jb, 0, 0, 0, 0);
      la_par ^= 1;
      continue;
    }
    la_leave();
    panel_chain(st, lay, jb, npmax, tall, coop_max_wgs);
    const int c0min = jb + 16;     // a problem with one panel left starts its trailing tiles here
    const int ntile_max = cols_max > c0min ? (cols_max - c0min + 15) / 16 : 0;
    if (ntile_max > 0) {
      // Many problems: one wave per tile pair over all rows (fused, the tuned wg::qr_trail4); few: tiles x row chunks
      // over the grid in two launches.  Problems with fewer than four panels left always take the second form.
      const bool fused = npmax == 4 && ntile4 > 0 && (int64_t)P * ((ntile4 + 1) / 2) >= 1024 && !getenv("MPBP_DEBUG_NO_FUSED_TRAIL");
      static const int trail_nt = [] { const char* e = getenv("MPBP_TRAIL_NT"); return e ? atoi(e) : 2; }();
      if (fused) {
        if (trail_nt == 1) hipLaunchKernelGGL(v2::k_trail4f<1>, dim3((ntile4 + 3) / 4, P), dim3(256), 0, st, d_probs, lay, jb);
        else hipLaunchKernelGGL(v2::k_trail4f<2>, dim3((ntile4 + 7) / 8, P), dim3(256), 0, st, d_probs, lay, jb);
      }
      // few large problems: (16 tiles x row chunk) workgroups with the panels shared through LDS, two launches
      const bool coop = !fused && npmax == 4 && ntile4 > 0 && !getenv("MPBP_DEBUG_NO_COOP_TRAIL");
      if (coop) trail_coop(st, lay, jb, 0, ntile4);
      const int only_short = (fused || coop) ? 1 : 0;
      if ((!fused && !coop) || kmax_min - jb < 64) {
        const dim3 g((ntile_max + 3) / 4, nchunk, P);
        switch (npmax) {
          case 1: hipLaunchKernelGGL(v2::k_trailW<1>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0);
                  hipLaunchKernelGGL(v2::k_trailU<1>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0); break;
          case 2: hipLaunchKernelGGL(v2::k_trailW<2>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0);
                  hipLaunchKernelGGL(v2::k_trailU<2>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0); break;
          case 3: hipLaunchKernelGGL(v2::k_trailW<3>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0);
                  hipLaunchKernelGGL(v2::k_trailU<3>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0); break;
          default: hipLaunchKernelGGL(v2::k_trailW<4>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0);
                   hipLaunchKernelGGL(v2::k_trailU<4>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short, 0); break;
        }
      }
    }
  }
  la_leave();
  return hipGetLastError() == hipSuccess ? 0 : -2;
}


}  // namespace

void v2_device_acquire(int dev) {
  if (dev < 0 || dev >= MAX_DEV) return;
  std::lock_guard<std::mutex> lk(g_dev[dev].mu);
  g_dev[dev].refs++;
}
void v2_device_release(int dev) {
  if (dev < 0 || dev >= MAX_DEV) return;
  DeviceShared& D = g_dev[dev];
  std::lock_guard<std::mutex> lk(D.mu);
  if (D.refs > 0 && --D.refs == 0) { int cur = 0; (void)hipGetDevice(&cur); (void)hipSetDevice(dev); la_destroy(D.la); (void)hipSetDevice(cur); }
}
namespace { struct DeviceHold { int dev; explicit DeviceHold(int d) : dev(d) { v2_device_acquire(d); } ~DeviceHold() { v2_device_release(dev); } }; }

// ================================================================================================
// self test: nprob independent rows x cols matrices through the batched QR; R[p] = [kmax x cols] (ld kmax)
// ================================================================================================
static int st2_fail(const char* what, hipError_t e) { g_create_error = std::string(what) + ": " + hipGetErrorString(e); return MPBP_EHIP; }
#define ST2CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return st2_fail(#call, e_); } while (0)

extern "C" int mpbp_selftest_qr_batched(int32_t device, int32_t rows, int32_t cols, int32_t nprob, int32_t force_tall,
                                        const double* A, double* R, double* ms_out) {
  ST2CHK(hipSetDevice(device));
  if (rows < 1 || cols < 1 || nprob < 1) { g_create_error = "bad shape"; return MPBP_EINVAL; }
  DeviceHold hold(device);
  const int ld = r32i(rows), c16 = r16i(cols) + 16, kmax = std::min(rows, cols);
  const size_t per = (size_t)ld * c16;
  const int nchunk = (ld + v2::CH - 1) / v2::CH, ntile = c16 / 16;
  const v2::AuxLay lay = v2::make_auxlay(nchunk, ntile);
  const size_t auxd = (size_t)v2::auxlay_doubles(nchunk, ntile);
  double *dY = nullptr, *dAux = nullptr; v2::QrProb* dP = nullptr;
  ST2CHK(hipMalloc(&dY, sizeof(double) * per * nprob));
  ST2CHK(hipMalloc(&dAux, sizeof(double) * 2 * auxd * nprob));          // two scratch copies per problem: look-ahead of qr_batch
  ST2CHK(hipMalloc(&dP, sizeof(v2::QrProb) * nprob));
  ST2CHK(hipMemset(dAux, 0, sizeof(double) * 2 * auxd * nprob));
  std::vector<double> Y(per, 0.0);
  std::vector<v2::QrProb> hp(nprob);
  std::vector<QrDims> dims(nprob);
  for (int p = 0; p < nprob; p++) {
    std::fill(Y.begin(), Y.end(), 0.0);
    const double* Ap = A + (size_t)p * rows * cols;
    for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) Y[i + (size_t)ld * j] = Ap[i + (size_t)rows * j];
    ST2CHK(hipMemcpy(dY + per * p, Y.data(), sizeof(double) * per, hipMemcpyHostToDevice));
    hp[p] = v2::QrProb{dY + per * p, dAux + 2 * auxd * p, ld, rows, cols, kmax};
    dims[p] = QrDims{rows, cols, kmax};
  }
  ST2CHK(hipMemcpy(dP, hp.data(), sizeof(v2::QrProb) * nprob, hipMemcpyHostToDevice));
  (void)lookahead_streams();          // a context creates them once in its life: keep that out of the timed region
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  int* dErr = nullptr;
  ST2CHK(hipMalloc(&dErr, sizeof(int)));
  ST2CHK(hipMemset(dErr, 0, sizeof(int)));
  int ncu = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) ncu = pr.multiProcessorCount; }
  const int rc = qr_batch(0, dP, dims, lay, force_tall != 0, dErr, 128, (int64_t)auxd, ncu);
  hipEventRecord(e1, 0);
  ST2CHK(hipDeviceSynchronize());
  if (rc != 0) { g_create_error = "qr_batch launch failed"; return MPBP_EHIP; }
  { int herr = 0; ST2CHK(hipMemcpy(&herr, dErr, sizeof(int), hipMemcpyDeviceToHost)); hipFree(dErr);
    if (herr) { g_create_error = "cooperative panel: an arrival counter timed out"; return MPBP_EHIP; } }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  if (ms_out) *ms_out = ms;
  hipEventDestroy(e0); hipEventDestroy(e1);
  for (int p = 0; p < nprob; p++) {
    ST2CHK(hipMemcpy(Y.data(), dY + per * p, sizeof(double) * per, hipMemcpyDeviceToHost));
    double* Rp = R + (size_t)p * kmax * cols;
    for (int j = 0; j < cols; j++) for (int i = 0; i < kmax; i++) Rp[i + (size_t)kmax * j] = (j >= i) ? Y[i + (size_t)ld * j] : 0.0;
  }
  hipFree(dY); hipFree(dAux); hipFree(dP);
  return MPBP_OK;
}

// self test: ONE matrix per call, the calls of a sequence sharing one scratch sized for the largest (as the time steps of a
// gauge sweep do): rows[s] x cols, A / R concatenated; path[s] = the form qr_batch took (1 communication-avoiding, 2
// look-ahead, 0 launch per panel).  Pins that a communication-avoiding call leaves the arrival counters of a later
// look-ahead call intact (round-3 advisor: its node slots used to run over the second scratch copy's header).
extern "C" int mpbp_selftest_qr_batched_seq(int32_t device, int32_t nshape, const int32_t* rows, int32_t cols, const double* A, double* R, int32_t* path) {
  ST2CHK(hipSetDevice(device));
  if (nshape < 1 || cols < 1 || !rows) { g_create_error = "bad shape"; return MPBP_EINVAL; }
  DeviceHold hold(device);
  int rmax = 0;
  for (int s = 0; s < nshape; s++) { if (rows[s] < 1) { g_create_error = "bad shape"; return MPBP_EINVAL; } rmax = std::max(rmax, rows[s]); }
  const int ldmax = r32i(rmax), c16 = r16i(cols) + 16;
  const int nchunk = (ldmax + v2::CH - 1) / v2::CH, ntile = c16 / 16;
  const v2::AuxLay lay = v2::make_auxlay(nchunk, ntile);
  const size_t auxd = (size_t)v2::auxlay_doubles(nchunk, ntile);
  double *dY = nullptr, *dAux = nullptr; v2::QrProb* dP = nullptr; int* dErr = nullptr;
  ST2CHK(hipMalloc(&dY, sizeof(double) * (size_t)ldmax * c16));
  ST2CHK(hipMalloc(&dAux, sizeof(double) * 2 * auxd));
  ST2CHK(hipMalloc(&dP, sizeof(v2::QrProb)));
  ST2CHK(hipMalloc(&dErr, sizeof(int)));
  ST2CHK(hipMemset(dAux, 0, sizeof(double) * 2 * auxd));                 // once, as v2_gauge_sweep does
  ST2CHK(hipMemset(dErr, 0, sizeof(int)));
  int ncu = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) ncu = pr.multiProcessorCount; }
  std::vector<double> Y((size_t)ldmax * c16);
  size_t aoff = 0, roff = 0;
  int rc_all = MPBP_OK;
  for (int s = 0; s < nshape && rc_all == MPBP_OK; s++) {
    const int m = rows[s], ld = r32i(m), kmax = std::min(m, cols);
    std::fill(Y.begin(), Y.end(), 0.0);
    for (int j = 0; j < cols; j++) for (int i = 0; i < m; i++) Y[i + (size_t)ld * j] = A[aoff + i + (size_t)m * j];
    ST2CHK(hipMemcpy(dY, Y.data(), sizeof(double) * (size_t)ld * c16, hipMemcpyHostToDevice));
    const v2::QrProb hp{dY, dAux, ld, m, cols, kmax};
    ST2CHK(hipMemcpy(dP, &hp, sizeof hp, hipMemcpyHostToDevice));
    std::vector<QrDims> dims{QrDims{m, cols, kmax}};
    int pth = 0;
    const int rc = qr_batch(0, dP, dims, lay, false, dErr, ncu * 3 / 4, (int64_t)auxd, ncu, &pth);
    ST2CHK(hipDeviceSynchronize());
    if (path) path[s] = pth;
    int herr = 0; ST2CHK(hipMemcpy(&herr, dErr, sizeof(int), hipMemcpyDeviceToHost));
    if (rc != 0) { g_create_error = "qr_batch launch failed"; rc_all = MPBP_EHIP; }
    else if (herr) { g_create_error = "cooperative panel: an arrival counter timed out"; rc_all = MPBP_EHIP; }
    ST2CHK(hipMemcpy(Y.data(), dY, sizeof(double) * (size_t)ld * c16, hipMemcpyDeviceToHost));
    for (int j = 0; j < cols; j++) for (int i = 0; i < kmax; i++) R[roff + i + (size_t)kmax * j] = (j >= i) ? Y[i + (size_t)ld * j] : 0.0;
    aoff += (size_t)m * cols; roff += (size_t)kmax * cols;
  }
  hipFree(dY); hipFree(dAux); hipFree(dP); hipFree(dErr);
  return rc_all;
}

// self test of the multi-launch Jacobi (k_jac_round / k_jac_check): A [m x n] (ld m|1 inside) -> column norms after
// convergence (= singular values, unsorted) and the number of sweeps (-1: not converged within maxsweeps)
static int jacobi_selftest(int32_t device, int32_t m, int32_t n, const double* A, double* sigma, int32_t maxsweeps, int32_t* sweeps, bool block) {
  ST2CHK(hipSetDevice(device));
  if (m < 1 || n < 1 || n > m || m > 1024) { g_create_error = "need 1 <= n <= m <= 1024"; return MPBP_EINVAL; }
  const int ldJ = v2::jac_ld(m);
  std::vector<double> JA((size_t)ldJ * n, 0.0);
  double fro2 = 0.0;
  for (int c = 0; c < n; c++) for (int r = 0; r < m; r++) { const double v = A[r + (size_t)m * c]; JA[r + (size_t)ldJ * c] = v; fro2 += v * v; }
  double *dJA = nullptr, *dscal = nullptr; v2::SvdDesc* dd = nullptr;
  ST2CHK(hipMalloc(&dJA, sizeof(double) * JA.size())); ST2CHK(hipMalloc(&dscal, 512)); ST2CHK(hipMalloc(&dd, sizeof(v2::SvdDesc)));
  ST2CHK(hipMemcpy(dJA, JA.data(), sizeof(double) * JA.size(), hipMemcpyHostToDevice));
  double hs[8] = {0, 0, 0, fro2, 0, 0, (n < 2) ? 1.0 : 0.0, 0};
  ST2CHK(hipMemcpy(dscal, hs, sizeof hs, hipMemcpyHostToDevice));
  hs[7] = (double)n;
  ST2CHK(hipMemcpy(dscal, hs, sizeof hs, hipMemcpyHostToDevice));
  int32_t* dact = nullptr;
  ST2CHK(hipMalloc(&dact, sizeof(int32_t) * n));
  { std::vector<int32_t> ha(n); for (int i = 0; i < n; i++) ha[i] = i; ST2CHK(hipMemcpy(dact, ha.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice)); }
  v2::SvdDesc D{};
  D.JA = dJA; D.Rr = m; D.r1 = n; D.scal = dscal; D.act = dact;
  ST2CHK(hipMemcpy(dd, &D, sizeof D, hipMemcpyHostToDevice));
  if (block) {
    set_func_attrs_once();
    const int nb = jac_block_nb(m, n);
    if (nb < 2) { g_create_error = "factor too tall for an LDS-resident block pair"; return MPBP_EUNSUPPORTED; }
    const int nblk = (n + nb - 1) / nb, ne = (nblk + 1) & ~1;
    const size_t jlds = sizeof(double) * ((size_t)(m | 1) * 2 * nb + 32);
    for (int sw = 0; sw < maxsweeps; sw++) {
      for (int r = 0; r < std::max(1, ne - 1); r++) hipLaunchKernelGGL(v2::k_jac_block, dim3(ne / 2, 1), dim3(512), jlds, 0, (const v2::SvdDesc*)dd, r, nb);
      hipLaunchKernelGGL(v2::k_jac_deflate, dim3(1), dim3(512), 0, 0, (const v2::SvdDesc*)dd);
    }
  } else {
    const int ne = (n + 1) & ~1;
    for (int sw = 0; sw < maxsweeps; sw++) {
      for (int r = 0; r < ne - 1; r++) hipLaunchKernelGGL(v2::k_jac_round, dim3((ne / 2 + 15) / 16, 1), dim3(512), 0, 0, (const v2::SvdDesc*)dd, r);
      hipLaunchKernelGGL(v2::k_jac_check, dim3(1), dim3(64), 0, 0, (const v2::SvdDesc*)dd, 1);
    }
  }
  ST2CHK(hipDeviceSynchronize());
  ST2CHK(hipGetLastError());
  ST2CHK(hipMemcpy(JA.data(), dJA, sizeof(double) * JA.size(), hipMemcpyDeviceToHost));
  ST2CHK(hipMemcpy(hs, dscal, sizeof hs, hipMemcpyDeviceToHost));
  for (int c = 0; c < n; c++) { double s = 0; for (int r = 0; r < m; r++) s += JA[r + (size_t)ldJ * c] * JA[r + (size_t)ldJ * c]; sigma[c] = sqrt(s); }
  *sweeps = hs[6] != 0.0 ? (int)hs[5] : -1;
  hipFree(dJA); hipFree(dscal); hipFree(dd); hipFree(dact);
  return MPBP_OK;
}
extern "C" int mpbp_selftest_jacobi_grid(int32_t device, int32_t m, int32_t n, const double* A, double* sigma, int32_t maxsweeps, int32_t* sweeps) {
  return jacobi_selftest(device, m, n, A, sigma, maxsweeps, sweeps, false);
}
// the two-level (block) Jacobi of the truncating sweep (v2::k_jac_block) on one m x n matrix, same outputs
extern "C" int mpbp_selftest_jacobi_block(int32_t device, int32_t m, int32_t n, const double* A, double* sigma, int32_t maxsweeps, int32_t* sweeps) {
  return jacobi_selftest(device, m, n, A, sigma, maxsweeps, sweeps, true);
}

// ================================================================================================
// the batched gauge sweep
// ================================================================================================
namespace {

struct StepDims { int a, an, b, bn, r1, rows, cols, kmax; };

struct ProbPlan {
  std::vector<StepDims> st;        // [L]; entries 1 .. L-1 used
  std::vector<int64_t> lfoff;      // [L+1]
  std::vector<int32_t> rdim;       // [L+1]
  int64_t lf_doubles = 0, y_doubles = 0, z_doubles = 0, e_doubles = 0;
  int rows32_max = 0, cols_max = 0;
  // truncating sweep (planned when the kept ranks follow from the dimensions)
  std::vector<int> kc;             // [L+1]: left bond of output core t
  int64_t c_doubles = 0, t1_doubles = 0, nt_doubles = 0, mt_doubles = 0, ja_doubles = 0, u_doubles = 0;
  int rr_max = 0;
};

inline v2::Map2 lin(int64_t s) { return v2::Map2{1 << 30, s, 0}; }


}  // namespace

int v2_gather_bonds(mpbp_ctx* c, const EngProb* probs, int n, std::vector<int32_t>& hb) {
  const int L = c->L;
  hipStream_t st = c->stream;
  hb.resize((size_t)n * 2 * (L + 1));
  if (n <= 0) return MPBP_OK;
  std::vector<v2::BondSrc> src(n);
  for (int i = 0; i < n; i++) src[i] = v2::BondSrc{probs[i].bond1, probs[i].bond2};
  const size_t bsrc = (sizeof(v2::BondSrc) * n + 255) & ~size_t(255), bout = sizeof(int32_t) * hb.size();
  int rc = ensure_arena(c, c->v2arena, bsrc + bout + 4096);
  if (rc != MPBP_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(c->v2arena.base, src.data(), sizeof(v2::BondSrc) * n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(v2::k_gather_bonds, dim3(n), dim3(64), 0, st, (const v2::BondSrc*)c->v2arena.base, (int32_t*)(c->v2arena.base + bsrc), L);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(hb.data(), c->v2arena.base + bsrc, bout, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return MPBP_OK;
}


// MPBP_V2_TIMING=1: device time of the batched sweeps by section (HIP events on the stream, read once per call) on stderr
struct V2Timing {
  bool on; hipStream_t st; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[4]; hipEvent_t cur;
  explicit V2Timing(hipStream_t s) : on(getenv("MPBP_V2_TIMING") != nullptr), st(s), cur(nullptr) {}
  void begin() { if (on) { (void)hipEventCreate(&cur); (void)hipEventRecord(cur, st); } }
  void end(int cat) { if (on) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st); ev[cat].push_back({cur, e}); } }
  void report(int P, int L) {
    if (!on) return;
    (void)hipStreamSynchronize(st);
    const char* nm[4] = {"sweep 1 (assembly + QR + Lf)", "sweep 2 contractions + scaling", "sweep 2 QR of M_t", "sweep 2 Jacobi + truncation"};
    for (int k = 0; k < 4; k++) {
      double ms = 0;
      for (auto& pr : ev[k]) { float x = 0; (void)hipEventElapsedTime(&x, pr.first, pr.second); ms += x; (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
      if (!ev[k].empty()) fprintf(stderr, "[v2 timing] P=%d L=%d  %-34s %10.1f ms\n", P, L, nm[k], ms);
    }
  }
};
int v2_gauge_sweep(mpbp_ctx* c, EngProb* probs, int n, const int32_t* hb, const mpbp_trunc* trunc2, int* n_done, int* did_sweep2) {
  *did_sweep2 = 0;
  *n_done = 0;
  if (n <= 0) return MPBP_OK;
  const int L = c->L;
  hipStream_t st = c->stream;
  for (int i = 0; i < n; i++)
    if (probs[i].mirror) return c->fail(MPBP_EINVAL, "internal: mirrored problem in the batched gauge sweep");
  // ---- dimensions of every time step (host: they follow from the bond tables)
  std::vector<ProbPlan> plan(n);
  for (int i = 0; i < n; i++) {
    const EngProb& P = probs[i];
    ProbPlan& pp = plan[i];
    const int32_t* b1 = hb + (size_t)i * 2 * (L + 1);
    const int32_t* b2 = b1 + (L + 1);
    pp.st.resize(L); pp.lfoff.assign(L + 1, 0); pp.rdim.assign(L + 1, 1);
    int64_t off = 0;
    pp.lfoff[L] = off; off += 4;                      // Lf_L = [1]
    for (int t = L - 1; t >= 1; t--) {
      StepDims d;
      d.a = b1[t]; d.an = b1[t + 1]; d.b = b2[t]; d.bn = b2[t + 1];
      d.r1 = pp.rdim[t + 1];
      d.rows = d.r1 * P.ny * P.q; d.cols = d.a * d.b; d.kmax = std::min(d.rows, d.cols);
      pp.rdim[t] = d.kmax;
      pp.st[t] = d;
      pp.lfoff[t] = off; off += ((int64_t)d.kmax * d.cols + 3) & ~int64_t(3);
      pp.y_doubles = std::max<int64_t>(pp.y_doubles, (int64_t)r32i(d.rows) * (r16i(d.cols) + 16));
      pp.z_doubles = std::max<int64_t>(pp.z_doubles, (int64_t)d.a * P.ny1 * P.q * d.r1 * d.bn);
      pp.e_doubles = std::max<int64_t>(pp.e_doubles, (int64_t)P.q * d.b * P.ny * d.bn * P.ny1);
      pp.rows32_max = std::max(pp.rows32_max, r32i(d.rows)); pp.cols_max = std::max(pp.cols_max, d.cols);
    }
    pp.e_doubles = std::max<int64_t>(pp.e_doubles, (int64_t)P.q * b2[0] * P.ny * b2[1] * P.ny1);
    pp.lf_doubles = off;
    if (trunc2) {
      // sweep 2, t = 0 .. L-1: kc_0 = 1, Rr = kc ny q, kept rank min(Rr, r_{t+1}, mprime, cap_out)
      pp.kc.assign(L + 1, 1);
      for (int t = 0; t < L; t++) {
        const int a = b1[t], an = b1[t + 1], b = b2[t], bn = b2[t + 1];
        const int kc = pp.kc[t], Rr = kc * P.ny * P.q;
        const int64_t Bn = (int64_t)an * bn;
        pp.rr_max = std::max(pp.rr_max, Rr);
        pp.c_doubles = std::max<int64_t>(pp.c_doubles, (int64_t)kc * a * b);
        pp.t1_doubles = std::max<int64_t>(pp.t1_doubles, (int64_t)kc * b * an * P.ny1 * P.q);
        pp.nt_doubles = std::max<int64_t>(pp.nt_doubles, (int64_t)Rr * Bn);
        if (t == L - 1) break;
        const int r1 = pp.rdim[t + 1];
        int kp = std::min(std::min(Rr, r1), trunc2->mprime);
        kp = std::max(1, std::min(kp, (int)P.cap_out));
        pp.kc[t + 1] = kp;
        pp.c_doubles = std::max<int64_t>(pp.c_doubles, (int64_t)kp * Bn);
        pp.mt_doubles = std::max<int64_t>(pp.mt_doubles, (int64_t)r32i(r1) * (r16i(Rr) + 16));
        pp.ja_doubles = std::max<int64_t>(pp.ja_doubles, (int64_t)v2::jac_ld(Rr) * std::min(r1, Rr));
        pp.u_doubles = std::max<int64_t>(pp.u_doubles, (int64_t)Rr * kp);
        pp.rows32_max = std::max(pp.rows32_max, r32i(r1)); pp.cols_max = std::max(pp.cols_max, Rr);
      }
    }
  }
  // ---- how many problems fit
  size_t freeb = 0, totb = 0;
  hipMemGetInfo(&freeb, &totb);
  const size_t budget = (size_t)((double)(freeb + c->v2arena.cap) * 0.80);
  auto al = [](int64_t d) { return ((size_t)d * 8 + 255) & ~size_t(255); };
  int P = 0; size_t bytes = 0;
  int nchunk = 1, ntile = 1;
  std::vector<size_t> per(n);
  for (int i = 0; i < n; i++) {
    const int nc = std::max(nchunk, (plan[i].rows32_max + v2::CH - 1) / v2::CH), nt = std::max(ntile, r16i(plan[i].cols_max) / 16 + 1);
    // aux is sized by the batch maxima: recompute the total when they grow
    size_t tot = 0;
    for (int k = 0; k <= i; k++)
      tot += al(plan[k].y_doubles) + al(plan[k].z_doubles) + al(plan[k].e_doubles) + al(plan[k].lf_doubles) + al(2 * v2::auxlay_doubles(nc, nt)) +
             (((size_t)(L + 1) * 12 + 255) & ~size_t(255)) +
             (trunc2 ? 2 * al(plan[k].c_doubles) + al(plan[k].t1_doubles) + al(plan[k].nt_doubles) + al(plan[k].mt_doubles) + al(plan[k].ja_doubles) + al(plan[k].u_doubles) + 512 + (((size_t)plan[k].rr_max * 4 + 255) & ~size_t(255)) + 256 : 0);
    const size_t desc = (size_t)(i + 1) * L * (sizeof(v2::QrProb) * 2 + sizeof(v2::GemmDesc) * (4 + 2 * c->q) + sizeof(v2::EDesc) + sizeof(v2::LfDesc) + sizeof(v2::ScaleDesc) + sizeof(v2::SvdDesc)) + 65536;
    if (i > 0 && tot + desc > budget) break;
    P = i + 1; bytes = tot + desc; nchunk = nc; ntile = nt;
  }
  if (bytes > budget) return c->fail(MPBP_ENOMEM, "batched gauge sweep: one problem needs %zu MiB, %zu MiB available", bytes >> 20, budget >> 20);
  {
    int rc = ensure_arena(c, c->v2arena, bytes + 65536);
    if (rc != MPBP_OK) return rc;
  }
  const v2::AuxLay lay = v2::make_auxlay(nchunk, ntile);
  const int64_t auxd = v2::auxlay_doubles(nchunk, ntile);
  // ---- carve the arena
  char* base = c->v2arena.base; size_t used = 0;
  auto take = [&](size_t b) { char* p = base + used; used += (b + 255) & ~size_t(255); return p; };
  struct Bufs { double *Y, *Z, *E, *aux, *lf; int64_t* lfoff; int32_t* rdim; double *C0, *C1, *T1, *Nt, *Mt, *JA, *U, *scal; int32_t* act; };
  std::vector<Bufs> bf(P);
  for (int i = 0; i < P; i++) {
    bf[i].Y = (double*)take(al(plan[i].y_doubles)); bf[i].Z = (double*)take(al(plan[i].z_doubles));
    bf[i].E = (double*)take(al(plan[i].e_doubles)); bf[i].aux = (double*)take(al(2 * auxd));          /* two scratch copies: look-ahead of qr_batch */ bf[i].lf = (double*)take(al(plan[i].lf_doubles));
    char* tb = take((size_t)(L + 1) * 12);
    bf[i].lfoff = (int64_t*)tb; bf[i].rdim = (int32_t*)(tb + (size_t)(L + 1) * 8);
    if (trunc2) {
      bf[i].C0 = (double*)take(al(plan[i].c_doubles)); bf[i].C1 = (double*)take(al(plan[i].c_doubles));
      bf[i].T1 = (double*)take(al(plan[i].t1_doubles)); bf[i].Nt = (double*)take(al(plan[i].nt_doubles));
      bf[i].Mt = (double*)take(al(plan[i].mt_doubles)); bf[i].JA = (double*)take(al(plan[i].ja_doubles));
      bf[i].U = (double*)take(al(plan[i].u_doubles)); bf[i].scal = (double*)take(512);     // [0] max slot, [1] log c
      bf[i].act = (int32_t*)take(sizeof(int32_t) * (size_t)std::max(1, plan[i].rr_max));
    }
  }
  // ---- descriptors of all time steps, one upload
  const int q = probs[0].q;
  std::vector<v2::QrProb> hq((size_t)P * L);
  std::vector<v2::GemmDesc> hg1((size_t)P * L), hg2((size_t)P * L * q);
  std::vector<v2::EDesc> he((size_t)P * L);
  std::vector<v2::LfDesc> hl((size_t)P * L);
  std::vector<v2::SetOne> hone(P);
  std::vector<char> htab((size_t)P * (L + 1) * 12);
  for (int i = 0; i < P; i++) {
    const EngProb& Pr = probs[i];
    if (Pr.q != q) return c->fail(MPBP_EINVAL, "internal: mixed q in one batch");
    memcpy(htab.data() + (size_t)i * (L + 1) * 12, plan[i].lfoff.data(), (size_t)(L + 1) * 8);
    memcpy(htab.data() + (size_t)i * (L + 1) * 12 + (size_t)(L + 1) * 8, plan[i].rdim.data(), (size_t)(L + 1) * 4);
    hone[i].p = bf[i].lf + plan[i].lfoff[L];
    {
      const int32_t* b2 = hb + (size_t)i * 2 * (L + 1) + (L + 1);
      for (int t = 0; t < L; t++)      // the coupling table of every time step (the truncating sweep starts at t = 0)
        he[(size_t)t * P + i] = v2::EDesc{Pr.A2 + (int64_t)t * Pr.stride2, Pr.pyy + (int64_t)t * Pr.pyy_tstride, bf[i].E, (int)b2[t], (int)b2[t + 1], Pr.ny, Pr.ny1, Pr.ny2, q};
    }
    for (int t = 1; t < L; t++) {
      const StepDims& d = plan[i].st[t];
      const size_t k = (size_t)t * P + i;
      const int ldY = r32i(d.rows);
      hq[k] = v2::QrProb{bf[i].Y, bf[i].aux, ldY, d.rows, d.cols, d.kmax};
      hl[k].Lf = bf[i].lf + plan[i].lfoff[t];
      const int64_t zld = (int64_t)d.r1 * d.bn;
      v2::GemmDesc g{};
      g.S = Pr.A1 + (int64_t)t * Pr.stride1; g.X = bf[i].lf + plan[i].lfoff[t + 1]; g.O = bf[i].Z;
      g.M = d.a * Pr.ny1 * q; g.N = d.r1 * d.bn; g.K = d.an;
      g.sro = v2::Map2{d.a, 1, (int64_t)d.a * d.an}; g.sco = lin(d.a);
      g.xro = lin(d.r1); g.xco = v2::Map2{d.r1, 1, (int64_t)d.r1 * d.an};
      g.oro = lin(zld); g.oco = lin(1);
      hg1[k] = g;
      const int M2 = d.b * Pr.ny, K2 = d.bn * Pr.ny1;
      for (int xi = 0; xi < q; xi++) {
        v2::GemmDesc h{};
        h.S = bf[i].E + (int64_t)xi * M2 * K2; h.X = bf[i].Z + zld * d.a * Pr.ny1 * xi; h.O = bf[i].Y + (int64_t)d.r1 * Pr.ny * xi;
        h.M = M2; h.N = d.r1 * d.a; h.K = K2;
        h.sro = lin(1); h.sco = lin(M2);
        h.xro = v2::Map2{d.bn, d.r1, zld * d.a}; h.xco = v2::Map2{d.r1, 1, zld};
        h.oro = v2::Map2{d.b, (int64_t)ldY * d.a, d.r1}; h.oco = v2::Map2{d.r1, 1, ldY};
        hg2[((size_t)t * P + i) * q + xi] = h;
      }
    }
  }
  v2::QrProb* dq = (v2::QrProb*)take(sizeof(v2::QrProb) * hq.size());
  v2::GemmDesc* dg1 = (v2::GemmDesc*)take(sizeof(v2::GemmDesc) * hg1.size());
  v2::GemmDesc* dg2 = (v2::GemmDesc*)take(sizeof(v2::GemmDesc) * hg2.size());
  v2::EDesc* de = (v2::EDesc*)take(sizeof(v2::EDesc) * he.size());
  v2::LfDesc* dl = (v2::LfDesc*)take(sizeof(v2::LfDesc) * hl.size());
  v2::SetOne* done = (v2::SetOne*)take(sizeof(v2::SetOne) * hone.size());
  if (used > c->v2arena.cap) return c->fail(MPBP_ENOMEM, "internal: gauge-sweep arena accounting (%zu > %zu)", used, c->v2arena.cap);
  HIPCHK(c, hipMemcpyAsync(dq, hq.data(), sizeof(v2::QrProb) * hq.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dg1, hg1.data(), sizeof(v2::GemmDesc) * hg1.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dg2, hg2.data(), sizeof(v2::GemmDesc) * hg2.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(de, he.data(), sizeof(v2::EDesc) * he.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(dl, hl.data(), sizeof(v2::LfDesc) * hl.size(), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(done, hone.data(), sizeof(v2::SetOne) * hone.size(), hipMemcpyHostToDevice, st));
  for (int i = 0; i < P; i++)
    HIPCHK(c, hipMemcpyAsync(bf[i].lfoff, htab.data() + (size_t)i * (L + 1) * 12, (size_t)(L + 1) * 12, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipStreamSynchronize(st));     // the host vectors go out of scope below only after the loop, but keep it simple
  hipLaunchKernelGGL(v2::k_set_one, dim3((P + 63) / 64), dim3(64), 0, st, (const v2::SetOne*)done, P);
  for (int i = 0; i < P; i++)                                                         // counters, T/S of both scratch copies (the two headers)
    HIPCHK(c, hipMemsetAsync(bf[i].aux, 0, sizeof(double) * (size_t)(2 * v2::AUX_HDR), st));
  // cooperative panels only while no launch of this context has timed out (launch_engine repeats a failed batch without them)
  int* coop_err = c->no_coop_panel ? nullptr : c->d_counter + 8;
  HIPCHK(c, hipMemsetAsync(c->d_counter + 8, 0, sizeof(int), st));
  const bool force_tall = [] { const char* e = getenv("MPBP_DEBUG_FORCE_TALL"); return e && e[0] == '1'; }();
  // ---- the time steps
  std::vector<QrDims> dims(P);
  V2Timing tm(st);
  tm.begin();
  for (int t = L - 1; t >= 1; t--) {
    int maxN1 = 0, maxN2 = 0, rows32m = 0, colsm = 0; int64_t maxE = 0;
    for (int i = 0; i < P; i++) {
      const StepDims& d = plan[i].st[t];
      dims[i] = QrDims{d.rows, d.cols, d.kmax};
      maxN1 = std::max(maxN1, d.r1 * d.bn); maxN2 = std::max(maxN2, d.r1 * d.a);
      maxE = std::max<int64_t>(maxE, (int64_t)q * d.b * probs[i].ny * d.bn * probs[i].ny1);
      rows32m = std::max(rows32m, r32i(d.rows)); colsm = std::max(colsm, d.cols);
    }
    const size_t o = (size_t)t * P;
    hipLaunchKernelGGL(v2::k_build_E, dim3((unsigned)std::min<int64_t>(64, (maxE + 255) / 256), P), dim3(256), 0, st, (const v2::EDesc*)(de + o));
    hipLaunchKernelGGL(v2::k_gemm, dim3(std::min(1024, (maxN1 + 127) / 128), P), dim3(512), 0, st, (const v2::GemmDesc*)(dg1 + o));
    hipLaunchKernelGGL(v2::k_zero_pads, dim3(std::min(256, std::max(1, rows32m / 8)), P), dim3(256), 0, st, (const v2::QrProb*)(dq + o), lay);
    hipLaunchKernelGGL(v2::k_gemm, dim3(std::min(1024, (maxN2 + 127) / 128), P * q), dim3(512), 0, st, (const v2::GemmDesc*)(dg2 + o * q));
    if (qr_batch(st, dq + o, dims, lay, force_tall, coop_err, c->num_cu * 3 / 4, auxd, c->num_cu) != 0) return c->fail(MPBP_EHIP, "batched QR launch failed: %s", hipGetErrorString(hipGetLastError()));
    const int gw = std::min(256, std::max(1, (colsm + 3) / 4));
    hipLaunchKernelGGL(v2::k_maxabs, dim3(gw, P), dim3(256), 0, st, (const v2::QrProb*)(dq + o), lay);
    hipLaunchKernelGGL(v2::k_lf_write, dim3(gw, P), dim3(256), 0, st, (const v2::QrProb*)(dq + o), (const v2::LfDesc*)(dl + o), lay);
  }
  tm.end(0);
  HIPCHK(c, hipGetLastError());
  for (int i = 0; i < P; i++) { probs[i].lf = bf[i].lf; probs[i].lfoff = bf[i].lfoff; probs[i].rdim = bf[i].rdim; }
  *n_done = P;
  if (!trunc2) { tm.report(P, L); return MPBP_OK; }
  // ================================================================ the truncating sweep on the grid
  {
    std::vector<v2::GemmDesc> gn1((size_t)P * L), gn2((size_t)P * L * q), gmt((size_t)P * L), gcr((size_t)P * L);
    std::vector<v2::QrProb> q2((size_t)P * L);
    std::vector<v2::ScaleDesc> sc((size_t)P * L);
    std::vector<v2::SvdDesc> sv((size_t)P * L);
    std::vector<v2::LastDesc> last(P);
    std::vector<v2::NormDesc> nrm(P);
    std::vector<v2::SetOne> one2(P);
    for (int i = 0; i < P; i++) {
      const EngProb& Pr = probs[i];
      const int32_t* b1 = hb + (size_t)i * 2 * (L + 1);
      const int32_t* b2 = b1 + (L + 1);
      one2[i].p = bf[i].C0;
      last[i] = v2::LastDesc{bf[i].Nt, Pr.out + (int64_t)(L - 1) * Pr.ostride, Pr.obond, plan[i].kc[L - 1] * Pr.ny * q, L};
      nrm[i] = v2::NormDesc{Pr.out, Pr.obond, Pr.ostride, Pr.logz1, Pr.logz2, bf[i].scal + 2, Pr.ologz, Pr.ny * q, L};
      for (int t = 0; t < L; t++) {
        const size_t k = (size_t)t * P + i;
        const int a = b1[t], an = b1[t + 1], b = b2[t], bn = b2[t + 1];
        const int kc = plan[i].kc[t], Rr = kc * Pr.ny * q;
        const int64_t Bn = (int64_t)an * bn, tld = (int64_t)kc * b;
        double* Ccur = (t & 1) ? bf[i].C1 : bf[i].C0;
        double* Cnew = (t & 1) ? bf[i].C0 : bf[i].C1;
        v2::GemmDesc g{};
        g.S = Pr.A1 + (int64_t)t * Pr.stride1; g.X = Ccur; g.O = bf[i].T1;
        g.M = an * Pr.ny1 * q; g.N = kc * b; g.K = a;
        g.sro = v2::Map2{an, a, (int64_t)a * an}; g.sco = lin(1);
        g.xro = lin(kc); g.xco = v2::Map2{kc, 1, (int64_t)kc * a};
        g.oro = lin(tld); g.oco = lin(1);
        gn1[k] = g;
        const int M2 = b * Pr.ny, K2 = bn * Pr.ny1;
        for (int xi = 0; xi < q; xi++) {
          v2::GemmDesc h{};
          h.S = bf[i].E + (int64_t)xi * M2 * K2; h.X = bf[i].T1 + tld * an * Pr.ny1 * xi; h.O = bf[i].Nt + (int64_t)kc * Pr.ny * xi;
          h.M = bn * Pr.ny; h.N = kc * an; h.K = b * Pr.ny1;
          h.sro = v2::Map2{bn, M2, b}; h.sco = v2::Map2{b, 1, (int64_t)M2 * bn};
          h.xro = v2::Map2{b, kc, tld * an}; h.xco = v2::Map2{kc, 1, tld};
          h.oro = v2::Map2{bn, (int64_t)Rr * an, kc}; h.oco = v2::Map2{kc, 1, Rr};
          gn2[k * q + xi] = h;
        }
        sc[k] = v2::ScaleDesc{bf[i].Nt, (int64_t)Rr * Bn, bf[i].scal + (t & 1), bf[i].scal + 2, bf[i].scal + ((t + 1) & 1)};
        if (t == L - 1) continue;
        const int r1 = plan[i].rdim[t + 1], kp = plan[i].kc[t + 1];
        const int ldM = r32i(r1);
        v2::GemmDesc m{};
        m.S = bf[i].Nt; m.X = bf[i].lf + plan[i].lfoff[t + 1]; m.O = bf[i].Mt;
        m.M = Rr; m.N = r1; m.K = (int)Bn;
        m.sro = lin(1); m.sco = lin(Rr); m.xro = lin(r1); m.xco = lin(1); m.oro = lin(ldM); m.oco = lin(1);
        gmt[k] = m;
        q2[k] = v2::QrProb{bf[i].Mt, bf[i].aux, ldM, r1, Rr, std::min(r1, Rr)};
        sv[k] = v2::SvdDesc{bf[i].Mt, bf[i].JA, bf[i].U, Pr.out + (int64_t)t * Pr.ostride, Pr.obond,
                            ldM, r1, Rr, kc, kp, t, L, trunc2->kind, trunc2->mprime, Pr.cap_out, bf[i].scal, bf[i].act};
        v2::GemmDesc cr{};
        cr.S = bf[i].U; cr.X = bf[i].Nt; cr.O = Cnew;
        cr.M = kp; cr.N = (int)Bn; cr.K = Rr;
        cr.sro = lin(Rr); cr.sco = lin(1); cr.xro = lin(1); cr.xco = lin(Rr); cr.oro = lin(1); cr.oco = lin(kp);
        gcr[k] = cr;
      }
    }
    v2::GemmDesc* dn1 = (v2::GemmDesc*)take(sizeof(v2::GemmDesc) * gn1.size());
    v2::GemmDesc* dn2 = (v2::GemmDesc*)take(sizeof(v2::GemmDesc) * gn2.size());
    v2::GemmDesc* dmt = (v2::GemmDesc*)take(sizeof(v2::GemmDesc) * gmt.size());
    v2::GemmDesc* dcr = (v2::GemmDesc*)take(sizeof(v2::GemmDesc) * gcr.size());
    v2::QrProb* dq2 = (v2::QrProb*)take(sizeof(v2::QrProb) * q2.size());
    v2::ScaleDesc* dsc = (v2::ScaleDesc*)take(sizeof(v2::ScaleDesc) * sc.size());
    v2::SvdDesc* dsv = (v2::SvdDesc*)take(sizeof(v2::SvdDesc) * sv.size());
    v2::LastDesc* dlast = (v2::LastDesc*)take(sizeof(v2::LastDesc) * P);
    v2::NormDesc* dnrm = (v2::NormDesc*)take(sizeof(v2::NormDesc) * P);
    v2::SetOne* done2 = (v2::SetOne*)take(sizeof(v2::SetOne) * P);
    if (used > c->v2arena.cap) return c->fail(MPBP_ENOMEM, "internal: gauge-sweep arena accounting (%zu > %zu)", used, c->v2arena.cap);
    HIPCHK(c, hipMemcpyAsync(dn1, gn1.data(), sizeof(v2::GemmDesc) * gn1.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dn2, gn2.data(), sizeof(v2::GemmDesc) * gn2.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dmt, gmt.data(), sizeof(v2::GemmDesc) * gmt.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dcr, gcr.data(), sizeof(v2::GemmDesc) * gcr.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dq2, q2.data(), sizeof(v2::QrProb) * q2.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dsc, sc.data(), sizeof(v2::ScaleDesc) * sc.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dsv, sv.data(), sizeof(v2::SvdDesc) * sv.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dlast, last.data(), sizeof(v2::LastDesc) * P, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(dnrm, nrm.data(), sizeof(v2::NormDesc) * P, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(done2, one2.data(), sizeof(v2::SetOne) * P, hipMemcpyHostToDevice, st));
    for (int i = 0; i < P; i++) HIPCHK(c, hipMemsetAsync(bf[i].scal, 0, 64, st));
    HIPCHK(c, hipStreamSynchronize(st));
    hipLaunchKernelGGL(v2::k_set_one, dim3((P + 63) / 64), dim3(64), 0, st, (const v2::SetOne*)done2, P);
    int rrm = 1;
    for (int i = 0; i < P; i++) rrm = std::max(rrm, plan[i].rr_max);
    const int jac_grid_sweeps = [] { const char* e = getenv("MPBP_JACOBI_GRID_SWEEPS"); return e ? atoi(e) : 60; }();
    const int jac_grid_min = [] { const char* e = getenv("MPBP_JACOBI_GRID_MIN"); return e ? atoi(e) : 384; }();
    // ... and only for a handful of problems: with many, one workgroup per problem keeps every CU busy and the 659
    // launches per sweep only add latency (configs[2] shard: 498 s with the grid form on every level, 327 s without)
    const int jac_grid_maxp = [] { const char* e = getenv("MPBP_JACOBI_GRID_MAXP"); return e ? atoi(e) : 4; }();
    // Form of the Jacobi (MPBP_JACOBI_FORM = wg | grid | block forces one):
    //   wg     one workgroup per problem (wg::jacobi_rsv inside k_svd_trunc): the default for factors below 320 columns and for
    //          batches of more than 32 problems - with a CU per problem the chip is full, and no pair is met twice per sweep;
    //   block  two-level (v2::k_jac_block + k_jac_deflate): block pairs LDS resident, blocks / 2 workgroups per problem, one
    //          launch per round of the block tournament: the default for factors of >= 320 columns in batches of <= 32
    //          problems - the hub levels of configs[2], where one CU per problem rotates 400 ... 780-column factors out of HBM
    //          while the rest of the chip idles.  Round 4, one configs[2] node block on one box (profiles/r04_jacobi_forms.txt):
    //          Jacobi + truncation in the batches of 2 - 4 problems 7.5 -> 2.4 s (the grid form's regime in round 3), of 5 - 32
    //          problems 9.8 -> 6.0 s, sweep 219.8 -> 212.8 s.  Below 320 columns it does not pay: configs[3] (160 columns) 89.1
    //          against 91.7 s, configs[4] (256 columns, <= 3 problems) 26.0 / 27.0 against 24.8 / 26.5 s per iteration.  It NEEDS
    //          the deflation of wg::jacobi_rsv (BP factors: numerical rank ~2/3, most columns null after two sweeps): without
    //          it the same configs[2] block took 351 s.
    //   grid   one launch per tournament round over the grid (k_jac_round, no deflation; round 3's form for >= 384 columns and
    //          <= 4 problems): superseded by block, kept selectable and tested.
    const int jac_form = [] { const char* e = getenv("MPBP_JACOBI_FORM"); return !e ? 0 : (!strcmp(e, "wg") ? 1 : (!strcmp(e, "grid") ? 2 : (!strcmp(e, "block") ? 3 : 0))); }();
    const int jac_block_min = [] { const char* e = getenv("MPBP_JACOBI_BLOCK_MIN"); return e ? atoi(e) : 320; }();
    // ... and only while one workgroup per problem leaves most of the chip idle: with a CU per problem for a whole batch the
    // one-workgroup form is at full occupancy and does fewer rotations (no pair is met twice per sweep)
    const int jac_block_maxp = [] { const char* e = getenv("MPBP_JACOBI_BLOCK_MAXP"); return e ? atoi(e) : 32; }();
    const size_t svd_lds = sizeof(double) * (32 + (size_t)rrm + (rrm + 1) / 2 + 4);
    HIPCHK(c, hipFuncSetAttribute((const void*)v2::k_svd_trunc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)svd_lds));
    for (int t = 0; t < L; t++) {
      int maxN1 = 0, maxN2 = 0, maxNm = 0, maxNc = 0, rows32m = 0; int64_t maxE = 0, maxNt = 0;
      for (int i = 0; i < P; i++) {
        const int32_t* b1 = hb + (size_t)i * 2 * (L + 1);
        const int32_t* b2 = b1 + (L + 1);
        const int kc = plan[i].kc[t], Rr = kc * probs[i].ny * q;
        maxN1 = std::max(maxN1, kc * (int)b2[t]); maxN2 = std::max(maxN2, kc * (int)b1[t + 1]);
        maxE = std::max<int64_t>(maxE, (int64_t)q * b2[t] * probs[i].ny * b2[t + 1] * probs[i].ny1);
        maxNt = std::max<int64_t>(maxNt, (int64_t)Rr * b1[t + 1] * b2[t + 1]);
        maxNc = std::max(maxNc, (int)b1[t + 1] * (int)b2[t + 1]);
        if (t < L - 1) {
          const int r1 = plan[i].rdim[t + 1];
          dims[i] = QrDims{r1, Rr, std::min(r1, Rr)};
          maxNm = std::max(maxNm, r1); rows32m = std::max(rows32m, r32i(r1));
        }
      }
      const size_t o = (size_t)t * P;
      tm.begin();
      hipLaunchKernelGGL(v2::k_build_E, dim3((unsigned)std::min<int64_t>(64, (maxE + 255) / 256), P), dim3(256), 0, st, (const v2::EDesc*)(de + o));
      hipLaunchKernelGGL(v2::k_gemm, dim3(std::min(1024, (maxN1 + 127) / 128), P), dim3(512), 0, st, (const v2::GemmDesc*)(dn1 + o));
      hipLaunchKernelGGL(v2::k_gemm, dim3(std::min(1024, (maxN2 + 127) / 128), P * q), dim3(512), 0, st, (const v2::GemmDesc*)(dn2 + o * q));
      const unsigned gsc = (unsigned)std::min<int64_t>(256, (maxNt + 2047) / 2048);
      hipLaunchKernelGGL(v2::k_absmax, dim3(gsc, P), dim3(256), 0, st, (const v2::ScaleDesc*)(dsc + o));
      hipLaunchKernelGGL(v2::k_scale, dim3(gsc, P), dim3(256), 0, st, (const v2::ScaleDesc*)(dsc + o), c->d_stats);
      if (t == L - 1) {
        hipLaunchKernelGGL(v2::k_lastcore, dim3(P), dim3(256), 0, st, (const v2::LastDesc*)dlast);
        tm.end(1);
        break;
      }
      hipLaunchKernelGGL(v2::k_zero_pads, dim3(std::min(256, std::max(1, rows32m / 8)), P), dim3(256), 0, st, (const v2::QrProb*)(dq2 + o), lay);
      hipLaunchKernelGGL(v2::k_gemm, dim3(std::min(1024, (maxNm + 127) / 128), P), dim3(512), 0, st, (const v2::GemmDesc*)(dmt + o));
      tm.end(1); tm.begin();
      if (qr_batch(st, dq2 + o, dims, lay, force_tall, coop_err, c->num_cu * 3 / 4, auxd, c->num_cu) != 0) return c->fail(MPBP_EHIP, "batched QR launch failed: %s", hipGetErrorString(hipGetLastError()));
      // the SVD of the triangular factor: inside one workgroup, or - factors of several hundred columns - as rounds of
      // rotations over the grid (659 launches per sweep at 660 columns: 20+ workgroups rotate at once, a one-workgroup
      // tournament of that size takes 0.3 s per time step)
      int rrt = 1, k2t = 1;
      for (int i = 0; i < P; i++) { const int Rr = plan[i].kc[t] * probs[i].ny * q; rrt = std::max(rrt, Rr); k2t = std::max(k2t, std::min(Rr, plan[i].rdim[t + 1])); }
      const v2::SvdDesc* dsvt = dsv + o;
      tm.end(2); tm.begin();
      // factors of a hundred columns and more: the two-level (block) Jacobi - block pairs LDS resident, blocks / 2
      // workgroups per problem, one launch per round of the block tournament (v2::k_jac_block)
      int jnb = 0;
      if ((jac_form == 0 || jac_form == 3) && k2t >= (jac_form == 3 ? 96 : jac_block_min) && rrt <= 1024 && P <= jac_block_maxp) jnb = jac_block_nb(rrt, k2t);
      const bool jgrid = !jnb && jac_form == 2 && rrt <= 1024 && k2t >= jac_grid_min && P <= jac_grid_maxp;
      if (jnb) {
        hipLaunchKernelGGL(v2::k_svd_trunc, dim3(P), dim3(512), svd_lds, st, dsvt, c->d_stats, 1);
        const size_t jlds = sizeof(double) * ((size_t)(rrt | 1) * 2 * jnb + 32);
        int nact = k2t;                    // the most active columns of any unconverged problem (read back after every sweep:
        for (int sweep = 0; sweep < jac_grid_sweeps; sweep++) {      //  the later sweeps run over a fraction of the blocks)
          const int nblk = (nact + jnb - 1) / jnb, ne = (nblk + 1) & ~1;
          for (int r = 0; r < std::max(1, ne - 1); r++)
            hipLaunchKernelGGL(v2::k_jac_block, dim3(ne / 2, P), dim3(512), jlds, st, dsvt, r, jnb);
          hipLaunchKernelGGL(v2::k_jac_deflate, dim3(P), dim3(512), 0, st, dsvt);
          int pending[2] = {0, 0};
          hipLaunchKernelGGL(v2::k_jac_pending2, dim3(1), dim3(256), 0, st, dsvt, P, c->d_counter + 9);
          HIPCHK(c, hipMemcpyAsync(pending, c->d_counter + 9, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
          HIPCHK(c, hipStreamSynchronize(st));
          if (pending[0] == 0) break;
          nact = std::max(2, std::min(nact, pending[1]));
        }
        hipLaunchKernelGGL(v2::k_svd_trunc, dim3(P), dim3(512), svd_lds, st, dsvt, c->d_stats, 2);
      } else if (!jgrid) hipLaunchKernelGGL(v2::k_svd_trunc, dim3(P), dim3(512), svd_lds, st, dsvt, c->d_stats, 0);
      else {
        hipLaunchKernelGGL(v2::k_svd_trunc, dim3(P), dim3(512), svd_lds, st, dsvt, c->d_stats, 1);
        const int ne = (k2t + 1) & ~1;
        // up to 60 sweeps, as the one-workgroup form allows; the host looks at the convergence flags every 4 sweeps
        // (one 4-byte copy) so that the ~6-10 sweeps of the usual case are not followed by 50 sweeps of empty launches
        for (int sweep = 0; sweep < jac_grid_sweeps; sweep++) {
          for (int r = 0; r < ne - 1; r++)
            hipLaunchKernelGGL(v2::k_jac_round, dim3((ne / 2 + 15) / 16, P), dim3(512), 0, st, dsvt, r);
          hipLaunchKernelGGL(v2::k_jac_check, dim3((P + 63) / 64), dim3(64), 0, st, dsvt, P);
          if ((sweep & 3) == 3) {
            int pending = 0;
            hipLaunchKernelGGL(v2::k_jac_pending, dim3(1), dim3(256), 0, st, dsvt, P, c->d_counter + 9);
            HIPCHK(c, hipMemcpyAsync(&pending, c->d_counter + 9, sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (pending == 0) break;
          }
        }
        hipLaunchKernelGGL(v2::k_svd_trunc, dim3(P), dim3(512), svd_lds, st, dsvt, c->d_stats, 2);
      }
      tm.end(3); tm.begin();
      hipLaunchKernelGGL(v2::k_gemm, dim3(std::min(1024, (maxNc + 127) / 128), P), dim3(512), 0, st, (const v2::GemmDesc*)(dcr + o));
      tm.end(1);
    }
    hipLaunchKernelGGL(v2::k_normalize_out, dim3(P), dim3(256), 0, st, (const v2::NormDesc*)dnrm, c->d_stats);
    HIPCHK(c, hipGetLastError());
    // the host vectors of this block must outlive the copies: they were synchronised above
    *did_sweep2 = 1;
  }
  tm.report(P, L);
  return MPBP_OK;
}
