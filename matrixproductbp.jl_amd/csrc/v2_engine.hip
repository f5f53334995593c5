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
#include "v2_kernels.h"
#include "ctx.h"
#include "v2_engine.h"

namespace {

inline int r16i(int x) { return (x + 15) & ~15; }
inline int r32i(int x) { return (x + 31) & ~31; }

struct QrDims { int rows, cols, kmax; };

// Launch sequence of one R-only QR over a batch whose dimensions `dims` are known on the host.
// d_probs: device array of v2::QrProb (same order as dims).  force_tall: column-step panels even when they would fit.
int qr_batch(hipStream_t st, const v2::QrProb* d_probs, const std::vector<QrDims>& dims, const v2::AuxLay& lay,
             bool force_tall) {
  const int P = (int)dims.size();
  if (P == 0) return 0;
  int kmax_max = 0, kmax_min = 1 << 30, rows32_max = 0, cols_max = 0;
  for (const QrDims& d : dims) {
    kmax_max = std::max(kmax_max, d.kmax); kmax_min = std::min(kmax_min, d.kmax);
    rows32_max = std::max(rows32_max, r32i(d.rows)); cols_max = std::max(cols_max, d.cols);
  }
  const int nchunk = (rows32_max + v2::CH - 1) / v2::CH;
  if (nchunk > lay.nchunk) return -1;
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)v2::k_fpanel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v2::fpanel_lds_bytes(3));
    attr_done = true;
  }
  for (int jb = 0; jb < kmax_max; jb += 64) {
    const int npmax = std::min(4, (kmax_max - jb + 15) / 16);
    // register panels / one workgroup per problem for the in-block updates while the rows below the diagonal fit
    const bool tall = force_tall || rows32_max - jb > v2::CH;
    for (int p = 0; p < npmax; p++) {
      const int jp = jb + 16 * p;
      if (p > 0) {
        if (!tall) {
          switch (p) {
            case 1: hipLaunchKernelGGL(v2::k_inblock<1>, dim3(P), dim3(512), 0, st, d_probs, lay, jb); break;
            case 2: hipLaunchKernelGGL(v2::k_inblock<2>, dim3(P), dim3(512), 0, st, d_probs, lay, jb); break;
            default: hipLaunchKernelGGL(v2::k_inblock<3>, dim3(P), dim3(512), 0, st, d_probs, lay, jb); break;
          }
        } else {
          const dim3 g(1, nchunk, P);
          switch (p) {
            case 1: hipLaunchKernelGGL(v2::k_trailW<1>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1, 0);
                    hipLaunchKernelGGL(v2::k_trailU<1>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1, 0); break;
            case 2: hipLaunchKernelGGL(v2::k_trailW<2>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1, 0);
                    hipLaunchKernelGGL(v2::k_trailU<2>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1, 0); break;
            default: hipLaunchKernelGGL(v2::k_trailW<3>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1, 0);
                     hipLaunchKernelGGL(v2::k_trailU<3>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1, 0); break;
          }
        }
      }
      if (tall) {
        for (int jj = 0; jj <= 16; jj++)
          hipLaunchKernelGGL(v2::k_colstep, dim3(nchunk, P), dim3(512), 0, st, d_probs, lay, jp, jj, p);
        hipLaunchKernelGGL(v2::k_gram, dim3(nchunk, P), dim3(512), 0, st, d_probs, lay, jb, p);
        hipLaunchKernelGGL(v2::k_build_T, dim3(P), dim3(64), 0, st, d_probs, lay, jb, p);
      } else {
        hipLaunchKernelGGL(v2::k_fpanel, dim3(P), dim3(512), v2::fpanel_lds_bytes(p), st, d_probs, lay, jb, p);
      }
    }
    const int c0min = jb + 16;     // a problem with one panel left starts its trailing tiles here
    const int ntile_max = cols_max > c0min ? (cols_max - c0min + 15) / 16 : 0;
    if (ntile_max > 0) {
      // Many problems: one wave per tile pair over all rows (fused, the tuned wg::qr_trail4); few: tiles x row chunks
      // over the grid in two launches.  Problems with fewer than four panels left always take the second form.
      const int ntile4 = cols_max > jb + 64 ? (cols_max - jb - 64 + 15) / 16 : 0;
      const bool fused = npmax == 4 && ntile4 > 0 && (int64_t)P * ((ntile4 + 1) / 2) >= 1024 && !getenv("MPBP_DEBUG_NO_FUSED_TRAIL");
      static const int trail_nt = [] { const char* e = getenv("MPBP_TRAIL_NT"); return e ? atoi(e) : 2; }();
      if (fused) {
        if (trail_nt == 1) hipLaunchKernelGGL(v2::k_trail4f<1>, dim3((ntile4 + 3) / 4, P), dim3(256), 0, st, d_probs, lay, jb);
        else hipLaunchKernelGGL(v2::k_trail4f<2>, dim3((ntile4 + 7) / 8, P), dim3(256), 0, st, d_probs, lay, jb);
      }
      const int only_short = fused ? 1 : 0;
      if (!fused || kmax_min - jb < 64) {
        const dim3 g((ntile_max + 3) / 4, nchunk, P);
        switch (npmax) {
          case 1: hipLaunchKernelGGL(v2::k_trailW<1>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short);
                  hipLaunchKernelGGL(v2::k_trailU<1>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short); break;
          case 2: hipLaunchKernelGGL(v2::k_trailW<2>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short);
                  hipLaunchKernelGGL(v2::k_trailU<2>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short); break;
          case 3: hipLaunchKernelGGL(v2::k_trailW<3>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short);
                  hipLaunchKernelGGL(v2::k_trailU<3>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short); break;
          default: hipLaunchKernelGGL(v2::k_trailW<4>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short);
                   hipLaunchKernelGGL(v2::k_trailU<4>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4, only_short); break;
        }
      }
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// ================================================================================================
// self test: nprob independent rows x cols matrices through the batched QR; R[p] = [kmax x cols] (ld kmax)
// ================================================================================================
static int st2_fail(const char* what, hipError_t e) { g_create_error = std::string(what) + ": " + hipGetErrorString(e); return MPBP_EHIP; }
#define ST2CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return st2_fail(#call, e_); } while (0)

extern "C" int mpbp_selftest_qr_batched(int32_t device, int32_t rows, int32_t cols, int32_t nprob, int32_t force_tall,
                                        const double* A, double* R, double* ms_out) {
  ST2CHK(hipSetDevice(device));
  if (rows < 1 || cols < 1 || nprob < 1) { g_create_error = "bad shape"; return MPBP_EINVAL; }
  const int ld = r32i(rows), c16 = r16i(cols) + 16, kmax = std::min(rows, cols);
  const size_t per = (size_t)ld * c16;
  const int nchunk = (ld + v2::CH - 1) / v2::CH, ntile = c16 / 16;
  const v2::AuxLay lay = v2::make_auxlay(nchunk, ntile);
  const size_t auxd = (size_t)v2::auxlay_doubles(nchunk, ntile);
  double *dY = nullptr, *dAux = nullptr; v2::QrProb* dP = nullptr;
  ST2CHK(hipMalloc(&dY, sizeof(double) * per * nprob));
  ST2CHK(hipMalloc(&dAux, sizeof(double) * auxd * nprob));
  ST2CHK(hipMalloc(&dP, sizeof(v2::QrProb) * nprob));
  ST2CHK(hipMemset(dAux, 0, sizeof(double) * auxd * nprob));
  std::vector<double> Y(per, 0.0);
  std::vector<v2::QrProb> hp(nprob);
  std::vector<QrDims> dims(nprob);
  for (int p = 0; p < nprob; p++) {
    std::fill(Y.begin(), Y.end(), 0.0);
    const double* Ap = A + (size_t)p * rows * cols;
    for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) Y[i + (size_t)ld * j] = Ap[i + (size_t)rows * j];
    ST2CHK(hipMemcpy(dY + per * p, Y.data(), sizeof(double) * per, hipMemcpyHostToDevice));
    hp[p] = v2::QrProb{dY + per * p, dAux + auxd * p, ld, rows, cols, kmax};
    dims[p] = QrDims{rows, cols, kmax};
  }
  ST2CHK(hipMemcpy(dP, hp.data(), sizeof(v2::QrProb) * nprob, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  const int rc = qr_batch(0, dP, dims, lay, force_tall != 0);
  hipEventRecord(e1, 0);
  ST2CHK(hipDeviceSynchronize());
  if (rc != 0) { g_create_error = "qr_batch launch failed"; return MPBP_EHIP; }
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

int v2_gauge_sweep(mpbp_ctx* c, EngProb* probs, int n, const int32_t* hb, int* n_done) {
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
    pp.lf_doubles = off;
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
      tot += al(plan[k].y_doubles) + al(plan[k].z_doubles) + al(plan[k].e_doubles) + al(plan[k].lf_doubles) + al(v2::auxlay_doubles(nc, nt)) +
             (((size_t)(L + 1) * 12 + 255) & ~size_t(255));
    const size_t desc = (size_t)(i + 1) * L * (sizeof(v2::QrProb) + sizeof(v2::GemmDesc) * (1 + c->q) + sizeof(v2::EDesc) + sizeof(v2::LfDesc)) + 65536;
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
  struct Bufs { double *Y, *Z, *E, *aux, *lf; int64_t* lfoff; int32_t* rdim; };
  std::vector<Bufs> bf(P);
  for (int i = 0; i < P; i++) {
    bf[i].Y = (double*)take(al(plan[i].y_doubles)); bf[i].Z = (double*)take(al(plan[i].z_doubles));
    bf[i].E = (double*)take(al(plan[i].e_doubles)); bf[i].aux = (double*)take(al(auxd)); bf[i].lf = (double*)take(al(plan[i].lf_doubles));
    char* tb = take((size_t)(L + 1) * 12);
    bf[i].lfoff = (int64_t*)tb; bf[i].rdim = (int32_t*)(tb + (size_t)(L + 1) * 8);
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
    for (int t = 1; t < L; t++) {
      const StepDims& d = plan[i].st[t];
      const size_t k = (size_t)t * P + i;
      const int ldY = r32i(d.rows);
      hq[k] = v2::QrProb{bf[i].Y, bf[i].aux, ldY, d.rows, d.cols, d.kmax};
      hl[k].Lf = bf[i].lf + plan[i].lfoff[t];
      he[k] = v2::EDesc{Pr.A2 + (int64_t)t * Pr.stride2, Pr.pyy + (int64_t)t * Pr.pyy_tstride, bf[i].E, d.b, d.bn, Pr.ny, Pr.ny1, Pr.ny2, q};
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
  const bool force_tall = [] { const char* e = getenv("MPBP_DEBUG_FORCE_TALL"); return e && e[0] == '1'; }();
  // ---- the time steps
  std::vector<QrDims> dims(P);
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
    if (qr_batch(st, dq + o, dims, lay, force_tall) != 0) return c->fail(MPBP_EHIP, "batched QR launch failed: %s", hipGetErrorString(hipGetLastError()));
    const int gw = std::min(256, std::max(1, (colsm + 3) / 4));
    hipLaunchKernelGGL(v2::k_maxabs, dim3(gw, P), dim3(256), 0, st, (const v2::QrProb*)(dq + o), lay);
    hipLaunchKernelGGL(v2::k_lf_write, dim3(gw, P), dim3(256), 0, st, (const v2::QrProb*)(dq + o), (const v2::LfDesc*)(dl + o), lay);
  }
  HIPCHK(c, hipGetLastError());
  for (int i = 0; i < P; i++) { probs[i].lf = bf[i].lf; probs[i].lfoff = bf[i].lfoff; probs[i].rdim = bf[i].rdim; }
  *n_done = P;
  return MPBP_OK;
}
