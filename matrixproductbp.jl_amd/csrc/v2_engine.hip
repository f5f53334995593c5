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
  int kmax_max = 0, rows32_max = 0, cols_max = 0;
  for (const QrDims& d : dims) { kmax_max = std::max(kmax_max, d.kmax); rows32_max = std::max(rows32_max, r32i(d.rows)); cols_max = std::max(cols_max, d.cols); }
  const int nchunk = (rows32_max + v2::CH - 1) / v2::CH;
  if (nchunk > lay.nchunk) return -1;
  for (int jb = 0; jb < kmax_max; jb += 64) {
    const int npmax = std::min(4, (kmax_max - jb + 15) / 16);
    for (int p = 0; p < npmax; p++) {
      const int jp = jb + 16 * p;
      if (p > 0) {
        const dim3 g(1, nchunk, P);
        switch (p) {
          case 1: hipLaunchKernelGGL(v2::k_trailW<1>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1);
                  hipLaunchKernelGGL(v2::k_trailU<1>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1); break;
          case 2: hipLaunchKernelGGL(v2::k_trailW<2>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1);
                  hipLaunchKernelGGL(v2::k_trailU<2>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1); break;
          default: hipLaunchKernelGGL(v2::k_trailW<3>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1);
                   hipLaunchKernelGGL(v2::k_trailU<3>, g, dim3(256), 0, st, d_probs, lay, jb, 1, 1); break;
        }
      }
      const bool tall = force_tall || rows32_max - jp > v2::CH;
      if (tall) {
        for (int jj = 0; jj <= 16; jj++)
          hipLaunchKernelGGL(v2::k_colstep, dim3(nchunk, P), dim3(512), 0, st, d_probs, lay, jp, jj, p);
      } else {
        hipLaunchKernelGGL(v2::k_fpanel, dim3(P), dim3(512), 0, st, d_probs, lay, jp, p);
      }
      hipLaunchKernelGGL(v2::k_gram, dim3(nchunk, P), dim3(512), 0, st, d_probs, lay, jb, p);
      hipLaunchKernelGGL(v2::k_build_T, dim3(P), dim3(64), 0, st, d_probs, lay, jb, p);
    }
    const int c0min = jb + 16;     // a problem with one panel left starts its trailing tiles here
    const int ntile_max = cols_max > c0min ? (cols_max - c0min + 15) / 16 : 0;
    if (ntile_max > 0) {
      const dim3 g((ntile_max + 3) / 4, nchunk, P);
      switch (npmax) {
        case 1: hipLaunchKernelGGL(v2::k_trailW<1>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4);
                hipLaunchKernelGGL(v2::k_trailU<1>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4); break;
        case 2: hipLaunchKernelGGL(v2::k_trailW<2>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4);
                hipLaunchKernelGGL(v2::k_trailU<2>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4); break;
        case 3: hipLaunchKernelGGL(v2::k_trailW<3>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4);
                hipLaunchKernelGGL(v2::k_trailU<3>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4); break;
        default: hipLaunchKernelGGL(v2::k_trailW<4>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4);
                 hipLaunchKernelGGL(v2::k_trailU<4>, g, dim3(256), 0, st, d_probs, lay, jb, 0, 4); break;
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
