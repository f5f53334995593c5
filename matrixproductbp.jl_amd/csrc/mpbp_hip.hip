// libmpbp_hip.so - host side of the C ABI declared in include/mpbp_hip.h.
// Host responsibilities: compose the per-node tables from the factor primitives (prob_y_partial,
// reference src/recursive_bp_factor.jl:49-54), unroll CavityTools.cavity into a levelled DAG of `op`s
// (src/recursive_bp_factor.jl:140), lay trains out in HBM and launch the kernels of kernels.h.
#include <chrono>
#include "kernels.h"
#include "ctx.h"
#include "v2_engine.h"

thread_local std::string g_create_error;

// ================================================================================================
// creation / destruction
// ================================================================================================
extern "C" const char* mpbp_last_error(const mpbp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int mpbp_create(mpbp_ctx** out, const mpbp_desc* d) {
  if (!out || !d) { g_create_error = "null argument"; return MPBP_EINVAL; }
  *out = nullptr;
  if (d->n_nodes <= 0 || d->n_edges <= 0 || d->T < 1 || d->q < 1 || d->q > 4 || d->max_bond < 1 || !d->nbr_ptr ||
      !d->in_edge || !d->out_edge) { g_create_error = "invalid descriptor (need n_nodes,n_edges>0, T>=1, 1<=q<=4, max_bond>=1)"; return MPBP_EINVAL; }
  mpbp_ctx* c = new mpbp_ctx();
  c->N = d->n_nodes; c->E = d->n_edges; c->T = d->T; c->L = d->T + 1; c->q = d->q; c->cap = d->max_bond; c->device = d->device; c->periodic = d->periodic != 0;
  c->nbr_ptr.assign(d->nbr_ptr, d->nbr_ptr + c->N + 1);
  const int nnz = c->nbr_ptr[c->N];
  c->in_edge.assign(d->in_edge, d->in_edge + nnz);
  c->out_edge.assign(d->out_edge, d->out_edge + nnz);
  for (int p = 0; p < nnz; p++)
    if (c->in_edge[p] < 0 || c->in_edge[p] >= c->E || c->out_edge[p] < 0 || c->out_edge[p] >= c->E) {
      g_create_error = "edge id out of range"; delete c; return MPBP_EINVAL;
    }
  c->nslots = d->n_slots > 0 ? d->n_slots : c->E;
  c->slot_of_edge.resize(c->E);
  for (int e = 0; e < c->E; e++) {
    c->slot_of_edge[e] = d->slot_of_edge ? d->slot_of_edge[e] : e;
    if (c->slot_of_edge[e] < 0 || c->slot_of_edge[e] >= c->nslots) { g_create_error = "slot_of_edge out of range"; delete c; return MPBP_EINVAL; }
  }
  c->fac.resize(c->N);
  c->qnode.assign(c->N, c->q);
  c->edge_src.assign(c->E, -1); c->edge_dst.assign(c->E, -1);
  for (int i = 0; i < c->N; i++)
    for (int p = c->nbr_ptr[i]; p < c->nbr_ptr[i + 1]; p++) { c->edge_dst[c->in_edge[p]] = i; c->edge_src[c->out_edge[p]] = i; }
  c->core_stride = (int64_t)c->cap * c->cap * c->q * c->q;
  c->slot_doubles = c->core_stride * c->L;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); delete c; return MPBP_EHIP; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  v2_device_acquire(c->device); c->dev_held = true;
  auto bail = [&](const char* what, hipError_t er) { g_create_error = std::string(what) + ": " + hipGetErrorString(er); mpbp_destroy(c); return er == hipErrorOutOfMemory ? MPBP_ENOMEM : MPBP_EHIP; };
  if (d->stream) c->stream = (hipStream_t)d->stream;
  else { if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return bail("hipStreamCreate", e); c->own_stream = true; }
  if (d->ext_cores) c->d_cores = (double*)d->ext_cores;
  else { if ((e = hipMalloc(&c->d_cores, sizeof(double) * c->slot_doubles * c->nslots)) != hipSuccess) return bail("hipMalloc(message slab)", e); c->own_cores = true; }
  if (d->ext_bonds) c->d_bonds = (int32_t*)d->ext_bonds;
  else { if ((e = hipMalloc(&c->d_bonds, sizeof(int32_t) * (c->L + 1) * c->nslots)) != hipSuccess) return bail("hipMalloc(bonds)", e); c->own_bonds = true; }
  if ((e = hipMalloc(&c->d_beliefs, sizeof(double) * c->q * c->L * c->N)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&c->d_logz_node, sizeof(double) * c->N)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&c->d_logz_pos, sizeof(double) * nnz)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&c->d_stats, sizeof(EngStats))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&c->d_counter, sizeof(int) * 64)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&c->d_prof, sizeof(wg::Prof))) != hipSuccess) return bail("hipMalloc", e);
  hipMemset(c->d_prof, 0, sizeof(wg::Prof));
  if ((e = hipMalloc(&c->d_one, sizeof(double) * 4)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&c->d_ones, sizeof(int32_t) * (c->L + 2))) != hipSuccess) return bail("hipMalloc", e);
  c->ident_n = c->q * c->q;
  if ((e = hipMalloc(&c->d_ident, sizeof(double) * (c->ident_n * c->ident_n + c->q * c->q))) != hipSuccess) return bail("hipMalloc", e);
  {
    double one[4] = {1.0, 0.0, 0.0, 0.0};
    std::vector<int32_t> ones(c->L + 2, 1);
    std::vector<double> id((size_t)c->ident_n * c->ident_n + (size_t)c->q * c->q, 0.0);
    for (int i = 0; i < c->ident_n; i++) id[i + (size_t)c->ident_n * i] = 1.0;
    for (int i = 0; i < c->q; i++) id[(size_t)c->ident_n * c->ident_n + i + (size_t)c->q * i] = 1.0;     // q x q identity (generic beliefs)
    hipMemcpy(c->d_one, one, sizeof one, hipMemcpyHostToDevice);
    hipMemcpy(c->d_ones, ones.data(), sizeof(int32_t) * ones.size(), hipMemcpyHostToDevice);
    hipMemcpy(c->d_ident, id.data(), sizeof(double) * id.size(), hipMemcpyHostToDevice);
    hipMemset(c->d_beliefs, 0, sizeof(double) * c->q * c->L * c->N);
    hipMemset(c->d_logz_node, 0, sizeof(double) * c->N);
    hipMemset(c->d_logz_pos, 0, sizeof(double) * nnz);
  }
  {
    // belief trains `bp.b[i]` are kept on the device when they fit comfortably (needed by twovar_marginals /
    // autocorrelations, reference src/mpbp.jl:245-255); otherwise mpbp_get_belief_train reports EUNSUPPORTED
    const int64_t capb = (int64_t)c->ct_factor() * c->cap;
    c->bt_stride = capb * capb * c->q;
    c->bt_slot = c->bt_stride * c->L;
    size_t freeb = 0, totb = 0;
    hipMemGetInfo(&freeb, &totb);
    const size_t need = sizeof(double) * (size_t)c->bt_slot * c->N;
    if (need < freeb / 8) {
      if (hipMalloc(&c->d_btrain, need) != hipSuccess) c->d_btrain = nullptr;
      if (c->d_btrain && hipMalloc(&c->d_bbond, sizeof(int32_t) * (c->L + 1) * c->N) != hipSuccess) { hipFree(c->d_btrain); c->d_btrain = nullptr; }
      if (c->d_bbond) hipMemset(c->d_bbond, 0, sizeof(int32_t) * (c->L + 1) * c->N);
    }
  }
  c->h_logz_node.assign(c->N, 0.0); c->h_logz_pos.assign(nnz, 0.0); c->h_f.assign(c->N, 0.0);
  c->phi.assign((size_t)c->q * c->L * c->N, 1.0);
  c->psi.assign((size_t)c->q * c->q * c->L * c->E, 1.0);
  *out = c;
  int rc = mpbp_reset_messages(c);
  if (rc != MPBP_OK) { g_create_error = c->err; mpbp_destroy(c); *out = nullptr; return rc; }
  return MPBP_OK;
}

extern "C" void mpbp_destroy(mpbp_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->own_cores && c->d_cores) hipFree(c->d_cores);
  if (c->own_bonds && c->d_bonds) hipFree(c->d_bonds);
  for (void* p : {(void*)c->d_beliefs, (void*)c->d_logz_node, (void*)c->d_logz_pos, (void*)c->d_stats, (void*)c->d_counter, (void*)c->d_prof, (void*)c->d_btrain, (void*)c->d_bbond,
                  (void*)c->d_one, (void*)c->d_ones, (void*)c->d_ident, (void*)c->d_tab, (void*)c->arena.base, (void*)c->scratch.base, (void*)c->v2arena.base})
    if (p) hipFree(p);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  if (c->dev_held) v2_device_release(c->device);
  delete c;
}

extern "C" int mpbp_slab_layout(const mpbp_ctx* c, mpbp_layout* out) {
  if (!c || !out) return MPBP_EINVAL;
  out->core_slot_doubles = c->slot_doubles; out->core_stride = c->core_stride; out->bonds_per_slot = c->L + 1; out->n_slots = c->nslots;
  return MPBP_OK;
}
extern "C" int mpbp_slab_pointers(const mpbp_ctx* c, void** cores, void** bonds) {
  if (!c) return MPBP_EINVAL;
  if (cores) *cores = c->d_cores;
  if (bonds) *bonds = c->d_bonds;
  return MPBP_OK;
}
extern "C" int mpbp_set_profiling(mpbp_ctx* c, int32_t on) { if (!c) return MPBP_EINVAL; c->profiling = on < 0 ? 0 : on; return MPBP_OK; }
extern "C" int mpbp_phase_profile(mpbp_ctx* c, double* seconds, int32_t n, int32_t reset) {
  if (!c || !seconds) return MPBP_EINVAL;
  hipSetDevice(c->device);
  wg::Prof h;
  HIPCHK(c, hipMemcpy(&h, c->d_prof, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < 24; i++) seconds[i] = (double)h.t[i] * 1e-8;   // wall_clock64 ticks at 100 MHz
  if (reset) HIPCHK(c, hipMemset(c->d_prof, 0, sizeof(wg::Prof)));
  return MPBP_OK;
}

// ================================================================================================
// inputs
// ================================================================================================
extern "C" int mpbp_set_factor(mpbp_ctx* c, int32_t node, int32_t deg, const int32_t* nstates, int32_t nt,
                               const double* prob_y, const double* prob_xy, const double* prob_yy, const double* prob_y0) {
  if (!c) return MPBP_EINVAL;
  if (node < 0 || node >= c->N) return c->fail(MPBP_EINVAL, "node %d out of range", node);
  if (deg != c->nbr_ptr[node + 1] - c->nbr_ptr[node]) return c->fail(MPBP_EINVAL, "node %d: degree %d does not match the graph (%d)", node, deg, c->nbr_ptr[node + 1] - c->nbr_ptr[node]);
  if (nt != 1 && nt != c->L) return c->fail(MPBP_EINVAL, "nt must be 1 or T+1");
  if (!nstates || !prob_y || !prob_yy || !prob_y0 || (deg > 0 && !prob_xy)) return c->fail(MPBP_EINVAL, "null table");
  NodeFactor& f = c->fac[node];
  f.generic = false; f.gen_w.clear();
  f.set = true; f.deg = deg; f.nt = nt; f.ny.assign(nstates, nstates + deg + 1);
  for (int l = 0; l <= deg; l++) if (f.ny[l] < 1) return c->fail(MPBP_EINVAL, "nstates must be >= 1");
  const int q = c->q;
  const int64_t ny_sz = (int64_t)q * q * f.ny[deg];
  const int64_t xy_sz = deg > 0 ? (int64_t)deg * f.ny[1] * q * q : 0;
  f.yy_off.assign((size_t)(deg + 1) * (deg + 1), -1);
  int64_t off = 0;
  for (int d1 = 0; d1 <= deg; d1++)
    for (int d2 = 0; d2 <= deg - d1; d2++) { f.yy_off[d1 * (deg + 1) + d2] = off; off += (int64_t)f.ny[d1 + d2] * f.ny[d1] * f.ny[d2] * q; }
  f.yy_tblock = off;
  f.prob_y.assign(prob_y, prob_y + ny_sz * nt);
  f.prob_xy.assign(prob_xy, prob_xy + xy_sz * nt);
  f.prob_yy.assign(prob_yy, prob_yy + off * nt);
  f.prob_y0.assign(prob_y0, prob_y0 + (int64_t)f.ny[0] * q * nt);
  c->tables_dirty = true;
  return MPBP_OK;
}

extern "C" int mpbp_set_generic_factor(mpbp_ctx* c, int32_t node, int32_t deg, int32_t nt, const double* w) {
  if (!c) return MPBP_EINVAL;
  if (node < 0 || node >= c->N) return c->fail(MPBP_EINVAL, "node %d out of range", node);
  if (deg != c->nbr_ptr[node + 1] - c->nbr_ptr[node]) return c->fail(MPBP_EINVAL, "node %d: degree %d does not match the graph (%d)", node, deg, c->nbr_ptr[node + 1] - c->nbr_ptr[node]);
  if (nt != 1 && nt != c->L) return c->fail(MPBP_EINVAL, "nt must be 1 or T+1");
  if (!w) return c->fail(MPBP_EINVAL, "null table");
  if (c->periodic) return c->fail(MPBP_EUNSUPPORTED, "generic factors on chains periodic in time are not supported on the device");
  if (deg > KRON_MAXK) return c->fail(MPBP_EUNSUPPORTED, "generic factor of degree %d: the exhaustive update enumerates q^degree neighbour states, at most degree %d is supported", deg, KRON_MAXK);
  int64_t sz = (int64_t)c->q * c->q;
  for (int k = 0; k < deg; k++) { sz *= c->q; if (sz > ((int64_t)1 << 26)) return c->fail(MPBP_EUNSUPPORTED, "generic factor table of node %d is too large", node); }
  {
    // the limit of the exhaustive update on the device, reported where the factor is set (construction of the host
    // mirror), not at the first sweep: the embeddings are compressed by one 512-thread workgroup each, whose register
    // panel holds 2048 rows of Y_t, and the work trains are sized by the bond CAP
    int64_t bm = c->ct_factor(), bb = c->ct_factor();
    for (int k = 0; k < deg; k++) { if (k < deg - 1) bm = std::min<int64_t>(bm * c->cap, (int64_t)1 << 40); bb = std::min<int64_t>(bb * c->cap, (int64_t)1 << 40); }
    if (deg > 0 && (bm * c->q * c->q > 2048 || bb * c->q > 2048))
      return c->fail(MPBP_EUNSUPPORTED, "node %d: generic factor of degree %d with max_bond %d needs product bonds %lld / %lld (limit: q^2 x bond <= 2048 rows); "
                     "the exhaustive update is exponential in the degree - use a RecursiveBPFactor model, a smaller max_bond or fewer neighbours",
                     node, deg, c->cap, (long long)bm, (long long)bb);
  }
  NodeFactor& f = c->fac[node];
  f = NodeFactor{};
  f.set = true; f.generic = true; f.deg = deg; f.nt = nt;
  f.gen_w.assign(w, w + sz * nt);
  c->tables_dirty = true;
  return MPBP_OK;
}

extern "C" int mpbp_set_node_states(mpbp_ctx* c, const int32_t* q_node) {
  if (!c || !q_node) return MPBP_EINVAL;
  bool het = false;
  for (int i = 0; i < c->N; i++) {
    if (q_node[i] < 1 || q_node[i] > c->q) return c->fail(MPBP_EINVAL, "node %d: nstates %d outside 1 .. q = %d", i, q_node[i], c->q);
    het = het || q_node[i] != c->q;
  }
  c->qnode.assign(q_node, q_node + c->N);
  c->hetero_q = het;
  c->tables_dirty = true;
  return mpbp_reset_messages(c);          // the initial messages are uniform over the REAL states of both end nodes
}

extern "C" int mpbp_set_phi(mpbp_ctx* c, const double* phi) {
  if (!c || !phi) return MPBP_EINVAL;
  c->phi.assign(phi, phi + (size_t)c->q * c->L * c->N); c->tables_dirty = true; return MPBP_OK;
}
extern "C" int mpbp_set_psi(mpbp_ctx* c, const double* psi) {
  if (!c || !psi) return MPBP_EINVAL;
  c->psi.assign(psi, psi + (size_t)c->q * c->q * c->L * c->E); c->tables_dirty = true; return MPBP_OK;
}

extern "C" int mpbp_set_messages(mpbp_ctx* c, const int32_t* bonds, const int64_t* offsets, const double* data) {
  if (!c || !bonds || !offsets || !data) return MPBP_EINVAL;
  hipSetDevice(c->device);
  const int L = c->L, qq = c->q * c->q;
  std::vector<double> slot((size_t)c->slot_doubles);
  for (int e = 0; e < c->E; e++) {
    if (offsets[e] < 0) continue;      // negative offset: keep this edge's message (partial upload)
    const int32_t* b = bonds + (int64_t)e * (L + 1);
    if (b[0] != 1 || b[L] != 1) return c->fail(MPBP_EINVAL, "edge %d: open-chain messages need bond 1 at both ends", e);
    const double* src = data + offsets[e];
    for (int t = 0; t < L; t++) {
      if (b[t] < 1 || b[t] > c->cap || b[t + 1] > c->cap) return c->fail(MPBP_ECAPACITY, "edge %d: bond %d exceeds max_bond %d", e, std::max(b[t], b[t + 1]), c->cap);
      const int64_t n = (int64_t)b[t] * b[t + 1] * qq;
      memcpy(slot.data() + (int64_t)t * c->core_stride, src, sizeof(double) * n);
      src += n;
    }
    HIPCHK(c, hipMemcpyAsync(c->slot_cores(e), slot.data(), sizeof(double) * c->slot_doubles, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->slot_bonds(e), b, sizeof(int32_t) * (L + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return MPBP_OK;
}

extern "C" int mpbp_get_bonds(mpbp_ctx* c, int32_t* bonds) {
  if (!c || !bonds) return MPBP_EINVAL;
  hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int e = 0; e < c->E; e++)
    HIPCHK(c, hipMemcpy(bonds + (int64_t)e * (c->L + 1), c->slot_bonds(e), sizeof(int32_t) * (c->L + 1), hipMemcpyDeviceToHost));
  return MPBP_OK;
}

extern "C" int mpbp_get_messages(mpbp_ctx* c, const int64_t* offsets, double* data) {
  if (!c || !offsets || !data) return MPBP_EINVAL;
  hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int L = c->L, qq = c->q * c->q;
  std::vector<double> slot((size_t)c->slot_doubles);
  std::vector<int32_t> b(L + 1);
  for (int e = 0; e < c->E; e++) {
    if (offsets[e] < 0) continue;      // negative offset: skip this edge (partial download)
    HIPCHK(c, hipMemcpy(slot.data(), c->slot_cores(e), sizeof(double) * c->slot_doubles, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(b.data(), c->slot_bonds(e), sizeof(int32_t) * (L + 1), hipMemcpyDeviceToHost));
    double* dst = data + offsets[e];
    for (int t = 0; t < L; t++) {
      const int64_t n = (int64_t)b[t] * b[t + 1] * qq;
      memcpy(dst, slot.data() + (int64_t)t * c->core_stride, sizeof(double) * n);
      dst += n;
    }
  }
  return MPBP_OK;
}

extern "C" int mpbp_reset_messages(mpbp_ctx* c) {
  if (!c) return MPBP_EINVAL;
  hipSetDevice(c->device);
  const int L = c->L, qq = c->q * c->q;
  // uniform normalised bond-1 message: every core = 1/(q*q) -> sum over all x of the product = 1
  std::vector<double> slot((size_t)c->slot_doubles, 0.0);
  for (int t = 0; t < L; t++) for (int s = 0; s < qq; s++) slot[(int64_t)t * c->core_stride + s] = 1.0 / qq;
  std::vector<int32_t> b(L + 1, 1);
  for (int e = 0; e < c->E; e++) {
    if (c->hetero_q) {
      // flat_mpem2(q_src, q_dst, T) of the reference (src/mpbp.jl:66): uniform over the real states, zero on the padding.
      // (An edge id that no node lists - possible on the infinite graphs' compact edge sets - keeps all q states.)
      const int qs = c->edge_src[e] >= 0 ? c->qnode[c->edge_src[e]] : c->q, qd = c->edge_dst[e] >= 0 ? c->qnode[c->edge_dst[e]] : c->q;
      for (int t = 0; t < L; t++)
        for (int xd = 0; xd < c->q; xd++)
          for (int xs = 0; xs < c->q; xs++)
            slot[(int64_t)t * c->core_stride + xs + (int64_t)c->q * xd] = (xs < qs && xd < qd) ? 1.0 / (qs * qd) : 0.0;
    }
    HIPCHK(c, hipMemcpy(c->slot_cores(e), slot.data(), sizeof(double) * c->slot_doubles, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->slot_bonds(e), b.data(), sizeof(int32_t) * (L + 1), hipMemcpyHostToDevice));
  }
  return MPBP_OK;
}

// ================================================================================================
// table composition
// ================================================================================================
namespace {

struct TableStore {
  std::vector<double> data;
  std::unordered_map<std::string, int64_t> seen;
  int64_t add(const std::vector<double>& v) {
    std::string key((const char*)v.data(), v.size() * sizeof(double));
    auto it = seen.find(key);
    if (it != seen.end()) return it->second;
    // keep every table 32-byte aligned
    while (data.size() % 4) data.push_back(0.0);
    int64_t off = (int64_t)data.size();
    data.insert(data.end(), v.begin(), v.end());
    seen.emplace(std::move(key), off);
    return off;
  }
};

}  // namespace

static int build_tables(mpbp_ctx* c) {
  const int N = c->N, L = c->L, q = c->q;
  for (int i = 0; i < N; i++) if (!c->fac[i].set) return c->fail(MPBP_EINVAL, "factor of node %d was never set (mpbp_set_factor)", i);
  TableStore ts;
  const int nnz = c->nnz();
  c->pxy_off.assign(nnz, 0); c->pxy_tstride.assign(nnz, 0); c->wmsg_off.assign(nnz, 0);
  c->wbel_off.assign(N, 0); c->init_off.assign(N, 0); c->pyy_base.assign(N, 0);
  // heterogeneous nstates: the states x >= qnode[i] of node i are padding - phi and psi are zero there, and so is every
  // message entry, belief and pair belief; normalisations and free energies are those of the unpadded model
  auto QS = [&](int e) { return c->edge_src[e] >= 0 ? c->qnode[c->edge_src[e]] : q; };
  auto QD = [&](int e) { return c->edge_dst[e] >= 0 ? c->qnode[c->edge_dst[e]] : q; };
  auto PHI = [&](int i, int t, int x) { return x < c->qnode[i] ? c->phi[x + (size_t)q * (t + (size_t)L * i)] : 0.0; };
  auto PSI = [&](int e, int t, int xs, int xd) { return (xs < QS(e) && xd < QD(e)) ? c->psi[xs + (size_t)q * (xd + (size_t)q * (t + (size_t)L * e))] : 0.0; };
  for (int i = 0; i < N; i++) {
    const NodeFactor& f = c->fac[i];
    const int z = f.deg;
    if (f.generic) {
      // Generic factor (reference src/bp_core.jl:18-57, :60-93): W[t][x', x, xj, y] with y = the joint state of the other
      // neighbours (position order, first fastest) = w(x' | x_1..x_z, x) phi(x) prod_{k != j} psi_{i->k}(x, x_k); the last
      // site carries phi and the psi only (bp_core.jl:41-45).  The belief table sums over all z neighbours (qj = 1).
      int64_t qz = 1;
      for (int k = 0; k < z; k++) qz *= q;
      auto GW = [&](int t, int xn, int x, int64_t cfg) { return f.gen_w[(f.nt == 1 ? 0 : (size_t)t * q * q * qz) + xn + q * (x + (size_t)q * cfg)]; };
      for (int j = 0; j < z; j++) {
        const int p = c->nbr_ptr[i] + j;
        const int64_t nyo = qz / q;
        std::vector<double> w((size_t)L * q * q * q * nyo);
        for (int t = 0; t < L; t++)
          for (int64_t y = 0; y < nyo; y++)
            for (int xj = 0; xj < q; xj++) {
              // full configuration: the others in order, xj inserted at position j
              int64_t cfg = 0, mul = 1, yy = y; int xs[KRON_MAXK + 1];
              for (int k = 0; k < z; k++) { xs[k] = (k == j) ? xj : (int)(yy % q); if (k != j) yy /= q; cfg += mul * xs[k]; mul *= q; }
              for (int x = 0; x < q; x++) {
                double ps = (xj < QD(c->out_edge[p])) ? PHI(i, t, x) : 0.0;        // padding states of the receiving neighbour
                for (int k = 0; k < z; k++) if (k != j) ps *= PSI(c->out_edge[c->nbr_ptr[i] + k], t, x, xs[k]);
                for (int xn = 0; xn < q; xn++)
                  w[(size_t)t * q * q * q * nyo + xn + q * (x + q * (xj + (size_t)q * y))] = (t == L - 1) ? ps : ps * GW(t, xn, x, cfg);
              }
            }
        c->wmsg_off[p] = ts.add(w);
      }
      std::vector<double> w((size_t)L * q * q * qz);
      for (int t = 0; t < L; t++)
        for (int64_t y = 0; y < qz; y++) {
          int64_t yy = y; int xs[KRON_MAXK + 1];
          for (int k = 0; k < z; k++) { xs[k] = (int)(yy % q); yy /= q; }
          for (int x = 0; x < q; x++) {
            double ps = PHI(i, t, x);
            for (int k = 0; k < z; k++) ps *= PSI(c->out_edge[c->nbr_ptr[i] + k], t, x, xs[k]);
            for (int xn = 0; xn < q; xn++) w[(size_t)t * q * q * qz + xn + q * (x + (size_t)q * y)] = (t == L - 1) ? ps : ps * GW(t, xn, x, y);
          }
        }
      c->wbel_off[i] = ts.add(w);
      continue;
    }
    const int nyz = f.ny[z];
    auto PY = [&](int t, int xn, int x, int y) { return f.prob_y[(f.nt == 1 ? 0 : (size_t)t * q * q * nyz) + xn + q * (x + (size_t)q * y)]; };
    const int ny1 = z > 0 ? f.ny[1] : 1;
    auto PXY = [&](int t, int k, int y, int xk, int xi) {
      return f.prob_xy[(f.nt == 1 ? 0 : (size_t)t * z * ny1 * q * q) + (size_t)k * ny1 * q * q + y + ny1 * (xk + (size_t)q * xi)];
    };
    auto PYY = [&](int t, int d1, int d2, int y, int y1, int y2, int xi) {
      const int64_t o = f.yy_off[d1 * (z + 1) + d2];
      return f.prob_yy[(f.nt == 1 ? 0 : (size_t)t * f.yy_tblock) + o + y + f.ny[d1 + d2] * (y1 + (size_t)f.ny[d1] * (y2 + (size_t)f.ny[d2] * xi))];
    };
    // prob_yy blob as is (layout already [y][y1][y2][xi] per block, per time)
    c->pyy_base[i] = ts.add(f.prob_yy);
    // init train core [ny0][q] (time constant or not: use time 0..L-1 blocks)
    {
      std::vector<double> v((size_t)f.ny[0] * q * L);
      for (int t = 0; t < L; t++)
        for (int k = 0; k < f.ny[0] * q; k++) v[(size_t)t * f.ny[0] * q + k] = f.prob_y0[(f.nt == 1 ? 0 : (size_t)t * f.ny[0] * q) + k];
      c->init_off[i] = ts.add(v);
    }
    for (int k = 0; k < z; k++) {
      const int p = c->nbr_ptr[i] + k;
      const int eo = c->out_edge[p];
      // pxy (x psi of the out-edge i->k, indexed [xi][xk]): tab[t][y + ny1*(xk + q*xi)]
      std::vector<double> v((size_t)L * ny1 * q * q);
      for (int t = 0; t < L; t++)
        for (int xi = 0; xi < q; xi++)
          for (int xk = 0; xk < q; xk++)
            for (int y = 0; y < ny1; y++)
              v[(size_t)t * ny1 * q * q + y + ny1 * (xk + q * xi)] = PXY(t, k, y, xk, xi) * PSI(eo, t, xi, xk);
      c->pxy_off[p] = ts.add(v); c->pxy_tstride[p] = (int64_t)ny1 * q * q;
      // W for the message to neighbour k: prob_y_partial(x',x,xj,y1; d=z-1, k) * phi
      const int nyc = f.ny[z - 1];     // states of the cavity accumulator
      std::vector<double> w((size_t)L * q * q * q * nyc);
      for (int t = 0; t < L; t++)
        for (int y1 = 0; y1 < nyc; y1++)
          for (int xj = 0; xj < q; xj++)
            for (int x = 0; x < q; x++)
              for (int xn = 0; xn < q; xn++) {
                double val;
                if (xj >= QD(eo)) val = 0.0;                                      // padding states of the receiving neighbour
                else if (t == L - 1 && !c->periodic) val = PHI(i, t, x);
                else {     // (periodic chains: the last factor couples x^{T+1} to x' = x^1, recursive_bp_factor.jl:94-98)
                  double s = 0.0;
                  for (int y = 0; y < nyz; y++)
                    for (int y2 = 0; y2 < ny1; y2++) {
                      const double pyy = PYY(t, z - 1, 1, y, y1, y2, x);
                      if (pyy != 0.0) s += PY(t, xn, x, y) * PXY(t, k, y2, xj, x) * pyy;
                    }
                  val = s * PHI(i, t, x);
                }
                w[(size_t)t * q * q * q * nyc + xn + q * (x + q * (xj + (size_t)q * y1))] = val;
              }
      c->wmsg_off[p] = ts.add(w);
    }
    // W for the belief: prob_y(x',x,y,z) * phi   (qj = 1)
    {
      std::vector<double> w((size_t)L * q * q * nyz);
      for (int t = 0; t < L; t++)
        for (int y = 0; y < nyz; y++)
          for (int x = 0; x < q; x++)
            for (int xn = 0; xn < q; xn++)
              w[(size_t)t * q * q * nyz + xn + q * (x + (size_t)q * y)] = (t == L - 1 && !c->periodic) ? PHI(i, t, x) : PY(t, xn, x, y) * PHI(i, t, x);
      c->wbel_off[i] = ts.add(w);
    }
  }
  if (c->d_tab) { hipFree(c->d_tab); c->d_tab = nullptr; }
  c->tab_doubles = ts.data.size();
  HIPCHK(c, hipMalloc(&c->d_tab, sizeof(double) * std::max<size_t>(ts.data.size(), 4)));
  HIPCHK(c, hipMemcpy(c->d_tab, ts.data.data(), sizeof(double) * ts.data.size(), hipMemcpyHostToDevice));
  c->tables_dirty = false;
  return MPBP_OK;
}

// ================================================================================================
// the sweep
// ================================================================================================
namespace {

struct OpRec { int in1, in2, out, d1, d2, node, level; };   // indices into the train table

// a pair of HIP events that is destroyed on every exit path
struct EventPair {
  hipEvent_t a = nullptr, b = nullptr;
  EventPair() { hipEventCreate(&a); hipEventCreate(&b); }
  ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
  EventPair(const EventPair&) = delete;
  EventPair& operator=(const EventPair&) = delete;
};

struct EngLaunchPlan {
  std::vector<EngProb> probs; std::vector<double> cost;
  int cap1 = 1, cap2 = 1, ny1 = 1, ny2 = 1, ny = 1, q = 1, capout = 1;
  bool small = false;       // run on the single-wave engine variant (v64)
  bool ext = false;         // sweep 1 done by the batched gauge sweep: no Lf stack / Z / Y in the slots
};

// A problem goes to the single-wave engine when every QR panel of its first sweep fits the register panel of
// one wave (rows of Y_t = B*ny*q <= QR_RS*64): these are the latency-bound problems (finalisation, products with
// the bond-1 initial train, low bond dimensions).  MPBP_DEBUG_NO_SMALL=1 sends everything to the 512-thread one.
static inline bool small_problem(int64_t B, int ny, int q) {
  const bool off = [] { const char* e = getenv("MPBP_DEBUG_NO_SMALL"); return e && e[0] == '1'; }();
  return !off && B * ny * q <= v64::wg::QR_RS * 64;
}

static inline int r16h(int x) { return (x + 15) & ~15; }

// fills cfg + returns LDS bytes and per-slot scratch doubles
static void plan_cfg(const EngLaunchPlan& pl, int L, mpbp_trunc trunc, EngCfg& cfg, size_t& lds_bytes) {
  memset(&cfg, 0, sizeof cfg);
  cfg.L = L; cfg.trunc = trunc;
  const int64_t Bmax = (int64_t)pl.cap1 * pl.cap2;
  const int nmax = pl.capout * pl.ny * pl.q;
  cfg.Bmax = (int)Bmax; cfg.nmax = nmax;
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t o = off; off += (n + 15) & ~int64_t(15); return o; };
  if (pl.ext) { cfg.lf_stride = 16; cfg.off_Lf = take(16); cfg.off_Z = take(16); cfg.off_Y = take(16); }
  else {
    cfg.lf_stride = (Bmax * Bmax + 15) & ~int64_t(15);
    cfg.off_Lf = take(cfg.lf_stride * (L + 1));
    cfg.off_Z = take((int64_t)pl.cap1 * pl.ny1 * pl.q * pl.cap2 * Bmax);
    cfg.off_Y = take((int64_t)(r16h((int)(Bmax * pl.ny * pl.q)) + 32) * (r16h((int)Bmax) + 16));
  }
  cfg.off_C0 = take((int64_t)pl.capout * Bmax);
  cfg.off_C1 = take((int64_t)pl.capout * Bmax);
  cfg.off_T1 = take((int64_t)pl.cap1 * pl.ny1 * pl.q * pl.capout * pl.cap2);
  cfg.off_Nt = take((int64_t)nmax * Bmax);
  cfg.off_Mt = take((int64_t)(r16h((int)Bmax) + 32) * (r16h(nmax) + 16));
  cfg.off_JA = take((int64_t)((nmax + 15) & ~15) * nmax);       // leading dimension of the Jacobi matrix in HBM: engine.h
  cfg.off_JV = take(16);
  const int64_t nA1 = (int64_t)pl.cap1 * pl.cap1 * pl.ny1 * pl.q, nA2 = (int64_t)pl.cap2 * pl.cap2 * pl.ny2 * pl.q;
  const int64_t nE = (int64_t)pl.q * pl.cap2 * pl.ny * pl.cap2 * pl.ny1;
  cfg.off_A1c = take(nA1); cfg.off_A2c = take(nA2); cfg.off_E = take(nE);
  cfg.slot_doubles = off;
  // LDS: [qr][misc][rdim] fixed, then union{gemm tiles / QR scratch, cores+E, jacobi} (the last two if they fit)
  int64_t l = 0;
  auto ltake = [&](int64_t n) { int64_t o = l; l += (n + 3) & ~int64_t(3); return (int32_t)o; };
  cfg.lds_qr = ltake(pl.small ? v64::wg::QR_LDS_DOUBLES : v512::wg::QR_LDS_DOUBLES);
  cfg.lds_misc = ltake(32 + nmax + (nmax + 1) / 2 + 4);
  cfg.lds_rdim = ltake((L + 2 + 1) / 2 + 2);
  // 512-thread variant: one workgroup per CU, 158 of the CU's 160 KiB (the kernel has no static LDS; 150 KiB until round 3:
  // the 257 x 64 Jacobi of a product with the bond-1 init train at bond 64 - configs[4] - needs 156 KiB with the fixed
  // part and ran from HBM, 2 - 5 ms per time step); single-wave variant: four per CU
  const int64_t budget = pl.small ? (38 * 1024) / 8 : (158 * 1024) / 8;
  const int64_t base = l;
  const int64_t coresE = ((nA1 + 3) & ~3) + ((nA2 + 3) & ~3) + ((nE + 3) & ~3);
  // JA only (V is not accumulated): [Rr | 1] x min(r1, Rr) with Rr <= nmax, r1 <= Bmax
  const int64_t jac = (((int64_t)(nmax + 1) * std::min<int64_t>(nmax, Bmax) + 3) & ~3);
  // MPBP_DEBUG_FORCE_GENERIC=1 forces the large-problem paths (operands in global memory, global QR panel)
  // on small inputs so that tests can cover them
  const char* dbg = getenv("MPBP_DEBUG_FORCE_GENERIC");
  const bool force = dbg && dbg[0] == '1';
  cfg.force_generic = force ? 1 : 0;
  bool cores_fit = !force && base + coresE <= budget, jac_fit = !force && base + jac <= budget;
  if (cores_fit) { int64_t o = base; cfg.lds_A1c = (int32_t)o; o += (nA1 + 3) & ~3; cfg.lds_A2c = (int32_t)o; o += (nA2 + 3) & ~3; cfg.lds_E = (int32_t)o; }
  else { cfg.lds_A1c = cfg.lds_A2c = cfg.lds_E = -1; }
  if (jac_fit) { cfg.lds_JA = (int32_t)base; cfg.lds_JV = -1; }
  else { cfg.lds_JA = cfg.lds_JV = -1; }
  // the gemm tiles (which double as the QR's `big` scratch, WG_WAVES*512 doubles) share the same region: within a
  // time step the staged cores / E are dead once Y_t (sweep 1) or N_t (sweep 2) is built, before any gemm / QR
  const int64_t gemm_d = pl.small ? 512 : std::max<int64_t>(v512::wg::GM_LDS_DOUBLES, v512::wg::QR_BIG_DOUBLES);   // QR quad scratch / reflector staging
  cfg.lds_gemm = (int32_t)base;
  int64_t top = base + std::max(gemm_d, std::max(cores_fit ? coresE : 0, jac_fit ? jac : 0));
  lds_bytes = (size_t)top * 8;
}

}  // namespace

static int launch_engine_impl(mpbp_ctx* c, EngLaunchPlan& pl, mpbp_trunc trunc, bool count_as_orth, float* ms_orth, int* n_orth);
// MPBP_V2_TIMING=1: wall time of every engine launch (host planning + device, stream drained) on stderr
static int launch_engine(mpbp_ctx* c, EngLaunchPlan& pl, mpbp_trunc trunc, bool count_as_orth, float* ms_orth, int* n_orth) {
  static const bool tm = getenv("MPBP_V2_TIMING") != nullptr;
  if (!tm) return launch_engine_impl(c, pl, trunc, count_as_orth, ms_orth, n_orth);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = launch_engine_impl(c, pl, trunc, count_as_orth, ms_orth, n_orth);
  (void)hipStreamSynchronize(c->stream);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  bool mir = false; for (const EngProb& P : pl.probs) mir = mir || P.mirror;
  fprintf(stderr, "[engine launch] nprob=%zu caps=%dx%d->%d ny=%d q=%d small=%d grid=%d mirror=%d cavity=%d  %10.1f ms\n", pl.probs.size(), pl.cap1, pl.cap2,
          pl.capout, pl.ny, pl.q, (int)pl.small, (int)pl.ext, (int)mir, (int)count_as_orth, ms);
  return rc;
}
static int launch_engine_impl(mpbp_ctx* c, EngLaunchPlan& pl, mpbp_trunc trunc, bool count_as_orth, float* ms_orth, int* n_orth) {
  const int nprob = (int)pl.probs.size();
  if (nprob == 0) return MPBP_OK;
  // sort by decreasing cost (longest first)
  std::vector<int> idx(nprob);
  for (int i = 0; i < nprob; i++) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return pl.cost[a] > pl.cost[b]; });
  std::vector<EngProb> sorted(nprob);
  for (int i = 0; i < nprob; i++) sorted[i] = pl.probs[idx[i]];
  // The plan's capacities are upper bounds (max_bond); what the operands actually hold is on the device.  One small
  // copy of their bond tables lets the launch be sized by the real dimensions: during the first sweeps (bonds 1, 2, 4,
  // ...) a "large" level is really a batch of small problems, and slots sized by max_bond^4 would be wasted.
  std::vector<int32_t> hb;
  {
    int rc = v2_gather_bonds(c, sorted.data(), nprob, hb);
    if (rc != MPBP_OK) return rc;
  }
  int a1 = 1, a2 = 1; int64_t bmact = 1;
  {
    const int L1 = c->L + 1;
    for (int i = 0; i < nprob; i++) {
      const int32_t* b1 = hb.data() + (size_t)i * 2 * L1; const int32_t* b2 = b1 + L1;
      for (int t = 0; t < L1; t++) { a1 = std::max(a1, (int)b1[t]); a2 = std::max(a2, (int)b2[t]); bmact = std::max<int64_t>(bmact, (int64_t)b1[t] * b2[t]); }
    }
    pl.cap1 = std::min(pl.cap1, a1); pl.cap2 = std::min(pl.cap2, a2);
  }
  const bool no_small = [] { const char* e = getenv("MPBP_DEBUG_NO_SMALL"); return e && e[0] == '1'; }();
  if (!pl.small && !no_small && (int64_t)pl.cap1 * pl.cap2 * pl.ny * pl.q <= v64::wg::QR_RS * 64 && nprob > 2 * c->num_cu) pl.small = true;
  // few single-wave problems (at most two rounds of 512-thread workgroups): the 512-thread engine finishes them
  // sooner, a single wave per problem only pays off when there are enough problems to fill 4 of them per CU
  // (MPBP_DEBUG_FORCE_SMALL=1 keeps them on the single-wave engine so that small tests cover it)
  if (pl.small && nprob <= 2 * c->num_cu && !getenv("MPBP_DEBUG_FORCE_SMALL")) pl.small = false;
  // Where sweep 1 runs: in the problem's own workgroup (slot-resident Lf stack), or as the batched, grid-level gauge
  // sweep of v2_engine.hip.  MPBP_GAUGE=grid|wg forces one; otherwise the grid form is taken when a workgroup cannot
  // hold the problem (Y_t beyond one register panel's rows) or its slot would be huge.
  bool grid = false;
  if (!pl.small) {
    const int64_t Bm = (int64_t)pl.cap1 * pl.cap2;
    const char* gm = getenv("MPBP_GAUGE");
    if (gm && !strcmp(gm, "grid")) grid = true;
    else if (gm && !strcmp(gm, "wg")) grid = false;
    else grid = bmact * pl.ny * pl.q > 2048 || Bm * Bm * (c->L + 1) * 8 > (int64_t)256 << 20;
    for (const EngProb& P : sorted) if (P.mirror) grid = false;
  }
  pl.ext = grid;
  hipEvent_t e0_at_kernel = nullptr;
  auto run = [&](EngProb* ps, int np) -> int {
    EngCfg cfg; size_t lds_bytes;
    plan_cfg(pl, c->L, trunc, cfg, lds_bytes);
    // phase timers cover the 512-thread cavity launches; MPBP_PROF_SMALL=1 covers the single-wave launches instead
    static const bool prof_small = [] { const char* e = getenv("MPBP_PROF_SMALL"); return e && e[0] == '1'; }();
    cfg.prof = (c->profiling >= 2 && (prof_small ? pl.small : count_as_orth)) ? c->d_prof : nullptr;
    const void* kern = pl.small ? (const void*)v64::eng_kernel : (const void*)v512::eng_kernel;
    const int nthreads = pl.small ? 64 : 512;
    HIPCHK(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int per_cu = 1;
    if (pl.small) hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v64::eng_kernel, nthreads, lds_bytes);
    else hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v512::eng_kernel, nthreads, lds_bytes);
    if (per_cu < 1) per_cu = 1;
    int nslots = std::min(np, c->num_cu * per_cu);
    // scratch: bounded by a budget; fewer slots if needed
    size_t slot_bytes = (size_t)cfg.slot_doubles * 8;
    size_t freeb = 0, totb = 0;
    hipMemGetInfo(&freeb, &totb);
    size_t budget = c->scratch.cap + (size_t)(freeb * 0.85);
    while (nslots > 1 && (size_t)nslots * slot_bytes > budget) nslots = (nslots + 1) / 2;
    int rc = ensure_arena(c, c->scratch, (size_t)nslots * slot_bytes + sizeof(EngProb) * np + 4096);
    if (rc != MPBP_OK) return rc;
    double* d_scr = (double*)c->scratch.base;
    EngProb* d_probs = (EngProb*)(c->scratch.base + (((size_t)nslots * slot_bytes + 255) & ~size_t(255)));
    HIPCHK(c, hipMemcpyAsync(d_probs, ps, sizeof(EngProb) * np, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_counter, 0, sizeof(int), c->stream));
    if (e0_at_kernel) hipEventRecord(e0_at_kernel, c->stream);      // the workgroup form is timed kernel only (host planning excluded)
    if (pl.small) hipLaunchKernelGGL(v64::eng_kernel, dim3(nslots), dim3(64), lds_bytes, c->stream, d_probs, np, c->d_counter, cfg, d_scr, c->d_stats);
    else hipLaunchKernelGGL(v512::eng_kernel, dim3(nslots), dim3(512), lds_bytes, c->stream, d_probs, np, c->d_counter, cfg, d_scr, c->d_stats);
    (void)kern;
    HIPCHK(c, hipGetLastError());
    return MPBP_OK;
  };
  const bool timed = count_as_orth && c->profiling;
  EventPair lev;
  hipEvent_t e0 = timed ? lev.a : nullptr, e1 = lev.b;
  e0_at_kernel = (timed && !grid) ? e0 : nullptr;
  if (timed && grid) hipEventRecord(e0, c->stream);                 // batched form: gauge sweep + sweep 2 together
  if (!grid) {
    int rc = run(sorted.data(), nprob);
    if (rc != MPBP_OK) return rc;
  } else {
    for (int done = 0; done < nprob;) {
      int nd = 0;
      // The truncating sweep goes to the grid as well when its ranks are plannable (TruncBond / TruncBondMax) and a time
      // step is heavy enough to pay for its launches (M_t = N_t Lf_{t+1} of >= 0.2 Gflop); else it runs in the
      // workgroup engine on the factors just computed.  MPBP_SWEEP2=grid|wg forces one.
      bool s2grid = (trunc.kind == MPBP_TRUNC_BOND || trunc.kind == MPBP_TRUNC_BOND_MAX) &&
                    2.0 * (double)bmact * (double)bmact * pl.capout * pl.ny * pl.q >= 2e8;
      if (const char* s2 = getenv("MPBP_SWEEP2")) { if (!strcmp(s2, "grid")) s2grid = trunc.kind == MPBP_TRUNC_BOND || trunc.kind == MPBP_TRUNC_BOND_MAX; else if (!strcmp(s2, "wg")) s2grid = false; }
      int did2 = 0;
      // The cooperative panel kernel (k_colsteps_coop) assumes that this process owns the GPU: its row-chunk workgroups
      // wait for each other inside one launch.  If they are not co-resident (a second process or stream on the device)
      // an arrival counter times out; the kernel then leaves Y untouched and raises a flag.  Nothing of a batch is
      // committed to the message slab before this point (products live in the work arena), so the batch is simply
      // repeated with one launch per column step, and the context stays in that mode.
      for (int attempt = 0;; attempt++) {
        EngStats sbak;
        const bool may_retry = !c->no_coop_panel;
        if (may_retry) HIPCHK(c, hipMemcpyAsync(&sbak, c->d_stats, sizeof sbak, hipMemcpyDeviceToHost, c->stream));
        int rc = v2_gauge_sweep(c, sorted.data() + done, nprob - done, hb.data() + (size_t)done * 2 * (c->L + 1), s2grid ? &trunc : nullptr, &nd, &did2);
        if (rc != MPBP_OK) return rc;
        if (!did2) {
          rc = run(sorted.data() + done, nd);
          if (rc != MPBP_OK) return rc;
        }
        // the triangular factors live in c->v2arena until sweep 2 has consumed them
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int herr = 0;
        HIPCHK(c, hipMemcpy(&herr, c->d_counter + 8, sizeof(int), hipMemcpyDeviceToHost));
        const bool inject = getenv("MPBP_DEBUG_COOP_FAIL_ONCE") != nullptr;            // test hook: pretend the first attempt timed out
        if (inject && may_retry && attempt == 0) herr = 1;
        if (!herr) break;
        if (!may_retry) return c->fail(MPBP_EHIP, "batched gauge sweep: arrival counter timed out although the cooperative panel kernel was not used");
        c->no_coop_panel = true;
        HIPCHK(c, hipMemcpy(c->d_stats, &sbak, sizeof sbak, hipMemcpyHostToDevice));    // the repeated batch counts once
      }
      done += nd;
    }
  }
  if (e0) {
    hipEventRecord(e1, c->stream); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); *ms_orth += ms; *n_orth += 1;
  }
  // the host vectors `sorted` must outlive the async copy
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPBP_OK;
}

// internal: the work trains of this node list do not fit the device - the caller splits the list
#define MPBP_ESPLIT_INTERNAL (-1000)

// One pass over `nodes` (validated by mpbp_sweep).  Reads the incoming messages from c->read_cores / read_bonds (the
// live slab, or the snapshot taken by mpbp_sweep when a Jacobi sweep has to be split), writes the live slab.
static int sweep_nodes(mpbp_ctx* c, const int32_t* nodes, int32_t n_nodes, mpbp_trunc trunc, double damp, mpbp_stats* stats, bool may_split) {
  const int L = c->L, q = c->q, cap = c->cap;
  EventPair evs;
  hipEvent_t ev0 = evs.a, ev1 = evs.b;
  hipEventRecord(ev0, c->stream);
  HIPCHK(c, hipMemsetAsync(c->d_stats, 0, sizeof(EngStats), c->stream));
  float ms_orth = 0.f; int n_orth = 0;

  // ---------------------------------------------------------------- plan: trains, ops, levels
  struct TrainSpec { int cap, ny, qphys; int d; int level; int64_t tab_off; bool is_init; int node; int kron_skip = -2; };   // kron_skip >= -1: product of the node's in-messages except that position (generic factors)
  std::vector<TrainSpec> specs;     // scratch trains to allocate
  std::vector<DevTrain> tr;         // filled after allocation (same indexing)
  auto new_train = [&](int capv, int ny, int qphys, int d, int level) {
    specs.push_back({capv, ny, qphys, d, level, 0, false, -1});
    return (int)specs.size() - 1;
  };
  // generic factors (mpbp_set_generic_factor): cap^n bond and q^n joint neighbour states of a product of n messages
  auto ipow = [](int64_t b, int n) { int64_t r = 1; for (int k = 0; k < n; k++) { r *= b; if (r > ((int64_t)1 << 40)) break; } return r; };
  bool any_generic = false;
  std::vector<OpRec> ops;
  struct NodePlan { int node; std::vector<int> src, dest; int full; int init; };
  std::vector<NodePlan> plans(n_nodes);
  int maxlevel = 0;
  for (int k = 0; k < n_nodes; k++) {
    const int i = nodes[k];
    const NodeFactor& f = c->fac[i];
    const int z = f.deg;
    NodePlan& P = plans[k];
    P.node = i;
    if (f.generic) {
      // Exhaustive-trace update (reference src/bp_core.jl:18-93, src/mpbp.jl:117-154): no cavity; the message to neighbour j
      // sums over the joint states of the other z-1 neighbours, the belief over all z.  The mirrored engine compresses the
      // embeddings in one 512-thread workgroup each, whose register panel holds 2048 rows of Y_t.
      any_generic = true;
      const int64_t bm = ipow(cap, z - 1) * c->ct_factor(), bb = ipow(cap, z) * c->ct_factor();
      if (z > 0 && (bm * q * q > 2048 || bb * q > 2048))
        return c->fail(MPBP_EUNSUPPORTED, "node %d: generic factor of degree %d with max_bond %d needs product bonds %lld / %lld (limit: q^2 x bond <= 2048 rows); "
                       "the exhaustive update is exponential in the degree - use a RecursiveBPFactor model, a smaller max_bond or fewer neighbours",
                       i, z, cap, (long long)bm, (long long)bb);
      P.init = -1;
      P.dest.assign(z, -1);
      for (int j = 0; j < z; j++) {
        P.dest[j] = new_train((int)ipow(cap, z - 1), (int)ipow(q, z - 1), q, 0, 0);
        specs[P.dest[j]].node = i; specs[P.dest[j]].kron_skip = j;
      }
      P.full = new_train((int)ipow(cap, z), (int)ipow(q, z), q, 0, 0);
      specs[P.full].node = i; specs[P.full].kron_skip = -1;
      continue;
    }
    P.init = new_train(1, f.ny[0], q, 0, 0);
    specs[P.init].is_init = true; specs[P.init].node = i;
    for (int j = 0; j < z; j++) P.src.push_back(new_train(cap, f.ny[1], q, 1, 0));
    auto op = [&](int a, int b) {
      const int d1 = specs[a].d, d2 = specs[b].d;
      const int lev = std::max(specs[a].level, specs[b].level) + 1;
      const int o = new_train(cap, f.ny[d1 + d2], q, d1 + d2, lev);
      ops.push_back({a, b, o, d1, d2, i, lev});
      maxlevel = std::max(maxlevel, lev);
      return o;
    };
    // CavityTools.cavity (3z-2 calls in this order)
    P.dest.assign(z, -1);
    if (z == 0) P.full = P.init;
    else if (z == 1) { P.dest[0] = P.init; P.full = op(P.src[0], P.init); }
    else {
      P.dest[0] = P.src[0];
      for (int j = 1; j < z; j++) P.dest[j] = op(P.dest[j - 1], P.src[j]);
      P.full = op(P.dest[z - 1], P.init);
      int right = P.init;
      for (int j = z - 1; j >= 1; j--) { P.dest[j] = op(P.dest[j - 1], right); right = op(P.src[j], right); }
      P.dest[0] = right;
    }
  }
  // finalisation trains: ctilde (bond 2*cap... = q*cap), engine output (message), belief ctilde
  struct FinRec { int k, j, p, src, ct, out; int nrm = -1, sum = -1, out2 = -1, occ = 0; bool gen = false; };
  std::vector<FinRec> fins; std::vector<FinRec> bels;
  const int capct = c->ct_factor() * cap;
  for (int k = 0; k < n_nodes; k++) {
    NodePlan& P = plans[k];
    const int i = P.node; const int z = c->fac[i].deg;
    const bool gen = c->fac[i].generic;
    for (int j = 0; j < z; j++) {
      const int p = c->nbr_ptr[i] + j;
      const int ct = new_train(gen ? c->ct_factor() * specs[P.dest[j]].cap : capct, q * q, 1, 0, 0);      // explicit cores: ny = q*qj, engine q = 1
      const int out = new_train(cap, q * q, 1, 0, 0);
      FinRec fr{k, j, p, P.dest[j], ct, out};
      fr.gen = gen;
      if (damp > 0.0 && !gen) {        // the generic update assigns the new message without damping (reference src/mpbp.jl:131)
        fr.nrm = new_train(cap, q * q, 1, 0, 0);            // normalised new message
        fr.sum = new_train(2 * cap, q * q, 1, 0, 0);        // new + damp/(1-damp) * old  (direct sum)
        fr.out2 = new_train(cap, q * q, 1, 0, 0);           // compressed again
        for (int j2 = 0; j2 < j; j2++) if (c->out_edge[c->nbr_ptr[i] + j2] == c->out_edge[p]) fr.occ++;
      }
      fins.push_back(fr);
    }
    const int ctb = new_train(gen ? c->ct_factor() * specs[P.full].cap : capct, q, 1, 0, 0);           // belief: qj = 1
    FinRec br{k, -1, -1, P.full, ctb, -1};
    br.gen = gen;
    if (gen) br.out = new_train(cap, q, 1, 0, 0);           // the generic belief is compressed before it is marginalised (src/mpbp.jl:145-154)
    bels.push_back(br);
  }
  // ---------------------------------------------------------------- allocate the arena
  c->arena.reset();
  tr.resize(specs.size());
  auto bytes_of = [&](const TrainSpec& s) {
    int64_t stride = ((int64_t)s.cap * s.cap * s.ny * s.qphys + 3) & ~int64_t(3);
    return std::pair<int64_t, int64_t>(stride, stride * L);
  };
  if (may_split && n_nodes > 1) {
    // MPBP_DEBUG_SPLIT_NODES=k: take the split path whenever a pass lists more than k nodes (tests of the snapshot logic;
    // read on every call so that a test can switch it)
    const char* e = getenv("MPBP_DEBUG_SPLIT_NODES");
    const int dbg_split = e ? atoi(e) : 0;
    if (dbg_split > 0 && n_nodes > dbg_split) return MPBP_ESPLIT_INTERNAL;
  }
  for (int pass = 0; pass < 2; pass++) {
    c->arena.reset();
    bool ok = true;
    for (size_t s = 0; s < specs.size(); s++) {
      tr[s].cap = specs[s].cap; tr[s].ny = specs[s].ny; tr[s].d = specs[s].d; tr[s].level = specs[s].level;
      if (specs[s].is_init) {
        // the init train (src/recursive_bp_factor.jl:133-138) is read straight from the table blob:
        // cores [1,1,ny0,q] per time = prob_y0, all bonds 1, z = 1
        const int i = specs[s].node;
        tr[s].cores = c->d_tab + c->init_off[i]; tr[s].stride = (int64_t)c->fac[i].ny[0] * q;
        tr[s].bonds = c->d_ones; tr[s].logz = c->d_one + 1;
        continue;
      }
      auto bs = bytes_of(specs[s]);
      tr[s].stride = bs.first;
      tr[s].cores = (double*)c->arena.take(sizeof(double) * bs.second);
      tr[s].bonds = (int32_t*)c->arena.take(sizeof(int32_t) * (L + 2));
      tr[s].logz = (double*)c->arena.take(sizeof(double) * 2);
      if (!tr[s].cores || !tr[s].bonds || !tr[s].logz) ok = false;
    }
    if (ok) break;
    if (pass == 1) return c->fail(MPBP_ENOMEM, "work arena too small");
    if (may_split && n_nodes > 1) {
      // every work train of the pass is resident at once (config 3 on one GPU: 300 GB for the 2048 nodes): past 45 %
      // of what the context can get, the node list is split and the halves run one after the other
      size_t freeb = 0, totb = 0;
      hipMemGetInfo(&freeb, &totb);
      const size_t reach = freeb + c->arena.cap + c->scratch.cap + c->v2arena.cap;
      if (c->arena.want > (size_t)(0.45 * (double)reach)) return MPBP_ESPLIT_INTERNAL;
    }
    int rc = ensure_arena(c, c->arena, c->arena.want + (1 << 20));
    if (rc != MPBP_OK) return rc;
  }
  // ---------------------------------------------------------------- prep: sources from the messages
  {
    std::vector<PrepProb> pp;
    for (int k = 0; k < n_nodes; k++) {
      const int i = plans[k].node; const NodeFactor& f = c->fac[i];
      if (f.generic) continue;
      for (int j = 0; j < f.deg; j++) {
        const int p = c->nbr_ptr[i] + j;
        const int ein = c->in_edge[p];
        DevTrain& o = tr[plans[k].src[j]];
        PrepProb P{};
        P.msg = c->read_slot_cores(ein); P.mbond = c->read_slot_bonds(ein); P.mstride = c->core_stride;
        P.tab = c->d_tab + c->pxy_off[p]; P.tab_tstride = c->pxy_tstride[p];
        P.out = o.cores; P.obond = o.bonds; P.ostride = o.stride; P.ologz = o.logz; P.ny1 = f.ny[1]; P.q = q;
        pp.push_back(P);
      }
    }
    if (!pp.empty()) {
      int rc = ensure_arena(c, c->scratch, sizeof(PrepProb) * pp.size() + 4096);
      if (rc != MPBP_OK) return rc;
      HIPCHK(c, hipMemcpyAsync(c->scratch.base, pp.data(), sizeof(PrepProb) * pp.size(), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(prep_kernel, dim3(L, (unsigned)pp.size()), dim3(256), 0, c->stream, (const PrepProb*)c->scratch.base, L);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
  }
  // ---------------------------------------------------------------- generic factors: products of the in-messages
  if (any_generic) {
    std::vector<KronProb> kp;
    for (size_t s2 = 0; s2 < specs.size(); s2++) {
      if (specs[s2].kron_skip < -1) continue;
      const int i = specs[s2].node;
      KronProb K{};
      K.nk = 0; K.q = q; K.mstride = c->core_stride;
      for (int j = 0; j < c->fac[i].deg; j++) {
        if (j == specs[s2].kron_skip) continue;
        const int ein = c->in_edge[c->nbr_ptr[i] + j];
        K.msg[K.nk] = c->read_slot_cores(ein); K.mbond[K.nk] = c->read_slot_bonds(ein); K.nk++;
      }
      K.out = tr[s2].cores; K.obond = tr[s2].bonds; K.ostride = tr[s2].stride; K.ologz = tr[s2].logz;
      kp.push_back(K);
    }
    int rc = ensure_arena(c, c->scratch, sizeof(KronProb) * kp.size() + 4096);
    if (rc != MPBP_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->scratch.base, kp.data(), sizeof(KronProb) * kp.size(), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(kron_kernel, dim3(L, (unsigned)kp.size()), dim3(256), 0, c->stream, (const KronProb*)c->scratch.base, L);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  // ---------------------------------------------------------------- cavity ops
  // The single-wave problems go level by level.  The 512-thread problems are packed into launches by readiness
  // rather than by level: a launch takes the problems whose inputs exist, cut to a whole number of rounds of the
  // resident workgroups when more are ready (problems that others wait for first), so that a rank with few nodes
  // (128 nodes: 128 + 384 problems in two levels) fills the CUs in 2 rounds instead of 1 + 2 half-empty ones.
  // This needs the 512-thread problems to read single-wave results of level 1 only (products with the bond-1
  // initial train) - checked, with the plain level-by-level order as the fallback.
  {
    auto is_small = [&](const OpRec& o) { return small_problem((int64_t)tr[o.in1].cap * tr[o.in2].cap, tr[o.out].ny, q); };
    std::vector<char> train_small(specs.size(), 0), avail(specs.size(), 1);
    std::vector<int> train_level(specs.size(), 0);
    bool pack_ok = getenv("MPBP_DEBUG_NO_PACK") == nullptr;
    for (const OpRec& o : ops) { train_small[o.out] = is_small(o) ? 1 : 0; train_level[o.out] = o.level; avail[o.out] = 0; }
    for (const OpRec& o : ops)
      if (!is_small(o))
        for (int in : {o.in1, o.in2})
          if (train_level[in] > 1 && train_small[in]) pack_ok = false;
    auto add_problem = [&](EngLaunchPlan& pl, const OpRec& o) {
      const NodeFactor& f = c->fac[o.node];
      const DevTrain &a = tr[o.in1], &b = tr[o.in2], &out = tr[o.out];
      EngProb P{};
      P.A1 = a.cores; P.bond1 = a.bonds; P.stride1 = a.stride; P.ny1 = a.ny;
      P.A2 = b.cores; P.bond2 = b.bonds; P.stride2 = b.stride; P.ny2 = b.ny;
      P.logz1 = a.logz; P.logz2 = b.logz;
      P.pyy = c->d_tab + c->pyy_base[o.node] + f.yy_off[o.d1 * (f.deg + 1) + o.d2];
      P.pyy_tstride = f.nt == 1 ? 0 : f.yy_tblock;
      P.ny = out.ny; P.q = q; P.mirror = 0; P.cap_out = cap;
      P.out = out.cores; P.obond = out.bonds; P.ostride = out.stride; P.ologz = out.logz;
      pl.probs.push_back(P);
      const double B = (double)a.cap * b.cap;
      pl.cost.push_back(B * B * B * out.ny);
      pl.cap1 = std::max(pl.cap1, a.cap); pl.cap2 = std::max(pl.cap2, b.cap);
      pl.ny1 = std::max(pl.ny1, a.ny); pl.ny2 = std::max(pl.ny2, b.ny); pl.ny = std::max(pl.ny, out.ny);
    };
    auto small_level = [&](int lev) -> int {
      EngLaunchPlan pls; pls.q = q; pls.capout = cap; pls.small = true;
      for (const OpRec& o : ops) if (o.level == lev && is_small(o)) { add_problem(pls, o); avail[o.out] = 1; }
      return launch_engine(c, pls, trunc, false, &ms_orth, &n_orth);
    };
    if (pack_ok) {
      int rc = small_level(1);
      if (rc != MPBP_OK) return rc;
      std::vector<int> rem;                       // indices into ops of the 512-thread problems not launched yet
      for (int k = 0; k < (int)ops.size(); k++) if (!is_small(ops[k])) rem.push_back(k);
      std::vector<char> needed(specs.size(), 0);  // trains some remaining 512-thread problem reads
      const int round = std::max(1, c->num_cu);
      while (!rem.empty()) {
        std::vector<int> ready, later;
        for (int k : rem) (avail[ops[k].in1] && avail[ops[k].in2] ? ready : later).push_back(k);
        if (ready.empty()) return c->fail(MPBP_EINVAL, "internal: cavity dependency graph is not acyclic");
        if ((int)ready.size() > round && !later.empty()) {
          std::fill(needed.begin(), needed.end(), 0);
          for (int k : later) { needed[ops[k].in1] = 1; needed[ops[k].in2] = 1; }
          std::stable_sort(ready.begin(), ready.end(), [&](int a, int b) { return needed[ops[a].out] > needed[ops[b].out]; });
          const size_t take = (ready.size() / round) * round;
          later.insert(later.end(), ready.begin() + take, ready.end());
          ready.resize(take);
        }
        EngLaunchPlan plb; plb.q = q; plb.capout = cap;
        for (int k : ready) add_problem(plb, ops[k]);
        rc = launch_engine(c, plb, trunc, true, &ms_orth, &n_orth);
        if (rc != MPBP_OK) return rc;
        for (int k : ready) avail[ops[k].out] = 1;
        rem.swap(later);
      }
      // the single-wave problems above level 1 (products with the bond-1 init train: dest[z-1], full): by readiness as well -
      // everything they read exists by now, so they share ONE launch instead of one per level (each launch is a sequential
      // chain of T + 1 time steps: configs[4] 0.75 s, configs[3] ~0.3 s per level)
      std::vector<int> srem;
      for (int k = 0; k < (int)ops.size(); k++) if (is_small(ops[k]) && ops[k].level >= 2) srem.push_back(k);
      while (!srem.empty()) {
        std::vector<int> ready, later;
        for (int k : srem) (avail[ops[k].in1] && avail[ops[k].in2] ? ready : later).push_back(k);
        if (ready.empty()) return c->fail(MPBP_EINVAL, "internal: cavity dependency graph is not acyclic");
        EngLaunchPlan pls; pls.q = q; pls.capout = cap; pls.small = true;
        for (int k : ready) add_problem(pls, ops[k]);
        rc = launch_engine(c, pls, trunc, false, &ms_orth, &n_orth);
        if (rc != MPBP_OK) return rc;
        for (int k : ready) avail[ops[k].out] = 1;
        srem.swap(later);
      }
    } else {
      for (int lev = 1; lev <= maxlevel; lev++) {
        EngLaunchPlan plb; plb.q = q; plb.capout = cap;
        for (const OpRec& o : ops) if (o.level == lev && !is_small(o)) add_problem(plb, o);
        int rc = launch_engine(c, plb, trunc, true, &ms_orth, &n_orth);
        if (rc != MPBP_OK) return rc;
        rc = small_level(lev);
        if (rc != MPBP_OK) return rc;
      }
    }
  }
  // ---------------------------------------------------------------- finalise messages + beliefs
  {
    std::vector<CtProb> cps;
    for (const FinRec& fr : fins) {
      const int i = plans[fr.k].node; const NodeFactor& f = c->fac[i];
      const DevTrain &s = tr[fr.src], &ct = tr[fr.ct];
      CtProb P{};
      P.in = s.cores; P.ibond = s.bonds; P.istride = s.stride; P.ilogz = s.logz; P.ny = s.ny;
      P.W = c->d_tab + c->wmsg_off[fr.p]; P.q = q; P.qj = q; P.periodic = c->periodic ? 1 : 0;
      P.out = ct.cores; P.obond = ct.bonds; P.ostride = ct.stride; P.ologz = ct.logz;
      (void)f;
      cps.push_back(P);
    }
    for (const FinRec& fr : bels) {
      const int i = plans[fr.k].node;
      const DevTrain &s = tr[fr.src], &ct = tr[fr.ct];
      CtProb P{};
      P.in = s.cores; P.ibond = s.bonds; P.istride = s.stride; P.ilogz = s.logz; P.ny = s.ny;
      P.W = c->d_tab + c->wbel_off[i]; P.q = q; P.qj = 1; P.periodic = c->periodic ? 1 : 0;
      P.out = ct.cores; P.obond = ct.bonds; P.ostride = ct.stride; P.ologz = ct.logz;
      cps.push_back(P);
    }
    if (!cps.empty()) {
      int rc = ensure_arena(c, c->scratch, sizeof(CtProb) * cps.size() + 4096);
      if (rc != MPBP_OK) return rc;
      HIPCHK(c, hipMemcpyAsync(c->scratch.base, cps.data(), sizeof(CtProb) * cps.size(), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(ctilde_kernel, dim3(L, (unsigned)cps.size()), dim3(256), 0, c->stream, (const CtProb*)c->scratch.base, L);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    // engine (mirror): mpem2 |> compress!(:left) |> normalize_eachmatrix!
    EngLaunchPlan pl; pl.q = 1; pl.capout = cap; pl.cap1 = capct; pl.cap2 = 1; pl.ny1 = q * q; pl.ny2 = 1; pl.ny = q * q;
    pl.small = small_problem(capct, q * q, 1);
    // generic factors: embeddings of bond q cap^(z-1) (messages, ny = q qj) and q cap^z (beliefs, ny = q), own launches
    EngLaunchPlan plg = pl, plgb = pl;
    plg.cap1 = 1; plgb.cap1 = 1; plgb.ny1 = q; plgb.ny = q;
    auto mirror_problem = [&](EngLaunchPlan& plan, const FinRec& fr, int nyv, const double* ident) {
      const DevTrain &ct = tr[fr.ct], &out = tr[fr.out];
      EngProb P{};
      P.A1 = ct.cores; P.bond1 = ct.bonds; P.stride1 = ct.stride; P.ny1 = nyv;
      P.A2 = c->d_one; P.bond2 = c->d_ones; P.stride2 = 0; P.ny2 = 1;
      P.logz1 = ct.logz; P.logz2 = nullptr;
      P.pyy = ident; P.pyy_tstride = 0;
      P.ny = nyv; P.q = 1; P.mirror = 1; P.cap_out = cap;
      P.out = out.cores; P.obond = out.bonds; P.ostride = out.stride; P.ologz = out.logz;
      plan.probs.push_back(P); plan.cost.push_back((double)ct.cap);
      plan.cap1 = std::max(plan.cap1, ct.cap);
    };
    for (const FinRec& fr : fins) mirror_problem(fr.gen ? plg : pl, fr, q * q, c->d_ident);
    for (const FinRec& fr : bels) if (fr.gen) mirror_problem(plgb, fr, q, c->d_ident + (size_t)c->ident_n * c->ident_n);
    int rc = launch_engine(c, pl, trunc, false, &ms_orth, &n_orth);
    if (rc != MPBP_OK) return rc;
    for (EngLaunchPlan* pg : {&plg, &plgb}) {
      if (pg->probs.empty()) continue;
      pg->small = small_problem(pg->cap1, pg->ny, 1);
      rc = launch_engine(c, *pg, trunc, false, &ms_orth, &n_orth);
      if (rc != MPBP_OK) return rc;
    }
    // env: normalize! the messages into the slab (damp = 0) or into a temporary (damp > 0);
    //      beliefs marginals + log z_i
    std::vector<EnvProb> eps;
    size_t rv_doubles = 0;
    std::vector<size_t> rv_off;
    for (const FinRec& fr : fins) {
      const int i = plans[fr.k].node; const int z = c->fac[i].deg;
      const int eo = c->out_edge[fr.p];
      // last occurrence of an aliased out-edge wins (reference src/recursive_bp_factor.jl:154-159)
      bool last = true;
      for (int j2 = fr.j + 1; j2 < z; j2++) if (c->out_edge[c->nbr_ptr[i] + j2] == eo) last = false;
      const DevTrain& out = tr[fr.out];
      EnvProb P{};
      P.in = out.cores; P.ibond = out.bonds; P.istride = out.stride; P.ilogz = out.logz; P.p = q * q;
      if (damp > 0.0 && !fr.gen) { const DevTrain& nm = tr[fr.nrm]; P.dst = nm.cores; P.dbond = nm.bonds; P.dstride = nm.stride; }
      else { P.dst = last ? c->slot_cores(eo) : nullptr; P.dbond = last ? c->slot_bonds(eo) : nullptr; P.dstride = c->core_stride; }
      P.marg = nullptr; P.logz_out = c->d_logz_pos + fr.p; P.bmax = cap;
      rv_off.push_back(rv_doubles); rv_doubles += (size_t)(L + 1) * cap;
      eps.push_back(P);
    }
    for (const FinRec& fr : bels) {
      const int i = plans[fr.k].node;
      const DevTrain& ct = fr.gen ? tr[fr.out] : tr[fr.ct];
      EnvProb P{};
      P.in = ct.cores; P.ibond = ct.bonds; P.istride = ct.stride; P.ilogz = ct.logz; P.p = q;
      P.dst = c->d_btrain ? c->d_btrain + (int64_t)i * c->bt_slot : nullptr;
      P.dbond = c->d_btrain ? c->d_bbond + (int64_t)i * (L + 1) : nullptr; P.dstride = c->bt_stride;
      P.marg = c->d_beliefs + (size_t)q * L * i; P.logz_out = c->d_logz_node + i; P.bmax = fr.gen ? cap : capct;
      rv_off.push_back(rv_doubles); rv_doubles += (size_t)(L + 1) * capct;
      eps.push_back(P);
    }
    if (capct > 256) return c->fail(MPBP_EUNSUPPORTED, "q*max_bond (q*q*max_bond for periodic chains) > 256 not supported by the scan kernels yet");
    auto run_env = [&](std::vector<EnvProb>& ev, const std::vector<size_t>& off, size_t rvd) -> int {
      if (ev.empty()) return MPBP_OK;
      int rc2 = ensure_arena(c, c->scratch, sizeof(EnvProb) * ev.size() + sizeof(double) * rvd + 8192);
      if (rc2 != MPBP_OK) return rc2;
      double* rvbase = (double*)(c->scratch.base + ((sizeof(EnvProb) * ev.size() + 255) & ~size_t(255)));
      for (size_t s2 = 0; s2 < ev.size(); s2++) ev[s2].rvec = rvbase + off[s2];
      HIPCHK(c, hipMemcpyAsync(c->scratch.base, ev.data(), sizeof(EnvProb) * ev.size(), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(env_kernel, dim3((unsigned)ev.size()), dim3(256), 0, c->stream, (const EnvProb*)c->scratch.base, L);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));
      return MPBP_OK;
    };
    rc = run_env(eps, rv_off, rv_doubles);
    if (rc != MPBP_OK) return rc;
    // ---- damping (reference src/recursive_bp_factor.jl:172-176): mu = compress!(new + damp/(1-damp) old), normalize!
    if (damp > 0.0) {
      int maxocc = 0;
      for (const FinRec& fr : fins) if (!fr.gen) maxocc = std::max(maxocc, fr.occ);
      for (int round = 0; round <= maxocc; round++) {      // aliased out-edges compound in the reference's loop order
        std::vector<ComposeProb> cps2; EngLaunchPlan pl2; std::vector<EnvProb> ev2; std::vector<size_t> off2; size_t rvd2 = 0;
        pl2.q = 1; pl2.capout = cap; pl2.cap1 = 2 * cap; pl2.cap2 = 1; pl2.ny1 = q * q; pl2.ny2 = 1; pl2.ny = q * q;
        pl2.small = small_problem(2 * cap, q * q, 1);
        for (const FinRec& fr : fins) {
          if (fr.occ != round || fr.gen) continue;
          const int eo = c->out_edge[fr.p];
          const DevTrain &nm = tr[fr.nrm], &sm = tr[fr.sum], &o2 = tr[fr.out2];
          ComposeProb CP{};
          CP.an = nm.cores; CP.bn = nm.bonds; CP.nstride = nm.stride;
          CP.ao = c->slot_cores(eo); CP.bo = c->slot_bonds(eo); CP.ostride = c->core_stride;
          CP.out = sm.cores; CP.obond = sm.bonds; CP.outstride = sm.stride; CP.ologz = sm.logz;
          CP.c = damp / (1.0 - damp); CP.p = q * q;
          cps2.push_back(CP);
          EngProb EP{};
          EP.A1 = sm.cores; EP.bond1 = sm.bonds; EP.stride1 = sm.stride; EP.ny1 = q * q;
          EP.A2 = c->d_one; EP.bond2 = c->d_ones; EP.stride2 = 0; EP.ny2 = 1;
          EP.logz1 = sm.logz; EP.logz2 = nullptr; EP.pyy = c->d_ident; EP.pyy_tstride = 0;
          EP.ny = q * q; EP.q = 1; EP.mirror = 0; EP.cap_out = cap;
          EP.out = o2.cores; EP.obond = o2.bonds; EP.ostride = o2.stride; EP.ologz = o2.logz;
          pl2.probs.push_back(EP); pl2.cost.push_back(1.0);
          EnvProb VP{};
          VP.in = o2.cores; VP.ibond = o2.bonds; VP.istride = o2.stride; VP.ilogz = o2.logz; VP.p = q * q;
          VP.dst = c->slot_cores(eo); VP.dbond = c->slot_bonds(eo); VP.dstride = c->core_stride;
          VP.marg = nullptr; VP.logz_out = nullptr; VP.bmax = cap;
          off2.push_back(rvd2); rvd2 += (size_t)(L + 1) * cap;
          ev2.push_back(VP);
        }
        if (cps2.empty()) continue;
        rc = ensure_arena(c, c->scratch, sizeof(ComposeProb) * cps2.size() + 4096);
        if (rc != MPBP_OK) return rc;
        HIPCHK(c, hipMemcpyAsync(c->scratch.base, cps2.data(), sizeof(ComposeProb) * cps2.size(), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(compose_kernel, dim3(L, (unsigned)cps2.size()), dim3(256), 0, c->stream, (const ComposeProb*)c->scratch.base, L);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        rc = launch_engine(c, pl2, trunc, false, &ms_orth, &n_orth);
        if (rc != MPBP_OK) return rc;
        rc = run_env(ev2, off2, rvd2);
        if (rc != MPBP_OK) return rc;
      }
    }
  }
  // ---------------------------------------------------------------- f[i] (src/recursive_bp_factor.jl:163)
  hipEventRecord(ev1, c->stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->h_logz_node.data(), c->d_logz_node, sizeof(double) * c->N, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(c->h_logz_pos.data(), c->d_logz_pos, sizeof(double) * c->nnz(), hipMemcpyDeviceToHost));
  for (int k = 0; k < n_nodes; k++) {
    const int i = nodes[k]; const int z = c->fac[i].deg;
    double s = 0.0;
    for (int p = c->nbr_ptr[i]; p < c->nbr_ptr[i + 1]; p++) s += c->h_logz_pos[p];
    c->h_f[i] = (z / 2.0 - 1.0) * c->h_logz_node[i] - 0.5 * s;
  }
  EngStats hs;
  HIPCHK(c, hipMemcpy(&hs, c->d_stats, sizeof hs, hipMemcpyDeviceToHost));
  float ms = 0; hipEventElapsedTime(&ms, ev0, ev1);
  mpbp_stats st{};
  { double v; unsigned long long b = hs.maxerr_bits; memcpy(&v, &b, 8); st.maxerr = v; }
  st.n_compress = (int64_t)hs.n_compress; st.nan_flag = hs.nan_flag; st.capacity_flag = hs.capacity_flag;
  st.jacobi_not_converged = hs.jacobi_fail; st.jacobi_sweeps = (int64_t)hs.jac_sweeps; st.jacobi_calls = (int64_t)hs.jac_calls; st.ms_total = ms; st.ms_orth = ms_orth; st.n_orth_launches = n_orth;
  c->last = st;
  if (stats) *stats = st;
  if (hs.capacity_flag) return c->fail(MPBP_ECAPACITY, "a truncated bond exceeded max_bond=%d; results were clamped", cap);
  return MPBP_OK;
}

static void merge_stats(mpbp_stats& a, const mpbp_stats& b) {
  a.maxerr = std::max(a.maxerr, b.maxerr); a.n_compress += b.n_compress;
  a.nan_flag |= b.nan_flag; a.capacity_flag |= b.capacity_flag; a.jacobi_not_converged |= b.jacobi_not_converged;
  a.ms_total += b.ms_total; a.ms_orth += b.ms_orth; a.n_orth_launches += b.n_orth_launches;
  a.jacobi_sweeps += b.jacobi_sweeps; a.jacobi_calls += b.jacobi_calls;
}

static int sweep_split(mpbp_ctx* c, const int32_t* nodes, int32_t n, mpbp_trunc trunc, double damp, mpbp_stats* acc) {
  mpbp_stats st{};
  int rc = sweep_nodes(c, nodes, n, trunc, damp, &st, true);
  if (rc == MPBP_ESPLIT_INTERNAL) {
    const int h = n / 2;
    rc = sweep_split(c, nodes, h, trunc, damp, acc);
    if (rc != MPBP_OK && rc != MPBP_ECAPACITY) return rc;
    const int rc2 = sweep_split(c, nodes + h, n - h, trunc, damp, acc);
    return rc2 != MPBP_OK ? rc2 : rc;
  }
  if (rc == MPBP_OK || rc == MPBP_ECAPACITY) merge_stats(*acc, st);
  return rc;
}

extern "C" int mpbp_sweep(mpbp_ctx* c, const int32_t* nodes, int32_t n_nodes, mpbp_trunc trunc, double damp, mpbp_stats* stats) {
  if (!c) return MPBP_EINVAL;
  if (n_nodes < 0 || (n_nodes > 0 && !nodes)) return c->fail(MPBP_EINVAL, "bad node list");
  if (!(damp >= 0.0 && damp < 1.0)) return c->fail(MPBP_EINVAL, "damp must satisfy 0 <= damp < 1 (reference src/recursive_bp_factor.jl:169)");
  if (trunc.kind < 0 || trunc.kind > 3) return c->fail(MPBP_EINVAL, "unknown truncation kind %d", trunc.kind);
  if (trunc.kind != MPBP_TRUNC_THRESH && trunc.mprime < 1) return c->fail(MPBP_EINVAL, "mprime must be >= 1");
  hipSetDevice(c->device);
  if (c->tables_dirty) { int rc = build_tables(c); if (rc != MPBP_OK) return rc; }
  std::vector<char> seen(c->N, 0);
  for (int k = 0; k < n_nodes; k++) {
    if (nodes[k] < 0 || nodes[k] >= c->N) return c->fail(MPBP_EINVAL, "node id %d out of range", nodes[k]);
    if (seen[nodes[k]]) return c->fail(MPBP_EINVAL, "node %d listed twice", nodes[k]);
    seen[nodes[k]] = 1;
  }
  c->snap_cores = nullptr; c->snap_bonds = nullptr;
  mpbp_stats st{};
  int rc = sweep_nodes(c, nodes, n_nodes, trunc, damp, &st, true);
  if (rc == MPBP_ESPLIT_INTERNAL) {
    // The listed nodes are updated from the messages at entry (Jacobi semantics of one call): the halves read a
    // snapshot, so that the split is invisible in the results.  Only the in-edges of the listed nodes are read by a pass
    // (prep), so only their slots are copied - the snapshot is allocated exactly when memory is short, and a copy of the
    // whole slab (24 GB at configs[2]) could be what does not fit.
    c->snap_index.assign(c->slot_of_edge.size(), -1);
    std::vector<int> ins;
    for (int k = 0; k < n_nodes; k++)
      for (int p = c->nbr_ptr[nodes[k]]; p < c->nbr_ptr[nodes[k] + 1]; p++) {
        const int e = c->in_edge[p];
        if (c->snap_index[e] < 0) { c->snap_index[e] = (int32_t)ins.size(); ins.push_back(e); }
      }
    const size_t nin = std::max<size_t>(ins.size(), 1);
    const size_t cb = sizeof(double) * (size_t)c->slot_doubles * nin, bb = sizeof(int32_t) * (size_t)(c->L + 1) * nin;
    void* snap = nullptr;
    hipError_t e = hipMalloc(&snap, cb + bb + 256);
    if (e != hipSuccess) return c->fail(MPBP_ENOMEM, "hipMalloc(%zu MiB message snapshot for a split sweep) failed", (cb + bb) >> 20);
    double* sc = (double*)snap; int32_t* sb = (int32_t*)((char*)snap + ((cb + 255) & ~size_t(255)));
    for (size_t k = 0; k < ins.size(); k++) {
      hipMemcpyAsync(sc + (int64_t)k * c->slot_doubles, c->slot_cores(ins[k]), sizeof(double) * c->slot_doubles, hipMemcpyDeviceToDevice, c->stream);
      hipMemcpyAsync(sb + (int64_t)k * (c->L + 1), c->slot_bonds(ins[k]), sizeof(int32_t) * (c->L + 1), hipMemcpyDeviceToDevice, c->stream);
    }
    c->snap_cores = sc; c->snap_bonds = sb;
    st = mpbp_stats{};
    rc = sweep_split(c, nodes, n_nodes, trunc, damp, &st);
    hipStreamSynchronize(c->stream);
    c->snap_cores = nullptr; c->snap_bonds = nullptr;
    hipFree(snap);
    c->last = st;
  }
  if (stats) *stats = c->last;
  return rc;
}

// Two-time marginals of the listed nodes' beliefs on the device (reference src/mpbp.jl:245-286 via TensorTrains
// `twovar_marginals`): out[k][t][u][x + q*y] = p_{nodes[k]}(x^t = x, x^u = y) for t < u <= t + maxdist, 0 elsewhere.
extern "C" int mpbp_twovar_marginals(mpbp_ctx* c, const int32_t* nodes, int32_t n_nodes, int32_t maxdist, double* out) {
  if (!c || !out || n_nodes < 0 || (n_nodes > 0 && !nodes)) return MPBP_EINVAL;
  if (!c->d_btrain) return c->fail(MPBP_EUNSUPPORTED, "belief trains are not kept on this context (not enough device memory)");
  if (n_nodes == 0) return MPBP_OK;
  hipSetDevice(c->device);
  const int L = c->L, q = c->q;
  if (maxdist < 1 || maxdist > L) maxdist = L;
  const int bmax = c->ct_factor() * c->cap;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<int32_t> b0(c->N);
  HIPCHK(c, hipMemcpy2D(b0.data(), sizeof(int32_t), c->d_bbond, sizeof(int32_t) * (L + 1), sizeof(int32_t), c->N, hipMemcpyDeviceToHost));
  const size_t env_d = (size_t)(L + 1) * bmax, out_d = (size_t)L * L * q * q;
  const size_t per = sizeof(double) * (2 * env_d + out_d);
  int rc = ensure_arena(c, c->scratch, per * n_nodes + sizeof(TvProb) * n_nodes + 8192);
  if (rc != MPBP_OK) return rc;
  std::vector<TvProb> tp(n_nodes);
  char* base = c->scratch.base + ((sizeof(TvProb) * n_nodes + 255) & ~size_t(255));
  for (int k = 0; k < n_nodes; k++) {
    const int i = nodes[k];
    if (i < 0 || i >= c->N) return c->fail(MPBP_EINVAL, "node %d out of range", i);
    if (b0[i] == 0) return c->fail(MPBP_EINVAL, "node %d has not been updated yet", i);
    double* d = (double*)(base + per * k);
    tp[k] = TvProb{c->d_btrain + (int64_t)i * c->bt_slot, c->d_bbond + (int64_t)i * (L + 1), c->bt_stride, d, d + env_d, d + 2 * env_d, q, bmax};
  }
  HIPCHK(c, hipMemcpyAsync(c->scratch.base, tp.data(), sizeof(TvProb) * n_nodes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(base, 0, per * n_nodes, c->stream));
  hipLaunchKernelGGL(tv_env_kernel, dim3(n_nodes), dim3(256), 0, c->stream, (const TvProb*)c->scratch.base, L);
  const size_t lds = sizeof(double) * (2 * (size_t)q * bmax + 64 + 8);
  HIPCHK(c, hipFuncSetAttribute((const void*)tv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(tv_kernel, dim3(L, n_nodes), dim3(256), lds, c->stream, (const TvProb*)c->scratch.base, L, maxdist);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < n_nodes; k++)
    HIPCHK(c, hipMemcpy(out + out_d * k, base + per * k + sizeof(double) * 2 * env_d, sizeof(double) * out_d, hipMemcpyDeviceToHost));
  return MPBP_OK;
}

// ================================================================================================
// the exchange step of the multi-GPU path, inside the boundary
// ================================================================================================
#include <dlfcn.h>
// One in-place all-gather of the rank-major message slab (+ one of the bond table) over RCCL on the context's
// stream, then a stream synchronise: makes the messages written by `mpbp_sweep` on every rank visible on all of
// them - the `bp.mu[idx(e)] = muj` of the reference (src/recursive_bp_factor.jl:177) across GPUs.
// `nccl_comm`: an ncclComm_t of the RCCL library this process uses (the symbol is looked up among the loaded
// libraries first, so a host that already talks to RCCL - torch.distributed, MPI.jl + RCCL - shares its copy).
extern "C" int mpbp_allgather_slots(mpbp_ctx* c, void* nccl_comm, int32_t rank, int32_t world, int32_t slots_per_rank) {
  if (!c) return MPBP_EINVAL;
  if (!nccl_comm || world < 1 || rank < 0 || rank >= world || slots_per_rank < 1 || (int64_t)world * slots_per_rank != c->nslots)
    return c->fail(MPBP_EINVAL, "mpbp_allgather_slots: need world * slots_per_rank == n_slots (%d) and 0 <= rank < world", c->nslots);
  typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
  static allgather_fn fn = nullptr;
  if (!fn) {
    fn = (allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (!fn) {
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) fn = (allgather_fn)dlsym(h, "ncclAllGather");
    }
    if (!fn) return c->fail(MPBP_EUNSUPPORTED, "ncclAllGather not found: load RCCL (librccl.so) into the process first");
  }
  hipSetDevice(c->device);
  const size_t cb = sizeof(double) * (size_t)c->slot_doubles * slots_per_rank, bb = sizeof(int32_t) * (size_t)(c->L + 1) * slots_per_rank;
  int rc = fn((const char*)c->d_cores + cb * rank, c->d_cores, cb, /*ncclInt8*/ 0, nccl_comm, c->stream);
  if (rc == 0) rc = fn((const char*)c->d_bonds + bb * rank, c->d_bonds, bb, 0, nccl_comm, c->stream);
  if (rc != 0) return c->fail(MPBP_EHIP, "ncclAllGather failed with ncclResult_t %d", rc);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MPBP_OK;
}

// ================================================================================================
// observables
// ================================================================================================
extern "C" int mpbp_beliefs(mpbp_ctx* c, double* out) {
  if (!c || !out) return MPBP_EINVAL;
  hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, c->d_beliefs, sizeof(double) * c->q * c->L * c->N, hipMemcpyDeviceToHost));
  return MPBP_OK;
}
extern "C" int mpbp_free_energy(mpbp_ctx* c, double* f) {
  if (!c || !f) return MPBP_EINVAL;
  memcpy(f, c->h_f.data(), sizeof(double) * c->N);
  return MPBP_OK;
}
extern "C" int mpbp_logz(mpbp_ctx* c, double* ln, double* lm) {
  if (!c) return MPBP_EINVAL;
  if (ln) memcpy(ln, c->h_logz_node.data(), sizeof(double) * c->N);
  if (lm) {
    // per edge: value of the last neighbour position writing that edge
    for (int e = 0; e < c->E; e++) lm[e] = 0.0;
    for (int p = 0; p < c->nnz(); p++) lm[c->out_edge[p]] = c->h_logz_pos[p];
  }
  return MPBP_OK;
}

extern "C" int mpbp_get_belief_train(mpbp_ctx* c, int32_t node, int32_t* bonds, double* data, int64_t data_capacity) {
  if (!c || !bonds) return MPBP_EINVAL;
  if (node < 0 || node >= c->N) return c->fail(MPBP_EINVAL, "node %d out of range", node);
  if (!c->d_btrain) return c->fail(MPBP_EUNSUPPORTED, "belief trains are not kept on this context (not enough device memory)");
  hipSetDevice(c->device);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int L = c->L;
  HIPCHK(c, hipMemcpy(bonds, c->d_bbond + (int64_t)node * (L + 1), sizeof(int32_t) * (L + 1), hipMemcpyDeviceToHost));
  if (bonds[0] == 0) return c->fail(MPBP_EINVAL, "node %d has not been updated yet", node);
  if (!data) return MPBP_OK;                      // size query
  int64_t need = 0;
  for (int t = 0; t < L; t++) need += (int64_t)bonds[t] * bonds[t + 1] * c->q;
  if (need > data_capacity) return c->fail(MPBP_EINVAL, "buffer too small: need %lld doubles", (long long)need);
  std::vector<double> slot((size_t)c->bt_slot);
  HIPCHK(c, hipMemcpy(slot.data(), c->d_btrain + (int64_t)node * c->bt_slot, sizeof(double) * c->bt_slot, hipMemcpyDeviceToHost));
  double* dst = data;
  for (int t = 0; t < L; t++) {
    const int64_t n = (int64_t)bonds[t] * bonds[t + 1] * c->q;
    memcpy(dst, slot.data() + (int64_t)t * c->bt_stride, sizeof(double) * n);
    dst += n;
  }
  return MPBP_OK;
}

extern "C" int mpbp_pair_beliefs(mpbp_ctx* c, double* out, double* logz_pair) {
  if (!c || !out) return MPBP_EINVAL;
  hipSetDevice(c->device);
  const int L = c->L, q = c->q, cap = c->cap, E = c->E;
  // reverse edge: edge e = out_edge[p] of node i at position p pairs with in_edge[p] (the message k->i)
  std::vector<int> rev(E, -1);
  for (int p = 0; p < c->nnz(); p++) rev[c->out_edge[p]] = c->in_edge[p];
  for (int e = 0; e < E; e++) if (rev[e] < 0) return c->fail(MPBP_EINVAL, "edge %d is nobody's out-edge", e);
  const size_t per = (size_t)(L + 1) * cap * cap + 3 * (size_t)cap * cap;
  const size_t outd = (size_t)q * q * L * E;
  int rc = ensure_arena(c, c->scratch, sizeof(PairProb) * E + sizeof(double) * (per * E + outd + E) + sizeof(double) * q * q * L * E + 16384);
  if (rc != MPBP_OK) return rc;
  char* base = c->scratch.base;
  PairProb* d_p = (PairProb*)base; base += (sizeof(PairProb) * E + 255) & ~size_t(255);
  double* d_out = (double*)base; base += sizeof(double) * outd;
  double* d_lz = (double*)base; base += (sizeof(double) * E + 255) & ~size_t(255);
  double* d_psi = (double*)base; base += sizeof(double) * q * q * L * E;
  double* d_scr = (double*)base;
  HIPCHK(c, hipMemcpyAsync(d_psi, c->psi.data(), sizeof(double) * q * q * L * E, hipMemcpyHostToDevice, c->stream));
  std::vector<PairProb> pp(E);
  for (int e = 0; e < E; e++) {
    PairProb P{};
    P.aij = c->slot_cores(e); P.bij = c->slot_bonds(e); P.aji = c->slot_cores(rev[e]); P.bji = c->slot_bonds(rev[e]);
    P.stride = c->core_stride; P.psi = d_psi + (size_t)q * q * L * e; P.out = d_out + (size_t)q * q * L * e; P.logz = d_lz + e;
    P.scratch = d_scr + per * e; P.q = q; P.cap = cap;
    pp[e] = P;
  }
  HIPCHK(c, hipMemcpyAsync(d_p, pp.data(), sizeof(PairProb) * E, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(pair_kernel, dim3(E), dim3(256), 0, c->stream, (const PairProb*)d_p, L);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out, d_out, sizeof(double) * outd, hipMemcpyDeviceToHost));
  if (logz_pair) HIPCHK(c, hipMemcpy(logz_pair, d_lz, sizeof(double) * E, hipMemcpyDeviceToHost));
  return MPBP_OK;
}

// ================================================================================================
// self tests of the device building blocks
// ================================================================================================
static int st_fail(const char* what, hipError_t e) { g_create_error = std::string(what) + ": " + hipGetErrorString(e); return MPBP_EHIP; }
#define STCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return st_fail(#call, e_); } while (0)

extern "C" int mpbp_selftest_gemm(int32_t device, int32_t M, int32_t N, int32_t K, const double* A, const double* B, double* C) {
  STCHK(hipSetDevice(device));
  double *dA, *dB, *dC;
  STCHK(hipMalloc(&dA, sizeof(double) * M * K)); STCHK(hipMalloc(&dB, sizeof(double) * K * N)); STCHK(hipMalloc(&dC, sizeof(double) * M * N));
  STCHK(hipMemcpy(dA, A, sizeof(double) * M * K, hipMemcpyHostToDevice));
  STCHK(hipMemcpy(dB, B, sizeof(double) * K * N, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(st_gemm_kernel, dim3(1), dim3(WG_THREADS), wg::GM_LDS_DOUBLES * 8, 0, M, N, K, dA, dB, dC);
  STCHK(hipGetLastError()); STCHK(hipDeviceSynchronize());
  STCHK(hipMemcpy(C, dC, sizeof(double) * M * N, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dB); hipFree(dC);
  return MPBP_OK;
}

extern "C" int mpbp_selftest_qr(int32_t device, int32_t rows, int32_t cols, const double* A, double* R) {
  STCHK(hipSetDevice(device));
  const int ld = (rows + 31) & ~31, c16 = ((cols + 15) & ~15) + 16;
  std::vector<double> Y((size_t)ld * c16, 0.0);
  for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) Y[i + (size_t)ld * j] = A[i + (size_t)rows * j];
  double* dY;
  STCHK(hipMalloc(&dY, sizeof(double) * Y.size()));
  STCHK(hipMemcpy(dY, Y.data(), sizeof(double) * Y.size(), hipMemcpyHostToDevice));
  STCHK(hipFuncSetAttribute((const void*)st_qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (wg::QR_LDS_DOUBLES + wg::QR_BIG_DOUBLES) * 8));
  hipLaunchKernelGGL(st_qr_kernel, dim3(1), dim3(WG_THREADS), (wg::QR_LDS_DOUBLES + wg::QR_BIG_DOUBLES) * 8, 0, dY, ld, rows, cols, (wg::Prof*)nullptr);
  STCHK(hipGetLastError()); STCHK(hipDeviceSynchronize());
  STCHK(hipMemcpy(Y.data(), dY, sizeof(double) * Y.size(), hipMemcpyDeviceToHost));
  const int k = std::min(rows, cols);
  for (int j = 0; j < cols; j++) for (int i = 0; i < k; i++) R[i + (size_t)k * j] = (j >= i) ? Y[i + (size_t)ld * j] : 0.0;
  hipFree(dY);
  return MPBP_OK;
}

// times `reps` launches of nblocks concurrent QRs (each block its own random matrix); returns ms per launch
extern "C" int mpbp_selftest_qr_bench(int32_t device, int32_t rows, int32_t cols, int32_t nblocks, int32_t reps, double* ms_out) {
  STCHK(hipSetDevice(device));
  const int ld = (rows + 31) & ~31, c16 = ((cols + 15) & ~15) + 16;
  const size_t per = (size_t)ld * c16;
  std::vector<double> Y(per, 0.0);
  unsigned long long st = 88172645463325252ULL;
  for (int j = 0; j < cols; j++) for (int i = 0; i < rows; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; Y[i + (size_t)ld * j] = (double)(st % 2000001) / 1e6 - 1.0; }
  double* dY; double* dY0;
  STCHK(hipMalloc(&dY, sizeof(double) * per * nblocks)); STCHK(hipMalloc(&dY0, sizeof(double) * per));
  STCHK(hipMemcpy(dY0, Y.data(), sizeof(double) * per, hipMemcpyHostToDevice));
  STCHK(hipFuncSetAttribute((const void*)st_qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (wg::QR_LDS_DOUBLES + wg::QR_BIG_DOUBLES) * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0.f;
  wg::Prof* dprof = nullptr;                 // MPBP_QR_PROF=1: per-phase split of the factorisation on stderr
  if (getenv("MPBP_QR_PROF")) { STCHK(hipMalloc(&dprof, sizeof(wg::Prof))); STCHK(hipMemset(dprof, 0, sizeof(wg::Prof))); }
  for (int r = 0; r < reps + 1; r++) {
    if (dprof && r == 1) STCHK(hipMemset(dprof, 0, sizeof(wg::Prof)));
    for (int b = 0; b < nblocks; b++) STCHK(hipMemcpyAsync(dY + per * b, dY0, sizeof(double) * per, hipMemcpyDeviceToDevice, 0));
    STCHK(hipDeviceSynchronize());
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(st_qr_kernel, dim3(nblocks), dim3(WG_THREADS), (wg::QR_LDS_DOUBLES + wg::QR_BIG_DOUBLES) * 8, 0, dY, ld, rows, cols, dprof);
    hipEventRecord(e1, 0);
    STCHK(hipEventSynchronize(e1));
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0) tot += ms;
  }
  *ms_out = tot / reps;
  if (dprof) {
    wg::Prof hp; STCHK(hipMemcpy(&hp, dprof, sizeof(hp), hipMemcpyDeviceToHost));
    const char* nm[5] = {"panel_regs", "gram", "T_from_gram", "tile_update", "trail"};
    for (int k = 0; k < 5; k++) fprintf(stderr, "  qr phase %-12s %8.3f ms per workgroup per launch\n", nm[k], hp.t[14 + k] * 1e-5 / nblocks / reps);
    hipFree(dprof);
  }
  hipFree(dY); hipFree(dY0); hipEventDestroy(e0); hipEventDestroy(e1);
  return MPBP_OK;
}

// times `reps` launches of nblocks concurrent LDS-resident Jacobi SVDs (m x n); variant 0 = 512 threads, 1 = one wave
extern "C" int mpbp_selftest_jacobi_bench(int32_t device, int32_t m, int32_t n, int32_t nblocks, int32_t variant,
                                          int32_t reps, double* ms_out, double* avg_sweeps) {
  STCHK(hipSetDevice(device));
  int* dS; STCHK(hipMalloc(&dS, sizeof(int) * nblocks));
  const size_t lds = sizeof(double) * (64 + n + (size_t)(m | 1) * n + 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0.f;
  for (int r = 0; r < reps + 1; r++) {
    hipEventRecord(e0, 0);
    if (variant == 1) hipLaunchKernelGGL(v64::jac_bench_kernel, dim3(nblocks), dim3(64), lds, 0, m, n, dS);
    else hipLaunchKernelGGL(v512::jac_bench_kernel, dim3(nblocks), dim3(512), lds, 0, m, n, dS);
    hipEventRecord(e1, 0);
    STCHK(hipGetLastError()); STCHK(hipEventSynchronize(e1));
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0) tot += ms;
  }
  std::vector<int> hs(nblocks);
  STCHK(hipMemcpy(hs.data(), dS, sizeof(int) * nblocks, hipMemcpyDeviceToHost));
  double acc = 0; for (int v : hs) acc += v;
  *ms_out = tot / reps; *avg_sweeps = acc / nblocks;
  hipFree(dS); hipEventDestroy(e0); hipEventDestroy(e1);
  return MPBP_OK;
}

extern "C" int mpbp_selftest_svd(int32_t device, int32_t rows, int32_t cols, const double* A, double* sigma, double* V) {
  STCHK(hipSetDevice(device));
  double *dA, *dV, *dS; int* dW;
  STCHK(hipMalloc(&dA, sizeof(double) * rows * cols)); STCHK(hipMalloc(&dV, sizeof(double) * cols * cols));
  STCHK(hipMalloc(&dS, sizeof(double) * cols)); STCHK(hipMalloc(&dW, sizeof(int)));
  STCHK(hipMemcpy(dA, A, sizeof(double) * rows * cols, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(st_svd_kernel, dim3(1), dim3(WG_THREADS), (64 + cols) * 8, 0, dA, rows, cols, dV, dS, dW);
  STCHK(hipGetLastError()); STCHK(hipDeviceSynchronize());
  STCHK(hipMemcpy(sigma, dS, sizeof(double) * cols, hipMemcpyDeviceToHost));
  STCHK(hipMemcpy(V, dV, sizeof(double) * cols * cols, hipMemcpyDeviceToHost));
  int sw = 0; STCHK(hipMemcpy(&sw, dW, sizeof(int), hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dV); hipFree(dS); hipFree(dW);
  return sw < 0 ? MPBP_EUNSUPPORTED : MPBP_OK;
}
