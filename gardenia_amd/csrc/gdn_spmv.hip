// gdn_spmv.hip -- CSR SpMV, fp32, merge-based.
//
// Reference path: SpmvSolver (src/spmv/spmv.h:29); OpenMP src/spmv/omp_base.cc:7-42; CUDA
// src/spmv/base.cu:13 (thread per row), warp.cu:26 (warp per row), vector.cu:27 (sub-warp
// per row chosen by nnz/m, x through a texture), push.cu:11 (scatter with atomics).  One
// merge-path pass (gdn_mergepath.hpp) replaces all four: y[i] += sum_k Ax[k]*x[Aj[k]] over
// the rows of (Ap, Aj); products are rounded before the add like the reference's x86 build.
#include <string.h>

#include "gdn_mergepath.hpp"

struct gdn_spmv_plan {
  MpPlan mp;
};

struct SpmvOp {
  const float *__restrict__ Ax;
  const float *__restrict__ x;
  float *__restrict__ y;
  __device__ __forceinline__ float load(uint64_t j, vid_t col) const {
    return __fmul_rn(x[col], __builtin_nontemporal_load(Ax + j));
  }
  __device__ __forceinline__ double finish(int32_t row, float sum) const {
    y[row] = __fadd_rn(y[row], sum);
    return 0.0;
  }
};

extern "C" {

int gdn_spmv_plan_create(const gdn_graph *csr, gdn_spmv_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(csr != nullptr, "csr");
  gdn_spmv_plan *p = new gdn_spmv_plan();
  int st = mp_plan_build(p->mp, csr, 0);
  if (st == GDN_OK && hipDeviceSynchronize() != hipSuccess) {
    gdn_set_error("gdn_spmv_plan_create: tile table kernel failed");
    st = GDN_ERR_HIP;
  }
  if (st != GDN_OK) {
    delete p;
    return st;
  }
  *plan = p;
  return GDN_OK;
}

int gdn_spmv_plan_free(gdn_spmv_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_spmv_dev(gdn_spmv_plan *plan, const float *d_Ax, const float *d_x, float *d_y, void *stream) {
  GDN_REQUIRE(plan && d_Ax && d_x && d_y, "null argument");
  SpmvOp op;
  op.Ax = d_Ax;
  op.x = d_x;
  op.y = d_y;
  return mp_run(plan->mp, op, nullptr, (hipStream_t)stream);
}

int gdn_spmv_plan_kernel_time(gdn_spmv_plan *plan, int32_t reset, int32_t max_launches, double *total_ms,
                              int32_t *launches) {
  GDN_REQUIRE(plan != nullptr, "plan");
  if (total_ms) total_ms[0] = total_ms[1] = 0.0;
  return mp_plan_timing(plan->mp, reset, max_launches, total_ms, launches);
}

// 8(m+1) + 4 nnz [Aj] + 4 nnz [Ax] + 4 nnz [x gather] + 8 m [y r+w]
uint64_t gdn_spmv_bytes(const gdn_spmv_plan *plan) {
  if (!plan) return 0;
  const uint64_t m = (uint64_t)plan->mp.m, nnz = plan->mp.nnz;
  return 8 * (m + 1) + 12 * nnz + 8 * m;
}

// Host API: one call == SpmvSolver(g, Ax, x, y) (src/spmv/main.cc:39).
int gdn_spmv(int32_t m, uint64_t nnz, const uint64_t *Ap, const int32_t *Aj, const float *Ax,
             const float *x, float *y, gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && Ap && x && y && (Ax || nnz == 0), "null argument");
  GDN_TRY(gdn_require_device());
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer th2d, tprep, tsolve;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, Ap, Aj, &g));
  DevBuf<float> d_Ax, d_x, d_y;
  gdn_spmv_plan *plan = nullptr;
  int rc = GDN_OK;
  do {
    if ((rc = d_Ax.alloc(nnz)) || (rc = d_x.alloc(m)) || (rc = d_y.alloc(m))) break;
    if ((nnz && hipMemcpy(d_Ax.p, Ax, nnz * 4, hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(d_x.p, x, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_y.p, y, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) {
      gdn_set_error("gdn_spmv: upload failed");
      rc = GDN_ERR_HIP;
      break;
    }
    st.h2d_ms = th2d.stop_ms();
    tprep.start();
    if ((rc = gdn_spmv_plan_create(g, &plan))) break;
    st.prep_ms = tprep.stop_ms();
    tsolve.start();  // src/spmv/warp.cu:100-104: one timed launch
    if ((rc = gdn_spmv_dev(plan, d_Ax.p, d_x.p, d_y.p, nullptr))) break;
    st.solve_ms = tsolve.stop_ms();
    st.iterations = 1;
    st.edges_traversed = nnz;
    if (hipMemcpy(y, d_y.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_spmv: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  gdn_spmv_plan_free(plan);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
