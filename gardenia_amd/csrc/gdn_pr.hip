// gdn_pr.hip -- PageRank, pull direction, one fused pass per iteration.
//
// Reference path: PRSolver (src/pr/pr.h:31); OpenMP src/pr/omp_base.cc:8-42; CUDA
// src/pr/base.cu:77-135 = three launches per iteration (contrib :14, pull_step :19,
// l1norm :37) + a blocking 4-byte D2H of `diff` (:124).  Here one merge-path pass
// (gdn_mergepath.hpp) does gather + score update + L1 norm + next iteration's contrib;
// contrib is double-buffered so the iteration stays Jacobi like omp_base.cc:23-33.
// Arithmetic follows the reference in fp32: base_score = (1-d)/m, contrib = score/out_degree,
// new = base + d*sum (no FMA contraction), error accumulated in double (omp_base.cc:22).
#include <string.h>

#include "gdn_mergepath.hpp"

struct gdn_pr_plan {
  MpPlan mp;
  const int32_t *out_degree = nullptr;  // device, m_local
  int32_t m_global = 0;
  int32_t row_base = 0;
};

struct PrOp {
  const float *__restrict__ contrib_in;
  float *__restrict__ scores;
  float *__restrict__ contrib_out;  // already offset by row_base
  const int32_t *__restrict__ out_degree;
  float base_score;
  float damping;
  __device__ __forceinline__ float load(uint64_t, vid_t col) const { return contrib_in[col]; }
  __device__ __forceinline__ double finish(int32_t row, float sum) const {
    const float old_score = scores[row];
    const float new_score = __fadd_rn(base_score, __fmul_rn(damping, sum));
    scores[row] = new_score;
    contrib_out[row] = __fdiv_rn(new_score, (float)out_degree[row]);
    return (double)fabsf(__fsub_rn(new_score, old_score));
  }
};

__global__ void __launch_bounds__(GDN_BLOCK)
pr_contrib_kernel(const float *__restrict__ scores, const int32_t *__restrict__ out_degree, int32_t m,
                  float *__restrict__ contrib) {
  const int32_t v = blockIdx.x * GDN_BLOCK + threadIdx.x;
  if (v < m) contrib[v] = __fdiv_rn(scores[v], (float)out_degree[v]);
}

extern "C" {

int gdn_pr_plan_create(const gdn_graph *in_csr, const int32_t *d_out_degree, int32_t m_global,
                       int32_t row_base, gdn_pr_plan **plan) {
  GDN_REQUIRE(plan != nullptr, "plan");
  *plan = nullptr;
  GDN_REQUIRE(in_csr != nullptr && d_out_degree != nullptr, "in_csr / d_out_degree");
  GDN_REQUIRE(m_global >= in_csr->m && row_base >= 0 && row_base + in_csr->m <= m_global, "row range");
  gdn_pr_plan *p = new gdn_pr_plan();
  p->out_degree = d_out_degree;
  p->m_global = m_global;
  p->row_base = row_base;
  int st = mp_plan_build(p->mp, in_csr, 0);
  if (st == GDN_OK && hipDeviceSynchronize() != hipSuccess) {
    gdn_set_error("gdn_pr_plan_create: tile table kernel failed");
    st = GDN_ERR_HIP;
  }
  if (st != GDN_OK) {
    delete p;
    return st;
  }
  *plan = p;
  return GDN_OK;
}

int gdn_pr_plan_free(gdn_pr_plan *plan) {
  delete plan;
  return GDN_OK;
}

int gdn_pr_contrib_dev(gdn_pr_plan *plan, const float *d_scores, float *d_contrib, void *stream) {
  GDN_REQUIRE(plan && d_scores && d_contrib, "null argument");
  hipLaunchKernelGGL(pr_contrib_kernel, dim3(gdn_nblocks((uint64_t)plan->mp.m)), dim3(GDN_BLOCK), 0,
                     (hipStream_t)stream, d_scores, plan->out_degree, plan->mp.m, d_contrib + plan->row_base);
  GDN_HIP(hipGetLastError());
  return GDN_OK;
}

int gdn_pr_pull_dev(gdn_pr_plan *plan, const float *d_contrib_in, float *d_scores, float *d_contrib_out,
                    double *d_diff, float damping, void *stream) {
  GDN_REQUIRE(plan && d_contrib_in && d_scores && d_contrib_out, "null argument");
  GDN_REQUIRE(d_contrib_in != d_contrib_out, "contrib_in and contrib_out must differ (Jacobi)");
  PrOp op;
  op.contrib_in = d_contrib_in;
  op.scores = d_scores;
  op.contrib_out = d_contrib_out + plan->row_base;
  op.out_degree = plan->out_degree;
  op.base_score = (1.0f - damping) / (float)plan->m_global;
  op.damping = damping;
  return mp_run(plan->mp, op, d_diff, (hipStream_t)stream);
}

int gdn_pr_plan_kernel_time(gdn_pr_plan *plan, int32_t reset, int32_t max_launches, double *total_ms,
                            int32_t *launches) {
  GDN_REQUIRE(plan != nullptr, "plan");
  return mp_plan_timing(plan->mp, reset, max_launches, total_ms, launches);
}

uint64_t gdn_pr_iter_bytes(const gdn_pr_plan *plan) {
  if (!plan) return 0;
  const uint64_t m = (uint64_t)plan->mp.m, nnz = plan->mp.nnz;
  return 8 * (m + 1) + 4 * nnz + 4 * nnz + 16 * m;
}

// Host API: one call == PRSolver(g, scores) (src/pr/main.cc:19).
int gdn_pr(int32_t m, uint64_t nnz, const uint64_t *in_rowptr, const int32_t *in_colidx,
           const int32_t *out_degree, float *scores, float damping, double epsilon, int32_t max_iter,
           gdn_stats *stats) {
  GDN_REQUIRE(m > 0 && in_rowptr && out_degree && scores, "null argument");
  GDN_REQUIRE(max_iter >= 1, "max_iter");
  GDN_TRY(gdn_require_device());
  gdn_stats st;
  memset(&st, 0, sizeof(st));
  HostTimer th2d, tprep, tsolve;
  th2d.start();
  gdn_graph *g = nullptr;
  GDN_TRY(gdn_graph_upload(m, nnz, in_rowptr, in_colidx, &g));
  DevBuf<int32_t> d_deg;
  DevBuf<float> d_scores, d_c0, d_c1;
  DevBuf<double> d_diff;
  int rc = GDN_OK;
  gdn_pr_plan *plan = nullptr;
  do {
    if ((rc = d_deg.alloc(m)) || (rc = d_scores.alloc(m)) || (rc = d_c0.alloc(m)) || (rc = d_c1.alloc(m)) ||
        (rc = d_diff.alloc(1)))
      break;
    if (hipMemcpy(d_deg.p, out_degree, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_scores.p, scores, (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess) {
      gdn_set_error("gdn_pr: upload failed");
      rc = GDN_ERR_HIP;
      break;
    }
    st.h2d_ms = th2d.stop_ms();
    tprep.start();
    if ((rc = gdn_pr_plan_create(g, d_deg.p, m, 0, &plan))) break;
    st.prep_ms = tprep.stop_ms();
    // timed region == src/pr/base.cu:110-128 (t.Start .. t.Stop around the do/while)
    tsolve.start();
    if ((rc = gdn_pr_contrib_dev(plan, d_scores.p, d_c0.p, nullptr))) break;
    float *cin = d_c0.p, *cout = d_c1.p;
    int iter = 0;
    double diff = 0;
    for (iter = 0; iter < max_iter; iter++) {
      if ((rc = gdn_pr_pull_dev(plan, cin, d_scores.p, cout, d_diff.p, damping, nullptr))) break;
      if (hipMemcpy(&diff, d_diff.p, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
        gdn_set_error("gdn_pr: diff readback failed: %s", hipGetErrorString(hipGetLastError()));
        rc = GDN_ERR_HIP;
        break;
      }
      float *tmp = cin;
      cin = cout;
      cout = tmp;
      if (diff < epsilon) break;  // omp_base.cc:36
    }
    if (rc) break;
    st.solve_ms = tsolve.stop_ms();
    st.iterations = iter + 1;  // the reference prints iter+1 (omp_base.cc:39)
    st.last_error = diff;
    st.edges_traversed = nnz * (uint64_t)(iter < max_iter ? iter + 1 : max_iter);
    if (hipMemcpy(scores, d_scores.p, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      gdn_set_error("gdn_pr: download failed");
      rc = GDN_ERR_HIP;
    }
  } while (0);
  gdn_pr_plan_free(plan);
  gdn_graph_free(g);
  if (stats) *stats = st;
  return rc;
}

}  // extern "C"
