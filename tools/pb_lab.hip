// pb_lab.hip -- measurement tool (not part of the product): variants of the two propagation-blocking
// kernels of gdn_pb.hpp on the REAL RMAT layout built by pb_build, one variant per line.  Timing-only
// variants produce wrong results on purpose; the product kernels are the ones in gdn_pb.hpp.
//   usage: pb_lab [scale=27] [log_chunk=15] [log_bin=14]
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../gardenia_amd/csrc/gdn_pb.hpp"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);      \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)
#define GK(x)                                                      \
  do {                                                             \
    int s = (x);                                                   \
    if (s != GDN_OK) {                                             \
      printf("gdn error %d (%s) at line %d\n", s, gdn_last_error(), __LINE__); \
      exit(1);                                                     \
    }                                                              \
  } while (0)

struct LabOp {  // PageRank epilogue, scalar + 16-byte forms (same arithmetic as PrOp)
  float *scores;
  float *contrib_out;
  const int32_t *deg;
  float base_score, damping;
  bool vec_ok;
  __device__ __forceinline__ unsigned long long to_fixed(float v, unsigned &bad) const { return pb_to_fixed(v, bad); }
  __device__ __forceinline__ float from_fixed(unsigned long long a, unsigned &bad) const {
    if (a >> 63) bad = 1u;
    return ldexpf((float)a, -PB_FIX_SHIFT);
  }
  struct Pre {
    float old_score;
    int32_t deg;
  };
  __device__ __forceinline__ Pre pre(int32_t row) const { return Pre{scores[row], deg[row]}; }
  __device__ __forceinline__ double fin(int32_t row, float sum, const Pre &p) const {
    const float ns = __fadd_rn(base_score, __fmul_rn(damping, sum));
    scores[row] = ns;
    contrib_out[row] = __fdiv_rn(ns, (float)p.deg);
    return (double)fabsf(__fsub_rn(ns, p.old_score));
  }
  struct Pre4 {
    pb_f32x4 old_score;
    pb_i32x4 deg;
  };
  __device__ __forceinline__ Pre4 pre4(int32_t row) const {
    return Pre4{*reinterpret_cast<const pb_f32x4 *>(scores + row), *reinterpret_cast<const pb_i32x4 *>(deg + row)};
  }
  __device__ __forceinline__ double fin4(int32_t row, const float (&sum)[4], const Pre4 &p) const {
    pb_f32x4 ns, nc;
    double d = 0.0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float v = __fadd_rn(base_score, __fmul_rn(damping, sum[c]));
      ns[c] = v;
      nc[c] = __fdiv_rn(v, (float)p.deg[c]);
      d += (double)fabsf(__fsub_rn(v, p.old_score[c]));
    }
    *reinterpret_cast<pb_f32x4 *>(scores + row) = ns;
    *reinterpret_cast<pb_f32x4 *>(contrib_out + row) = nc;
    return d;
  }
};

// ---- phase A variants.  FLAGS bit0: skip the slice load (timing), bit1: plain U/G loads, bit2: no G (sequential
// stores), bit3: UNR 4 instead of 8, bit4: no stores (read-only), bit5: no U/G loads (write-only)
template <int FLAGS, int THREADS>
__global__ void __launch_bounds__(THREADS)
labA(const float *__restrict__ x, int32_t m_global, int log_chunk, const eoff_t *__restrict__ chunk_ptr,
     const uint32_t *__restrict__ order, const uint16_t *__restrict__ U, const uint32_t *__restrict__ G,
     float *__restrict__ vals, const uint32_t *__restrict__ src_bits, const uint32_t *__restrict__ chunk_lo) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];
  __shared__ unsigned s_bits[PB_THREADS], s_pref[PB_THREADS], s_scr[PB_WAVES + 1];
  const unsigned ch = 1u << log_chunk;
  const unsigned c = order ? order[blockIdx.x] : blockIdx.x;
  if (!(FLAGS & 1)) {
    if (THREADS == PB_THREADS) pb_load_slice4(x, m_global, src_bits, chunk_lo[c], chunk_lo[c + 1], s_x, s_bits, s_pref, s_scr);
  }
  if (threadIdx.x == 0) s_x[ch] = 0.0f;
  __syncthreads();
  const eoff_t h0 = chunk_ptr[c] >> 2, h1 = chunk_ptr[c + 1] >> 2;
  const pb_u16x4 *U4 = reinterpret_cast<const pb_u16x4 *>(U);
  pb_f32x4 *X4 = reinterpret_cast<pb_f32x4 *>(vals);
  constexpr int UNR = (FLAGS & 8) ? 4 : 8;
  float sink = 0.f;
  for (eoff_t h = h0 + threadIdx.x; h < h1; h += UNR * THREADS) {
    pb_u16x4 u[UNR];
    unsigned d[UNR];
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t hh = h + (eoff_t)r * THREADS;
      if (hh < h1) {
        if (FLAGS & 32) {
          u[r] = pb_u16x4{(unsigned short)(hh & 32767), (unsigned short)((hh * 3) & 32767), (unsigned short)((hh * 5) & 32767), (unsigned short)((hh * 7) & 32767)};
          d[r] = (unsigned)(hh >> 1);
        } else {
          u[r] = (FLAGS & 2) ? U4[hh] : __builtin_nontemporal_load(U4 + hh);
          d[r] = (FLAGS & 4) ? (unsigned)(hh >> 1) : ((FLAGS & 2) ? G[hh >> 1] : __builtin_nontemporal_load(G + (hh >> 1)));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t hh = h + (eoff_t)r * THREADS;
      if (hh < h1) {
        pb_f32x4 o;
        o.x = s_x[u[r].x];
        o.y = s_x[u[r].y];
        o.z = s_x[u[r].z];
        o.w = s_x[u[r].w];
        if (FLAGS & 16) sink += o.x + o.y + o.z + o.w + (float)d[r];
        else X4[2 * (size_t)d[r] + (size_t)(hh & 1)] = o;
      }
    }
  }
  if ((FLAGS & 16) && sink == 1.2345f) vals[0] = sink;
}

// ---- phase B variants.  FLAGS bit0: no atomics, bit1: no epilogue, bit2: simple (non-pipelined) loop,
// bit3: scalar epilogue, bit4: plain (temporal) loads
template <int FLAGS, int THREADS, class Op>
__global__ void __launch_bounds__(THREADS)
labB(int32_t m_local, int log_bin, const eoff_t *__restrict__ bin_ptr, const uint32_t *__restrict__ order,
     const uint16_t *__restrict__ V, const float *__restrict__ vals, double *__restrict__ partial,
     unsigned *__restrict__ errflag, const uint32_t *__restrict__ dst_bits, const uint32_t *__restrict__ bin_lo, Op op) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];
  __shared__ unsigned s_bits[PB_THREADS], s_pref[PB_THREADS], s_scr[PB_WAVES + 1];
  const unsigned bn = 1u << log_bin;
  const unsigned b = order ? order[blockIdx.x] : blockIdx.x;
  for (unsigned i = threadIdx.x; i < bn; i += THREADS) s_acc[i] = 0ull;
  __syncthreads();
  const eoff_t q0 = bin_ptr[b] >> 2, q1 = bin_ptr[b + 1] >> 2;
  const pb_f32x4 *X4 = reinterpret_cast<const pb_f32x4 *>(vals);
  const pb_u16x4 *V4 = reinterpret_cast<const pb_u16x4 *>(V);
  unsigned bad = 0u;
  constexpr int UNR = 4;
  const eoff_t STEP = (eoff_t)UNR * THREADS;
  pb_f32x4 xs[UNR], nx[UNR];
  pb_u16x4 vs[UNR], nv[UNR];
  auto ld = [&](eoff_t qq, pb_f32x4 &xv, pb_u16x4 &vv) {
    if (FLAGS & 16) {
      xv = X4[qq];
      vv = V4[qq];
    } else {
      xv = __builtin_nontemporal_load(X4 + qq);
      vv = __builtin_nontemporal_load(V4 + qq);
    }
  };
  auto fold = [&](const pb_f32x4 &xv, const pb_u16x4 &vv) {
    if (FLAGS & 1) {
      bad |= (unsigned)(xv.x + xv.y + xv.z + xv.w == 123.456f) + (unsigned)(vv.x + vv.w == 77777u);
      return;
    }
    unsigned cur = vv.x;
    unsigned long long a = op.to_fixed(xv.x, bad);
    unsigned long long f = op.to_fixed(xv.y, bad);
    if (vv.y == cur) a += f;
    else {
      atomicAdd(&s_acc[cur], a);
      cur = vv.y;
      a = f;
    }
    f = op.to_fixed(xv.z, bad);
    if (vv.z == cur) a += f;
    else {
      atomicAdd(&s_acc[cur], a);
      cur = vv.z;
      a = f;
    }
    f = op.to_fixed(xv.w, bad);
    if (vv.w == cur) a += f;
    else {
      atomicAdd(&s_acc[cur], a);
      cur = vv.w;
      a = f;
    }
    atomicAdd(&s_acc[cur], a);
  };
  if (FLAGS & 4) {
    for (eoff_t q = q0 + threadIdx.x; q < q1; q += STEP) {
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const eoff_t qq = q + (eoff_t)r * THREADS;
        if (qq < q1) ld(qq, xs[r], vs[r]);
      }
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const eoff_t qq = q + (eoff_t)r * THREADS;
        if (qq < q1) fold(xs[r], vs[r]);
      }
    }
  } else {
    {
      const eoff_t q = q0 + threadIdx.x;
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const eoff_t qq = q + (eoff_t)r * THREADS;
        if (qq < q1) ld(qq, xs[r], vs[r]);
      }
    }
    for (eoff_t q = q0 + threadIdx.x; q < q1; q += STEP) {
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const eoff_t qq = q + STEP + (eoff_t)r * THREADS;
        if (qq < q1) ld(qq, nx[r], nv[r]);
      }
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        const eoff_t qq = q + (eoff_t)r * THREADS;
        if (qq < q1) fold(xs[r], vs[r]);
      }
#pragma unroll
      for (int r = 0; r < UNR; r++) {
        xs[r] = nx[r];
        vs[r] = nv[r];
      }
    }
  }
  __syncthreads();
  double dsum = 0.0;
  if (!(FLAGS & 2) && THREADS == PB_THREADS) {
    const unsigned lo = bin_lo[b], hi = bin_lo[b + 1];
    if (FLAGS & 8)
      dsum = pb_epilogue(dst_bits, lo, hi, s_bits, s_pref, s_scr, op, [&](unsigned k) { return op.from_fixed(s_acc[k], bad); });
    else
      dsum = pb_epilogue4(dst_bits, lo, hi, s_bits, s_pref, s_scr, op, [&](unsigned k) { return op.from_fixed(s_acc[k], bad); });
  }
  if (bad) *errflag = 1u;
  dsum = gdn_wave_sum(dsum);
  if (gdn_lane() == 0 && dsum != 0.0) partial[b] = dsum;
}

// ---- phase B with vals in CHUNK-major order, gathered through GB (bin-major group -> chunk-major group).
// FLAGS bit0: no atomics, bit1: no epilogue
template <int FLAGS, int UNR, class Op>
__global__ void __launch_bounds__(PB_THREADS)
labBg(int32_t m_local, int log_bin, const eoff_t *__restrict__ bin_ptr, const uint32_t *__restrict__ order,
      const uint16_t *__restrict__ V, const uint32_t *__restrict__ GB, const float *__restrict__ vals,
      double *__restrict__ partial, unsigned *__restrict__ errflag, const uint32_t *__restrict__ dst_bits,
      const uint32_t *__restrict__ bin_lo, Op op) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];
  __shared__ unsigned s_bits[PB_THREADS], s_pref[PB_THREADS], s_scr[PB_WAVES + 1];
  const unsigned bn = 1u << log_bin;
  const unsigned b = order ? order[blockIdx.x] : blockIdx.x;
  for (unsigned i = threadIdx.x; i < bn; i += PB_THREADS) s_acc[i] = 0ull;
  __syncthreads();
  const eoff_t q0 = bin_ptr[b] >> 2, q1 = bin_ptr[b + 1] >> 2;
  const pb_f32x4 *X4 = reinterpret_cast<const pb_f32x4 *>(vals);
  const pb_u16x4 *V4 = reinterpret_cast<const pb_u16x4 *>(V);
  unsigned bad = 0u;
  const eoff_t STEP = (eoff_t)UNR * PB_THREADS;
  pb_f32x4 xs[UNR];
  pb_u16x4 vs[UNR], nv[UNR];
  unsigned gb[UNR];
  auto fold = [&](const pb_f32x4 &xv, const pb_u16x4 &vv) {
    if (FLAGS & 1) {
      bad |= (unsigned)(xv.x + xv.y + xv.z + xv.w == 123.456f) + (unsigned)(vv.x + vv.w == 77777u);
      return;
    }
    unsigned cur = vv.x;
    unsigned long long a = op.to_fixed(xv.x, bad);
    unsigned long long f = op.to_fixed(xv.y, bad);
    if (vv.y == cur) a += f;
    else {
      atomicAdd(&s_acc[cur], a);
      cur = vv.y;
      a = f;
    }
    f = op.to_fixed(xv.z, bad);
    if (vv.z == cur) a += f;
    else {
      atomicAdd(&s_acc[cur], a);
      cur = vv.z;
      a = f;
    }
    f = op.to_fixed(xv.w, bad);
    if (vv.w == cur) a += f;
    else {
      atomicAdd(&s_acc[cur], a);
      cur = vv.w;
      a = f;
    }
    atomicAdd(&s_acc[cur], a);
  };
  // pipeline: GB and V of step i+1 are loaded while the vals of step i (addressed by the GB loaded one step
  // earlier) are in flight
  {
    const eoff_t q = q0 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + (eoff_t)r * PB_THREADS;
      if (qq < q1) {
        gb[r] = __builtin_nontemporal_load(GB + (qq >> 1));
        nv[r] = __builtin_nontemporal_load(V4 + qq);
      }
    }
  }
  for (eoff_t q = q0 + threadIdx.x; q < q1; q += STEP) {
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + (eoff_t)r * PB_THREADS;
      if (qq < q1) {
        xs[r] = __builtin_nontemporal_load(X4 + 2 * (size_t)gb[r] + (size_t)(qq & 1));
        vs[r] = nv[r];
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + STEP + (eoff_t)r * PB_THREADS;
      if (qq < q1) {
        gb[r] = __builtin_nontemporal_load(GB + (qq >> 1));
        nv[r] = __builtin_nontemporal_load(V4 + qq);
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; r++) {
      const eoff_t qq = q + (eoff_t)r * PB_THREADS;
      if (qq < q1) fold(xs[r], vs[r]);
    }
  }
  __syncthreads();
  double dsum = 0.0;
  if (!(FLAGS & 2)) {
    const unsigned lo = bin_lo[b], hi = bin_lo[b + 1];
    dsum = pb_epilogue4(dst_bits, lo, hi, s_bits, s_pref, s_scr, op, [&](unsigned k) { return op.from_fixed(s_acc[k], bad); });
  }
  if (bad) *errflag = 1u;
  dsum = gdn_wave_sum(dsum);
  if (gdn_lane() == 0 && dsum != 0.0) partial[b] = dsum;
}

__global__ void invert_groups(const uint32_t *__restrict__ G, uint32_t *__restrict__ GB, size_t ng, uint32_t dump) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < ng; i += st) {
    const uint32_t d = G[i];
    if (d != dump) GB[d] = (uint32_t)i;
  }
}

__global__ void fill_rand(float *p, size_t n, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 29;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 32;
    p[i] = (float)(z & 0xFFFFFF) * (1.0f / 16777216.0f) * scale;
  }
}

template <class F>
static float timeit(F f, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(a));
    f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return best;
}

int main(int argc, char **argv) {
  const int scale = argc > 1 ? atoi(argv[1]) : 27;
  const int lc = argc > 2 ? atoi(argv[2]) : 15, lb = argc > 3 ? atoi(argv[3]) : 14;
  gdn_graph *go = nullptr, *gi = nullptr;
  GK(gdn_rmat_build(scale, 16, 27491095ull, 1, &go, &gi));
  const int32_t m = gi->m;
  DevBuf<int32_t> deg;
  GK(deg.alloc(m));
  GK(gdn_graph_degrees_dev(go, deg.p, nullptr));
  gdn_graph_free(go);
  PbPlan pb;
  const unsigned pad = getenv("GDN_PB_PAD") ? (unsigned)atoi(getenv("GDN_PB_PAD")) : 32u;
  GK(pb_build(gi, m, lc, lb, pb, true, nullptr, nullptr, true, false, pad, 3));  // the lab kernels use 8-edge groups
  printf("RMAT-%d: m %d nnz %llu  chunks %u (2^%d) bins %u (2^%d) n_pad %llu (%.3f x nnz)\n", scale, m,
         (unsigned long long)gi->nnz, pb.nchunks, lc, pb.nbins, lb, (unsigned long long)pb.n_pad, (double)pb.n_pad / gi->nnz);
  DevBuf<float> x, scores, cout;
  GK(x.alloc(m));
  GK(scores.alloc(m));
  GK(cout.alloc(m));
  fill_rand<<<4096, 256>>>(x.p, (size_t)m, 1e-8f);
  fill_rand<<<4096, 256>>>(scores.p, (size_t)m, 1e-8f);
  CK(hipDeviceSynchronize());
  LabOp op;
  op.scores = scores.p;
  op.contrib_out = cout.p;
  op.deg = deg.p;
  op.base_score = 0.15f / (float)m;
  op.damping = 0.85f;
  op.vec_ok = true;
  const size_t ldsA = (sizeof(float) << lc) + 16, ldsB = sizeof(unsigned long long) << lb;
  const double ebytesA = 6.5, ebytesB = 6.0;
#define RUNA(FL, TH, ORD, GRID, name)                                                                                    \
  {                                                                                                                      \
    auto k = labA<FL, TH>;                                                                                               \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA));                     \
    float ms = timeit([&] {                                                                                              \
      hipLaunchKernelGGL(k, dim3(GRID), dim3(TH), ldsA, 0, x.p, m, lc, pb.chunk_ptr.p, ORD, pb.U.p, pb.G.p, pb.vals.p,   \
                         pb.src_bits.p, pb.chunk_lo.p);                                                                  \
    });                                                                                                                  \
    printf("A %-64s %7.3f ms  %5.2f TB/s\n", name, ms, pb.n_pad * ebytesA * ((double)(GRID) / pb.nchunks) / ms / 1e9);   \
    fflush(stdout);                                                                                                      \
  }
#define RUNB(FL, TH, ORD, GRID, name)                                                                                    \
  {                                                                                                                      \
    auto k = labB<FL, TH, LabOp>;                                                                                        \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));                     \
    float ms = timeit([&] {                                                                                              \
      hipLaunchKernelGGL(k, dim3(GRID), dim3(TH), ldsB, 0, m, lb, pb.bin_ptr.p, ORD, pb.V.p, pb.vals.p, pb.partial.p,    \
                         pb.errflag.p, pb.dst_bits.p, pb.bin_lo.p, op);                                                  \
    });                                                                                                                  \
    printf("B %-64s %7.3f ms  %5.2f TB/s\n", name, ms, pb.n_pad * ebytesB * ((double)(GRID) / pb.nbins) / ms / 1e9);     \
    fflush(stdout);                                                                                                      \
  }
  const uint32_t *NOORD = nullptr;
  RUNA(0, 1024, pb.chunk_order.p, pb.nchunks, "product form (largest-first order)");
  RUNA(0, 1024, NOORD, pb.nchunks, "identity order");
  RUNA(1, 1024, pb.chunk_order.p, pb.nchunks, "no slice load");
  RUNA(2, 1024, pb.chunk_order.p, pb.nchunks, "plain U/G loads");
  RUNA(4, 1024, pb.chunk_order.p, pb.nchunks, "sequential stores (no G)");
  RUNA(8, 1024, pb.chunk_order.p, pb.nchunks, "UNR 4");
  RUNA(16, 1024, pb.chunk_order.p, pb.nchunks, "read-only (no stores)");
  RUNA(17, 1024, pb.chunk_order.p, pb.nchunks, "read-only, no slice load");
  RUNA(32, 1024, pb.chunk_order.p, pb.nchunks, "write-only (no U/G loads, sequential)");
  RUNA(33, 1024, pb.chunk_order.p, pb.nchunks, "write-only, no slice load");
  RUNA(0, 1024, pb.chunk_order.p, 256, "first 256 chunks only (one round)");
  RUNA(0, 1024, pb.chunk_order.p, 512, "first 512 chunks only (two rounds)");
  RUNB(0, 1024, pb.bin_order.p, pb.nbins, "product form (largest-first order)");
  RUNB(0, 1024, NOORD, pb.nbins, "identity order");
  RUNB(8, 1024, pb.bin_order.p, pb.nbins, "scalar epilogue");
  RUNB(2, 1024, pb.bin_order.p, pb.nbins, "no epilogue");
  RUNB(3, 1024, pb.bin_order.p, pb.nbins, "no epilogue, no atomics (pure stream)");
  RUNB(3, 1024, NOORD, pb.nbins, "pure stream, identity order");
  RUNB(7, 1024, pb.bin_order.p, pb.nbins, "pure stream, simple loop");
  RUNB(7, 1024, NOORD, pb.nbins, "pure stream, simple loop, identity order");
  RUNB(19, 1024, pb.bin_order.p, pb.nbins, "pure stream, plain loads");
  RUNB(4, 1024, pb.bin_order.p, pb.nbins, "simple loop, full");
  RUNB(3, 512, pb.bin_order.p, pb.nbins, "pure stream, 512 threads");
  RUNB(3, 1024, pb.bin_order.p, 256, "pure stream, first 256 bins (one round)");
  RUNB(3, 1024, pb.bin_order.p, 512, "pure stream, first 512 bins (two rounds)");
  RUNB(0, 1024, pb.bin_order.p, 256, "product form, first 256 bins (one round)");
  {
    DevBuf<uint32_t> GB;
    const size_t ng = (size_t)(pb.n_pad >> 3);
    GK(GB.alloc(ng + 1));
    CK(hipMemset(GB.p, 0, (ng + 1) * 4));
    invert_groups<<<8192, 256>>>(pb.G.p, GB.p, ng, (uint32_t)ng);
    CK(hipDeviceSynchronize());
#define RUNBG(FL, UN, name)                                                                                             \
  {                                                                                                                      \
    auto k = labBg<FL, UN, LabOp>;                                                                                       \
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));                     \
    float ms = timeit([&] {                                                                                              \
      hipLaunchKernelGGL(k, dim3(pb.nbins), dim3(PB_THREADS), ldsB, 0, m, lb, pb.bin_ptr.p, pb.bin_order.p, pb.V.p, GB.p, \
                         pb.vals.p, pb.partial.p, pb.errflag.p, pb.dst_bits.p, pb.bin_lo.p, op);                         \
    });                                                                                                                  \
    printf("B %-64s %7.3f ms  %5.2f TB/s\n", name, ms, pb.n_pad * 6.5 / ms / 1e9);                                       \
    fflush(stdout);                                                                                                      \
  }
    RUNBG(0, 4, "GATHER vals through GB (chunk-major vals), full, UNR 4");
    RUNBG(0, 2, "GATHER full, UNR 2");
    RUNBG(0, 8, "GATHER full, UNR 8");
    RUNBG(3, 4, "GATHER pure stream, UNR 4");
    RUNBG(2, 4, "GATHER no epilogue, UNR 4");
  }
  unsigned ef = 0;
  CK(hipMemcpy(&ef, pb.errflag.p, 4, hipMemcpyDeviceToHost));
  printf("errflag %u\n", ef);
  gdn_graph_free(gi);
  return 0;
}
