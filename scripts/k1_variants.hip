// k1_variants.hip -- micro-benchmark of launch shapes / store flavours for the Hessian-assembly kernel
// (DESIGN.md section 4).  Not part of the library; the winning variant lives in csrc/mrs_tg_kernels.hip.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mrs_uav_trajectory_generation_amd/csrc -I include \
//         scripts/k1_variants.hip -o gpurun_out/k1_variants && gpurun_out/k1_variants [P] [S]
//
// Every variant writes the same slot-major SoA blocks: element e of segment j of path q at ((j*100+e)*P + q).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mrs_tg_device.hpp"
// the library's own assembly kernel, timed under the same conditions as the variants
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_kernels.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_tile.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_rows.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_quad.hip"
#include "../mrs_uav_trajectory_generation_amd/csrc/mrs_tg_pool.hip"

using namespace mrs_tg;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ void block_rows(double T, int d, int a, double (&hrow)[kN], double (&arow)[kN]) {
  double w[kHalf];
  w[0] = 1.0;
#pragma unroll
  for (int k = 1; k < kHalf; ++k) w[k] = w[k - 1] * T;
  const int pa = a % kHalf;
  double td = 1.0;
  if (d == 1) td = w[1];
  else if (d == 2) td = w[2];
  else if (d == 3) td = w[3];
  else if (d == 4) td = w[4];
  double wa = 1.0;
  if (pa == 1) wa = w[1];
  else if (pa == 2) wa = w[2];
  else if (pa == 3) wa = w[3];
  else if (pa == 4) wa = w[4];
  const double sa = (T / (td * td)) * wa;
#pragma unroll
  for (int c = 0; c < kN; ++c) hrow[c] = c_hbar[d][a][c] * sa * w[c % kHalf];
  const double ti = 1.0 / T;
  double tia = 1.0;
  for (int k = 0; k < a; ++k) tia *= ti;
#pragma unroll
  for (int c = 0; c < kN; ++c) arow[c] = c_abar_inv[a][c] * w[c % kHalf] * tia;
}

template <bool NT>
__device__ __forceinline__ void st2(double* p, double a, double b) {
  if (NT) {
    __builtin_nontemporal_store(a, p);
    __builtin_nontemporal_store(b, p + 1);
  } else {
    *reinterpret_cast<double2*>(p) = make_double2(a, b);
  }
}

typedef double dbl2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ void st2v(double* p, double a, double b) {
  dbl2 v = {a, b};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(p));
  else *reinterpret_cast<dbl2*>(p) = v;
}

// V_pair<BLOCK, NT>: the library's uniform kernel (two neighbouring paths per lane, blockIdx.y = row)
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_pair(int n_paths, int S, int d, const double* __restrict__ seg_times,
                                                double* __restrict__ Hout, double* __restrict__ Aout) {
  const int half = n_paths >> 1;
  const int idx = blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= half * S) return;
  const int a = blockIdx.y;
  const int j = idx / half;
  const int q = (idx - j * half) * 2;
  const double T0 = seg_times[(size_t)q * S + j];
  const double T1 = seg_times[(size_t)(q + 1) * S + j];
  double h0[kN], a0[kN], h1[kN], a1[kN];
  block_rows(T0, d, a, h0, a0);
  block_rows(T1, d, a, h1, a1);
  const size_t P = (size_t)n_paths;
  const size_t base = ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q;
#pragma unroll
  for (int c = 0; c < kN; ++c) st2v<NT>(Hout + base + (size_t)c * P, h0[c], h1[c]);
#pragma unroll
  for (int c = 0; c < kN; ++c) st2v<NT>(Aout + base + (size_t)c * P, a0[c], a1[c]);
}

// V_single<BLOCK, NT>: one path per lane, 8-byte stores
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_single(int n_paths, int S, int d, const double* __restrict__ seg_times,
                                                  double* __restrict__ Hout, double* __restrict__ Aout) {
  const int idx = blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= n_paths * S) return;
  const int a = blockIdx.y;
  const int j = idx / n_paths;
  const int q = idx - j * n_paths;
  const double T0 = seg_times[(size_t)q * S + j];
  double h0[kN], a0[kN];
  block_rows(T0, d, a, h0, a0);
  const size_t P = (size_t)n_paths;
  const size_t base = ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q;
#pragma unroll
  for (int c = 0; c < kN; ++c) {
    if (NT) __builtin_nontemporal_store(h0[c], Hout + base + (size_t)c * P);
    else Hout[base + (size_t)c * P] = h0[c];
  }
#pragma unroll
  for (int c = 0; c < kN; ++c) {
    if (NT) __builtin_nontemporal_store(a0[c], Aout + base + (size_t)c * P);
    else Aout[base + (size_t)c * P] = a0[c];
  }
}

// V_half<BLOCK, NT>: two paths per lane, blockIdx.y = (row, matrix): a thread writes one row of H or of A^-1
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_half(int n_paths, int S, int d, const double* __restrict__ seg_times,
                                                double* __restrict__ Hout, double* __restrict__ Aout) {
  const int half = n_paths >> 1;
  const int idx = blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= half * S) return;
  const int a = blockIdx.y >> 1;
  const int which = blockIdx.y & 1;
  const int j = idx / half;
  const int q = (idx - j * half) * 2;
  const double T0 = seg_times[(size_t)q * S + j];
  const double T1 = seg_times[(size_t)(q + 1) * S + j];
  double h0[kN], a0[kN], h1[kN], a1[kN];
  block_rows(T0, d, a, h0, a0);
  block_rows(T1, d, a, h1, a1);
  const size_t P = (size_t)n_paths;
  const size_t base = ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q;
  if (which == 0) {
#pragma unroll
    for (int c = 0; c < kN; ++c) st2v<NT>(Hout + base + (size_t)c * P, h0[c], h1[c]);
  } else {
#pragma unroll
    for (int c = 0; c < kN; ++c) st2v<NT>(Aout + base + (size_t)c * P, a0[c], a1[c]);
  }
}

// pure fill of the same two buffers (upper bound for any kernel of this size)
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_fill(size_t n_pairs, double* __restrict__ Hout, double* __restrict__ Aout) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (; i < n_pairs; i += stride) {
    st2v<NT>(Hout + 2 * i, 1.0, 2.0);
    st2v<NT>(Aout + 2 * i, 3.0, 4.0);
  }
}

// reciprocal by v_rcp_f64 + two Newton steps (<= 1 ulp) instead of the IEEE division sequence
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

template <bool FAST>
__device__ __forceinline__ void h_row(double T, int d, int a, double (&hrow)[kN]) {
  const double t2 = T * T, t3 = t2 * T, t4 = t2 * t2;
  const double w[kHalf] = {1.0, T, t2, t3, t4};
  const int pa = a % kHalf;
  const double td = (d == 0) ? 1.0 : (d == 1) ? T : (d == 2) ? t2 : (d == 3) ? t3 : t4;
  const double wa = (pa == 0) ? 1.0 : (pa == 1) ? T : (pa == 2) ? t2 : (pa == 3) ? t3 : t4;
  const double inv = FAST ? fast_rcp(td * td) : 1.0 / (td * td);
  const double sa = T * inv * wa;
#pragma unroll
  for (int c = 0; c < kN; ++c) hrow[c] = c_hbar[d][a][c] * sa * w[c % kHalf];
}
template <bool FAST>
__device__ __forceinline__ void a_row(double T, int a, double (&arow)[kN]) {
  const double t2 = T * T, t3 = t2 * T, t4 = t2 * t2;
  const double w[kHalf] = {1.0, T, t2, t3, t4};
  const double ti = FAST ? fast_rcp(T) : 1.0 / T;
  double tia = 1.0;
  for (int k = 0; k < a; ++k) tia *= ti;
#pragma unroll
  for (int c = 0; c < kN; ++c) arow[c] = c_abar_inv[a][c] * w[c % kHalf] * tia;
}

// V_z<BLOCK, FAST>: one path per lane; blockIdx.z = segment slot (no integer division), blockIdx.y = row;
// H row stored before the A^-1 row is computed
template <int BLOCK, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_z(int n_paths, int S, int d, const double* __restrict__ seg_times,
                                             double* __restrict__ Hout, double* __restrict__ Aout) {
  const int q = blockIdx.x * BLOCK + threadIdx.x;
  if (q >= n_paths) return;
  const int a = blockIdx.y;
  const int j = blockIdx.z;
  const double T0 = seg_times[(size_t)q * S + j];
  const size_t P = (size_t)n_paths;
  const size_t base = ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q;
  double h0[kN], a0[kN];
  h_row<FAST>(T0, d, a, h0);
#pragma unroll
  for (int c = 0; c < kN; ++c) Hout[base + (size_t)c * P] = h0[c];
  a_row<FAST>(T0, a, a0);
#pragma unroll
  for (int c = 0; c < kN; ++c) Aout[base + (size_t)c * P] = a0[c];
}

// V_zs<BLOCK, FAST>: as V_z, but blockIdx.y = 2*row + matrix: a thread writes one row of one matrix
template <int BLOCK, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_zs(int n_paths, int S, int d, const double* __restrict__ seg_times,
                                              double* __restrict__ Hout, double* __restrict__ Aout) {
  const int q = blockIdx.x * BLOCK + threadIdx.x;
  if (q >= n_paths) return;
  const int a = blockIdx.y >> 1;
  const int j = blockIdx.z;
  const double T0 = seg_times[(size_t)q * S + j];
  const size_t P = (size_t)n_paths;
  const size_t base = ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q;
  double r[kN];
  double* out;
  if (blockIdx.y & 1) {
    a_row<FAST>(T0, a, r);
    out = Aout;
  } else {
    h_row<FAST>(T0, d, a, r);
    out = Hout;
  }
#pragma unroll
  for (int c = 0; c < kN; ++c) out[base + (size_t)c * P] = r[c];
}

// V_z2<BLOCK, FAST>: two rows per thread (blockIdx.y = row pair)
template <int BLOCK, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_z2(int n_paths, int S, int d, const double* __restrict__ seg_times,
                                              double* __restrict__ Hout, double* __restrict__ Aout) {
  const int q = blockIdx.x * BLOCK + threadIdx.x;
  if (q >= n_paths) return;
  const int j = blockIdx.z;
  const double T0 = seg_times[(size_t)q * S + j];
  const size_t P = (size_t)n_paths;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int a = blockIdx.y * 2 + rr;
    const size_t base = ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q;
    double h0[kN], a0[kN];
    h_row<FAST>(T0, d, a, h0);
#pragma unroll
    for (int c = 0; c < kN; ++c) Hout[base + (size_t)c * P] = h0[c];
    a_row<FAST>(T0, a, a0);
#pragma unroll
    for (int c = 0; c < kN; ++c) Aout[base + (size_t)c * P] = a0[c];
  }
}

// V_zp<BLOCK, NP>: as V_z with NP (2 or 4) neighbouring paths per lane (16- / 2x16-byte stores)
template <int BLOCK, int NP>
__global__ __launch_bounds__(BLOCK) void k_zp(int n_paths, int S, int d, const double* __restrict__ seg_times,
                                              double* __restrict__ Hout, double* __restrict__ Aout) {
  const int q = (blockIdx.x * BLOCK + threadIdx.x) * NP;
  if (q >= n_paths) return;
  const int a = blockIdx.y;
  const int j = blockIdx.z;
  double T[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) T[i] = seg_times[(size_t)(q + i) * S + j];
  const size_t P = (size_t)n_paths;
  const size_t base = ((size_t)j * 100 + (size_t)a * kN) * P + (size_t)q;
  double r[NP][kN];
#pragma unroll
  for (int i = 0; i < NP; ++i) h_row<false>(T[i], d, a, r[i]);
#pragma unroll
  for (int c = 0; c < kN; ++c)
#pragma unroll
    for (int i = 0; i < NP; i += 2) st2v<false>(Hout + base + (size_t)c * P + i, r[i][c], r[i + 1][c]);
#pragma unroll
  for (int i = 0; i < NP; ++i) a_row<false>(T[i], a, r[i]);
#pragma unroll
  for (int c = 0; c < kN; ++c)
#pragma unroll
    for (int i = 0; i < NP; i += 2) st2v<false>(Aout + base + (size_t)c * P + i, r[i][c], r[i + 1][c]);
}

// V_zpad<BLOCK>: V_z with a padded slot stride (element e of slot j at (j*100+e)*stride + q)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_zpad(int n_paths, int S, int d, size_t stride, const double* __restrict__ seg_times,
                                                double* __restrict__ Hout, double* __restrict__ Aout) {
  const int q = blockIdx.x * BLOCK + threadIdx.x;
  if (q >= n_paths) return;
  const int a = blockIdx.y;
  const int j = blockIdx.z;
  const double T0 = seg_times[(size_t)q * S + j];
  const size_t base = ((size_t)j * 100 + (size_t)a * kN) * stride + (size_t)q;
  double h0[kN], a0[kN];
  h_row<false>(T0, d, a, h0);
#pragma unroll
  for (int c = 0; c < kN; ++c) Hout[base + (size_t)c * stride] = h0[c];
  a_row<false>(T0, a, a0);
#pragma unroll
  for (int c = 0; c < kN; ++c) Aout[base + (size_t)c * stride] = a0[c];
}

// pure fill with 8-byte stores per lane (calibrates WRITE_SIZE for the 8-B-per-lane store pattern of V_z)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fill8(size_t n, double* __restrict__ Hout, double* __restrict__ Aout) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (; i < n; i += stride) {
    Hout[i] = 1.0;
    Aout[i] = 3.0;
  }
}

__global__ void k_empty() {}

static int g_launches = 200, g_reps = 5, g_warm = 20;

template <typename F>
static void measure(const char* name, F launch, hipStream_t st, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < g_warm; ++i) launch();
  CK(hipStreamSynchronize(st));
  float best = 1e30f, sum = 0;
  const int reps = g_reps, n = g_launches;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
    sum += ms;
  }
  const double us_best = best * 1e3 / n, us_mean = sum * 1e3 / (n * reps);
  printf("%-28s  back-to-back us/launch: best %.3f mean %.3f   GB/s(best) %.0f  frac of 8 TB/s %.3f\n", name, us_best,
         us_mean, bytes / us_best * 1e-3, bytes / us_best * 1e-3 / 8000.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 1024;
  const int S = argc > 2 ? atoi(argv[2]) : 10;
  const int d = 4;
  if (argc > 3) {  // "pmc" mode: a handful of launches per variant so that a counter pass stays small
    g_launches = atoi(argv[3]);
    g_reps = 1;
    g_warm = 1;
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t nblk = (size_t)P * S * 100;
  double *H, *A, *T;
  CK(hipMalloc(&H, nblk * 8 + (size_t)S * 100 * 8 * 1024));
  CK(hipMalloc(&A, nblk * 8 + (size_t)S * 100 * 8 * 1024));
  CK(hipMalloc(&T, (size_t)P * S * 8));
  std::vector<double> t((size_t)P * S);
  for (size_t i = 0; i < t.size(); ++i) t[i] = 0.5 + 5.0 * ((i * 2654435761u) % 1000) / 1000.0;
  CK(hipMemcpy(T, t.data(), t.size() * 8, hipMemcpyHostToDevice));
  const double bytes = 1608.0 * P * S;
  printf("P=%d S=%d algorithmic bytes/launch %.0f\n", P, S, bytes);
  const int half = P / 2;
  auto cdiv = [](long long a, long long b) { return (unsigned)((a + b - 1) / b); };

  measure("empty", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, st, 0.0);
#define FILL(B, NT, G)                                                                                        \
  measure("fill<" #B "," #NT "> grid " #G, [&] { hipLaunchKernelGGL((k_fill<B, NT>), dim3(G), dim3(B), 0, st, nblk / 2, H, A); }, st, bytes)
  measure("fill8<256> grid 2048", [&] { hipLaunchKernelGGL((k_fill8<256>), dim3(2048), dim3(256), 0, st, nblk, H, A); }, st, bytes);
  FILL(256, false, 1024);
  FILL(256, true, 1024);
  FILL(256, false, 2048);
  FILL(256, true, 2048);
  FILL(64, false, 4096);
  FILL(1024, false, 256);
  FILL(512, false, 512);
#define PAIR(B, NT)                                                                                        \
  measure("pair<" #B "," #NT ">", [&] { hipLaunchKernelGGL((k_pair<B, NT>), dim3(cdiv((long long)half * S, B), kN), dim3(B), 0, st, P, S, d, T, H, A); }, st, bytes)
  PAIR(256, false);
  PAIR(256, true);
  PAIR(128, false);
  PAIR(128, true);
  PAIR(64, false);
  PAIR(64, true);
#define SINGLE(B, NT)                                                                                      \
  measure("single<" #B "," #NT ">", [&] { hipLaunchKernelGGL((k_single<B, NT>), dim3(cdiv((long long)P * S, B), kN), dim3(B), 0, st, P, S, d, T, H, A); }, st, bytes)
  SINGLE(256, false);
  SINGLE(256, true);
  SINGLE(64, false);
  SINGLE(64, true);
#define HALF(B, NT)                                                                                        \
  measure("half<" #B "," #NT ">", [&] { hipLaunchKernelGGL((k_half<B, NT>), dim3(cdiv((long long)half * S, B), 2 * kN), dim3(B), 0, st, P, S, d, T, H, A); }, st, bytes)
  HALF(256, false);
  HALF(256, true);
  HALF(128, false);
  HALF(64, false);
  HALF(64, true);
#define ZV(K, B, FAST, GY)                                                                                   \
  measure(#K "<" #B "," #FAST ">", [&] { hipLaunchKernelGGL((K<B, FAST>), dim3(cdiv(P, B), GY, S), dim3(B), 0, st, P, S, d, T, H, A); }, st, bytes)
  ZV(k_z, 256, false, kN);
  ZV(k_z, 256, true, kN);
  ZV(k_z, 128, true, kN);
  ZV(k_z, 64, true, kN);
  ZV(k_zs, 256, true, 2 * kN);
  ZV(k_zs, 128, true, 2 * kN);
  ZV(k_zs, 64, true, 2 * kN);
  ZV(k_z2, 256, true, kN / 2);
  ZV(k_z2, 64, true, kN / 2);
#define ZP(B, NP)                                                                                   \
  measure("k_zp<" #B "," #NP ">", [&] { hipLaunchKernelGGL((k_zp<B, NP>), dim3(cdiv(P, B * NP), kN, S), dim3(B), 0, st, P, S, d, T, H, A); }, st, bytes)
  ZP(256, 2);
  ZP(128, 2);
  ZP(64, 2);
  ZP(256, 4);
  ZP(128, 4);
  ZP(64, 4);
  ZV(k_z, 512, false, kN);
  ZV(k_z, 1024, false, kN);
#define ZPAD(B, PAD)                                                                                   \
  measure("k_zpad<" #B "> pad " #PAD, [&] { hipLaunchKernelGGL((k_zpad<B>), dim3(cdiv(P, B), kN, S), dim3(B), 0, st, P, S, d, (size_t)P + PAD, T, H, A); }, st, bytes)
  ZPAD(256, 0);
  ZPAD(256, 16);
  ZPAD(256, 32);
  ZPAD(256, 64);
  ZPAD(256, 128);
  ZPAD(256, 256);
  ZPAD(256, 512);
  ZPAD(128, 32);
  ZPAD(128, 64);
  ZPAD(128, 256);
  ZPAD(64, 64);
  ZPAD(512, 64);
  measure("library assemble_blocks_uniform_kernel", [&] {
    hipLaunchKernelGGL(assemble_blocks_uniform_kernel, dim3(cdiv(P, kAssembleChunk) * kN * S), dim3(kAssembleChunk), 0, st, P, S, d,
                       T, H, A);
  }, st, bytes);
  return 0;
}
