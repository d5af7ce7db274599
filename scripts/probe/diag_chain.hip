// Probe: clocks of the 16 x 16 diagonal-block chain (dev_diag_block of kernels_front_pivot.inc) on one wave, and its
// result against a plain host LDL^T + inverse.  Build twice to compare the instruction orders:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 diag_chain.hip -o diag_chain
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DHIPFACT_DIAG_INTERLEAVED diag_chain.hip -o diag_chain_interleaved
// MI355X: 2220 shader clocks as the compiler orders it (chain in one piece behind the row updates), 2416 with the
// row updates dealt between the chain instructions: the wave is bound by ISSUE (one instruction per 4 clocks,
// scripts/probe/valu_issue.hip: a dependent fp64 FMA costs no more than an independent one), not by latency.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../sleqp_amd/csrc/device_types.h"
#include "../../sleqp_amd/csrc/kernels_factor.hip"
using namespace hipfact;

__global__ __launch_bounds__(64) void k_probe(const double* __restrict__ Ain, double* __restrict__ out,
                                              long long* __restrict__ ticks, int* __restrict__ info, int reps) {
  __shared__ double lds[16 + 16 * 17 + 64];
  FrontCtx c;
  c.w = c.r = c.wp = 16;
  c.u = 0;
  c.nbk = 1;
  c.lda = 17;
  c.dd = lds;
  c.A = lds + 16;
  c.Yp = nullptr;
  c.P = nullptr;
  c.Us = nullptr;
  c.Xa = nullptr;
  double* scratch = lds + 16 + 16 * 17;
  long long best = 1ll << 60;
  for (int r = 0; r < reps; ++r) {
    for (int e = threadIdx.x; e < 256; e += 64) c.A[(e & 15) + (e >> 4) * 17] = Ain[e];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    dev_diag_block(c, scratch, 0, info);
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    best = (t1 - t0) < best ? (t1 - t0) : best;
  }
  for (int e = threadIdx.x; e < 256; e += 64) out[e] = c.A[(e & 15) + (e >> 4) * 17];
  if (threadIdx.x < 16) out[256 + threadIdx.x] = c.dd[threadIdx.x];
  if (threadIdx.x == 0) *ticks = best;
}

int main() {
  std::vector<double> A(256), L(256, 0.0), d(16), X(256, 0.0);
  // SPD: B B^T + 16 I, column major
  std::vector<double> B(256);
  unsigned s = 12345;
  for (auto& b : B) {
    s = s * 1664525u + 1013904223u;
    b = ((s >> 8) & 0xffff) / 65536.0 - 0.5;
  }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double v = (i == j) ? 4.0 : 0.0;
      for (int k = 0; k < 16; ++k) v += B[i + 16 * k] * B[j + 16 * k];
      A[i + 16 * j] = v;
    }
  // host LDL^T and inverse of the unit factor
  std::vector<double> W = A;
  for (int k = 0; k < 16; ++k) {
    d[k] = W[k + 16 * k];
    for (int i = k + 1; i < 16; ++i) L[i + 16 * k] = W[i + 16 * k] / d[k];
    for (int j = k + 1; j < 16; ++j)
      for (int i = j; i < 16; ++i) W[i + 16 * j] -= L[i + 16 * k] * d[k] * L[j + 16 * k];
  }
  for (int j = 0; j < 16; ++j) {
    X[j + 16 * j] = 1.0;
    for (int i = j + 1; i < 16; ++i) {
      double v = 0.0;
      for (int k = j; k < i; ++k) v -= L[i + 16 * k] * X[k + 16 * j];
      X[i + 16 * j] = v;
    }
  }
  double *dA, *dout;
  long long* dt;
  int* dinfo;
  hipMalloc(&dA, 256 * 8);
  hipMalloc(&dout, 272 * 8);
  hipMalloc(&dt, 8);
  hipMalloc(&dinfo, 4096);
  hipMemset(dinfo, 0, 4096);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dout, dt, dinfo, 200);
  std::vector<double> out(272);
  long long t;
  hipMemcpy(out.data(), dout, 272 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  double ex = 0.0, ed = 0.0;
  for (int j = 0; j < 16; ++j) {
    for (int i = j; i < 16; ++i) ex = fmax(ex, fabs(out[i + 16 * j] - X[i + 16 * j]));
    ed = fmax(ed, fabs(out[256 + j] - d[j]) / fabs(d[j]));
  }
  unsigned long long h = 1469598103934665603ull;
  for (int i = 0; i < 272; ++i) {
    unsigned long long bits;
    memcpy(&bits, &out[i], 8);
    if (i < 256 && (i & 15) < (i >> 4)) continue;  // strict upper part: not defined
    h = (h ^ bits) * 1099511628211ull;
  }
  printf("diagonal block: %lld shader clocks (best of 200), max |X - X_ref| = %.2e, max rel pivot error = %.2e, result hash %016llx\n",
         t, ex, ed, h);
  return 0;
}
