// Probe: lane layout of v_mfma_f64_16x16x4_f64 and the "accumulator as next B operand" identity.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D, double* D2) {
  // A: 16x8 (row-major A[i*8+k]), B: 8x16 (B[k*16+j]); D = A*B (16x16 row-major)
  int l = threadIdx.x, li = l & 15, lk = l >> 4;
  d4 acc = {0, 0, 0, 0};
  for (int s = 0; s < 2; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[li * 8 + 4 * s + lk], B[(4 * s + lk) * 16 + li], acc, 0, 0, 0);
  for (int q = 0; q < 4; ++q) D[(lk + 4 * q) * 16 + li] = acc[q];
  // second product: E = M * D with M = A2 (16x16, use A2[i][k] = i*0.5 - k), B operand taken from acc registers
  d4 acc2 = {0, 0, 0, 0};
  for (int s = 0; s < 4; ++s) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)(li * 0.5 - (4 * s + lk)), acc[s], acc2, 0, 0, 0);
  for (int q = 0; q < 4; ++q) D2[(lk + 4 * q) * 16 + li] = acc2[q];
}
int main() {
  double hA[128], hB[128], hD[256], hD2[256];
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 8; ++kk) hA[i * 8 + kk] = 1.0 + i * 0.37 - kk * 1.3 + (i * kk % 5);
  for (int kk = 0; kk < 8; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = 0.5 - j * 0.11 + kk * 0.7 + (j * 3 + kk) % 7;
  double *dA, *dB, *dD, *dD2;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dD2, sizeof hD2);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dD2);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hD2, dD2, sizeof hD2, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  double R[256];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 8; ++kk) s += hA[i * 8 + kk] * hB[kk * 16 + j]; R[i * 16 + j] = s; e1 = fmax(e1, fabs(s - hD[i * 16 + j])); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 16; ++kk) s += (i * 0.5 - kk) * R[kk * 16 + j]; e2 = fmax(e2, fabs(s - hD2[i * 16 + j])); }
  printf("layout err %.3e  chained err %.3e\n", e1, e2);
  return (e1 < 1e-9 && e2 < 1e-6) ? 0 : 1;
}
