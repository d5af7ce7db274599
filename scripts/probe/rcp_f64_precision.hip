// Probe: relative error of v_rcp_f64 (hardware estimate) and of the refinement variants on gfx950.
// hipcc -O3 --offload-arch=gfx950 rcp_f64_precision.hip -o rcp_probe && ./rcp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* d, double* e0, double* e_newton2, double* e_cubic, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = d[i];
  const double x0 = __builtin_amdgcn_rcp(v);
  double e = fma(-v, x0, 1.0);
  e0[i] = fabs(e);
  double x = fma(x0, e, x0);
  double e2 = fma(-v, x, 1.0);
  x = fma(x, e2, x);
  e_newton2[i] = fabs(x - 1.0 / v) / fabs(1.0 / v);
  const double t = fma(e, e, e);
  const double xc = fma(x0, t, x0);
  e_cubic[i] = fabs(xc - 1.0 / v) / fabs(1.0 / v);
}

int main() {
  const int n = 1 << 22;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;  // [1, 2)
    h[i] = ldexp(m, (int)(s % 41) - 20) * ((s & 1024) ? -1.0 : 1.0);
  }
  double *d, *a, *b, *c;
  hipMalloc(&d, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, a, b, c, n);
  std::vector<double> ha(n), hb(n), hc(n);
  hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) { m0 = fmax(m0, ha[i]); m1 = fmax(m1, hb[i]); m2 = fmax(m2, hc[i]); }
  printf("max |1 - d*rcp(d)| = %.3e (2^%.1f)\nmax rel err, two Newton steps = %.3e\nmax rel err, one cubic step = %.3e\n", m0, log2(m0), m1, m2);
  return 0;
}
