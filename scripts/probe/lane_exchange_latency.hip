// Probe: cycles per dependent / independent lane-exchange + FMA on one gfx950 wave (the
// building blocks of the 16 x 16 diagonal-block chain).
// hipcc -O3 -Wno-unused-value --offload-arch=gfx950 lane_exchange_latency.hip -o lane_probe && ./lane_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double dpp_b64(double v) {
  double old;
  asm volatile("" : "=v"(old));
  return __builtin_amdgcn_update_dpp(old, v, 0x150 | 3, 0xf, 0xf, false);
}
__device__ __forceinline__ double dpp_2x32(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x150 | 3, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 | 3, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double swz(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_swizzle((int)b, 0x10 | (3 << 5));
  const int hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), 0x10 | (3 << 5));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double bperm(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)b);
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double rdlane(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)b, 3);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 3);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void fmac_dpp(double& a, double b, double c) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c));
}

constexpr int N = 256;
// mode: which exchange; dep: 1 = each op consumes the previous result (latency), 0 = 8 independent chains (issue)
template <int MODE>
__global__ void k(double* out, long long* cyc, double seed) {
  double a[8];
  for (int j = 0; j < 8; ++j) a[j] = seed + threadIdx.x * 1e-3 + j;
  const double c = 1e-9;
  const long long w0 = wall_clock64();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) a[j] = fma(dpp_b64(a[j]), c, a[j]);
      if (MODE == 1) a[j] = fma(dpp_2x32(a[j]), c, a[j]);
      if (MODE == 2) a[j] = fma(swz(a[j]), c, a[j]);
      if (MODE == 3) a[j] = fma(bperm(a[j], (threadIdx.x + 16) & 63), c, a[j]);
      if (MODE == 4) a[j] = fma(rdlane(a[j]), c, a[j]);
      if (MODE == 5) fmac_dpp(a[j], a[j], c);
      if (MODE == 6) a[j] = fma(a[j], c, a[j]);
      // dependent variants: everything through a[0]
      if (MODE == 10) a[0] = fma(dpp_b64(a[0]), c, a[0]);
      if (MODE == 11) a[0] = fma(dpp_2x32(a[0]), c, a[0]);
      if (MODE == 12) a[0] = fma(swz(a[0]), c, a[0]);
      if (MODE == 13) a[0] = fma(bperm(a[0], (threadIdx.x + 16) & 63), c, a[0]);
      if (MODE == 14) a[0] = fma(rdlane(a[0]), c, a[0]);
      if (MODE == 15) fmac_dpp(a[0], a[0], c);
      if (MODE == 16) a[0] = fma(a[0], c, a[0]);
      if (MODE == 17) a[0] = __builtin_amdgcn_rcp(a[0]);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long w1 = wall_clock64();
  double s = 0;
  for (int j = 0; j < 8; ++j) s += a[j];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) {
    cyc[0] = t1 - t0;
    cyc[1] = w1 - w0;
  }
}

template <int MODE>
void run(const char* name, double* out, long long* cyc) {
  k<MODE><<<1, 64>>>(out, cyc, 1.0);
  k<MODE><<<1, 64>>>(out, cyc, 1.0);
  hipDeviceSynchronize();
  long long h[2];
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-44s %7.1f counter ticks = %6.1f ns per exchange+fma\n", name, (double)h[0] / (N * 8), (double)h[1] * 10.0 / (N * 8));
}

int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 64 * 8);
  hipMalloc(&cyc, 16);
  printf("(counter: __builtin_readcyclecounter; ns from wall_clock64 at 100 MHz)\n");
  run<6>("independent  fma only", out, cyc);
  run<0>("independent  v_mov_b64_dpp + fma", out, cyc);
  run<1>("independent  2 x v_mov_b32_dpp + fma", out, cyc);
  run<5>("independent  v_fmac_f64_dpp (+s_nop 1)", out, cyc);
  run<2>("independent  2 x ds_swizzle + fma", out, cyc);
  run<3>("independent  2 x ds_bpermute + fma", out, cyc);
  run<4>("independent  2 x v_readlane + fma", out, cyc);
  run<16>("dependent    fma only", out, cyc);
  run<17>("dependent    v_rcp_f64", out, cyc);
  run<10>("dependent    v_mov_b64_dpp + fma", out, cyc);
  run<11>("dependent    2 x v_mov_b32_dpp + fma", out, cyc);
  run<15>("dependent    v_fmac_f64_dpp (+s_nop 1)", out, cyc);
  run<12>("dependent    2 x ds_swizzle + fma", out, cyc);
  run<13>("dependent    2 x ds_bpermute + fma", out, cyc);
  run<14>("dependent    2 x v_readlane + fma", out, cyc);
  return 0;
}
