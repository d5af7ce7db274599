// Probe: issue cost (one wave alone on its SIMD) of independent fp64 vector instructions: plain FMA, FMA with a
// DPP row broadcast on the first source, 64-bit DPP move, v_cndmask pair, v_rcp_f64; and the latency of dependent ones.
// hipcc -O3 --offload-arch=gfx950 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
__global__ __launch_bounds__(64) void k(long long* out, double* sink) {
  double a0 = threadIdx.x, a1 = 1.5, a2 = 2.5, a3 = 3.5, a4 = 4.5, a5 = 5.5, a6 = 6.5, a7 = 7.5, m = 1e-3;
  long long t[8];
  t[0] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; ++i)
    asm volatile(REP8("v_fmac_f64 %0, %8, %8\n\tv_fmac_f64 %1, %8, %8\n\tv_fmac_f64 %2, %8, %8\n\tv_fmac_f64 %3, %8, %8\n\t"
                      "v_fmac_f64 %4, %8, %8\n\tv_fmac_f64 %5, %8, %8\n\tv_fmac_f64 %6, %8, %8\n\tv_fmac_f64 %7, %8, %8\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  t[1] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; ++i)
    asm volatile(REP8("v_fmac_f64_dpp %0, %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %2, %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %4, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                      "v_fmac_f64_dpp %6, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  t[2] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; ++i)  // dependent plain FMAs
    asm volatile(REP8(REP8("v_fmac_f64 %0, %1, %1\n\t")) : "+v"(a0) : "v"(m));
  t[3] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; ++i)  // dependent DPP FMAs
    asm volatile(REP8(REP8("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t")) : "+v"(a1) : "v"(m));
  t[4] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; ++i)  // independent rcp
    asm volatile(REP8("v_rcp_f64 %0, %8\n\tv_rcp_f64 %1, %8\n\tv_rcp_f64 %2, %8\n\tv_rcp_f64 %3, %8\n\tv_rcp_f64 %4, %8\n\tv_rcp_f64 %5, %8\n\tv_rcp_f64 %6, %8\n\tv_rcp_f64 %7, %8\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  t[5] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; ++i)  // dependent rcp
    asm volatile(REP8(REP8("v_rcp_f64 %0, %0\n\ts_nop 0\n\t")) : "+v"(a2));
  t[6] = __builtin_readcyclecounter();
  for (int i = 0; i < 16; ++i)  // independent 32-bit moves (cndmask-like cost)
    asm volatile(REP8("v_mov_b64 %0, %8\n\tv_mov_b64 %1, %8\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %8\n\tv_mov_b64 %4, %8\n\tv_mov_b64 %5, %8\n\tv_mov_b64 %6, %8\n\tv_mov_b64 %7, %8\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  t[7] = __builtin_readcyclecounter();
  if (threadIdx.x == 0)
    for (int i = 0; i < 7; ++i) out[i] = t[i + 1] - t[i];
  sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  long long* d;
  double* s;
  hipMalloc(&d, 64);
  hipMalloc(&s, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
  long long h[7];
  hipMemcpy(h, d, 56, hipMemcpyDeviceToHost);
  const char* names[7] = {"independent v_fmac_f64", "independent v_fmac_f64_dpp", "dependent v_fmac_f64", "dependent v_fmac_f64_dpp (+s_nop 1)",
                          "independent v_rcp_f64", "dependent v_rcp_f64 (+s_nop 0)", "independent v_mov_b64"};
  for (int i = 0; i < 7; ++i) printf("%-40s %7.2f counter ticks per instruction\n", names[i], h[i] / 1024.0);
  return 0;
}
