// Probe for look-ahead over a dense chain: can a latency-bound launch pair (1 workgroup with 133 KB of LDS, then 66
// workgroups) on one stream overlap a throughput-bound launch (2000 workgroups x 256 threads, 35 KB of LDS, ~45 us)
// on a second stream, level after level, with event dependencies between them?  Variants: plain second stream,
// second stream restricted to 224 of the 256 CUs (hipExtStreamCreateWithCUMask), serial reference.
// hipcc -O3 --offload-arch=gfx950 two_stream_overlap.hip -o two_stream_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_busy(double* out, int iters) {  // latency-bound stand-in: ~iters dependent FMAs
  extern __shared__ double lds[];
  double x = threadIdx.x * 1e-3;
  for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 1e-9);
  lds[threadIdx.x] = x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[blockDim.x - 1];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  double* out;
  hipMalloc(&out, 1 << 20);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_busy), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipStream_t s1, s2, s2m;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  std::vector<uint32_t> mask(8, 0xffffffffu);
  mask[7] = 0;  // 224 of 256 CUs
  hipError_t em = hipExtStreamCreateWithCUMask(&s2m, 8, mask.data());
  printf("CU-masked stream: %s\n", em == hipSuccess ? "ok" : hipGetErrorString(em));
  const int L = 28;
  std::vector<hipEvent_t> evM(L + 1), evD(L + 1);
  for (auto& e : evM) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (auto& e : evD) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  // calibrate: pivot-like 1 WG ~33 us, panel-like 66 WGs ~10 us, Schur-like 2000 WGs ~45 us
  const int itP = 2200, itC = 620, itD = 730;
  auto pivot = [&](hipStream_t s) { hipLaunchKernelGGL(k_busy, dim3(1), dim3(512), 133 * 1024, s, out, itP); };
  auto panel = [&](hipStream_t s) { hipLaunchKernelGGL(k_busy, dim3(66), dim3(512), 133 * 1024, s, out, itC); };
  auto schur = [&](hipStream_t s, int n) { hipLaunchKernelGGL(k_busy, dim3(n), dim3(256), 35 * 1024, s, out, itD); };
  auto timeit = [&](const char* name, auto&& f) {
    f();
    hipDeviceSynchronize();
    const double t0 = now();
    for (int r = 0; r < 5; ++r) f();
    hipDeviceSynchronize();
    printf("%-58s %8.1f us per level\n", name, (now() - t0) / 5 / L * 1e6);
  };
  timeit("pivot alone", [&] { for (int l = 0; l < L; ++l) pivot(s1); });
  timeit("panel alone", [&] { for (int l = 0; l < L; ++l) panel(s1); });
  timeit("Schur (2000 WGs) alone", [&] { for (int l = 0; l < L; ++l) schur(s1, 2000); });
  timeit("serial: pivot, panel, Schur on one stream", [&] {
    for (int l = 0; l < L; ++l) {
      pivot(s1);
      panel(s1);
      schur(s1, 2000);
    }
  });
  for (int v = 0; v < 2; ++v) {
    hipStream_t sd = v ? s2m : s2;
    if (v && em != hipSuccess) break;
    timeit(v ? "look-ahead, Schur rest on a 224-CU stream" : "look-ahead, Schur rest on a plain second stream", [&] {
      // level l: main: pivot, panel, [wait rest(l-1)], lead(l); side: [wait panel(l)] rest(l)
      for (int l = 0; l < L; ++l) {
        pivot(s1);
        panel(s1);
        hipEventRecord(evM[l], s1);
        if (l > 0) hipStreamWaitEvent(s1, evD[l - 1], 0);
        schur(s1, 130);  // lead tile columns
        hipStreamWaitEvent(sd, evM[l], 0);
        schur(sd, 1870);
        hipEventRecord(evD[l], sd);
      }
      hipStreamWaitEvent(s1, evD[L - 1], 0);
    });
  }
  return 0;
}
