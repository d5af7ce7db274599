// Probe: what does a kernel that exits at once cost inside a replayed hipGraph, and what does a
// host round trip (stream synchronise after a tiny kernel + pinned copy) cost?  Decides how the
// iterative-refinement loop is controlled (device-side early exit vs host check).
// hipcc -O3 --offload-arch=gfx950 graph_noop.hip -o graph_noop && ./graph_noop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_maybe(const int* flag, double* out, int n) {
  if (*flag) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] += 1.0;
}
__global__ void k_tiny(double* out) { out[threadIdx.x] += 1.0; }

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  int* flag;
  double* out;
  hipMalloc(&flag, 4);
  hipMalloc(&out, 1 << 24);
  hipMemset(out, 0, 1 << 24);
  double* pin;
  hipHostMalloc(&pin, 4096, hipHostMallocDefault);
  for (int grid : {1, 256, 2048}) {
    for (int nk : {1, 5, 10, 20, 40}) {
      for (int fl : {1, 0}) {
        hipMemcpy(flag, &fl, 4, hipMemcpyHostToDevice);
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(k_maybe, dim3(grid), dim3(256), 0, st, flag, out, 1 << 16);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int w = 0; w < 5; ++w) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        const int reps = 50;
        const double t0 = now();
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        const double dt = (now() - t0) / reps;
        printf("graph grid=%5d kernels=%2d exit=%d : %8.2f us per graph, %6.2f us per kernel\n", grid, nk, fl, dt * 1e6,
               dt * 1e6 / nk);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
      }
    }
  }
  // host round trip: tiny kernel + 16-byte D2H into pinned memory + stream synchronise
  {
    const int reps = 200;
    for (int w = 0; w < 10; ++w) {
      hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, out);
      hipMemcpyAsync(pin, out, 16, hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
    }
    double t0 = now();
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, out);
      hipMemcpyAsync(pin, out, 16, hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
    }
    printf("launch + 16 B D2H + sync        : %8.2f us\n", (now() - t0) / reps * 1e6);
    t0 = now();
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, out);
      hipStreamSynchronize(st);
    }
    printf("launch + sync                   : %8.2f us\n", (now() - t0) / reps * 1e6);
    // kernel writes the result straight into pinned host memory, host spins on it
    double* dpin;
    hipHostGetDevicePointer((void**)&dpin, pin, 0);
    t0 = now();
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, dpin);
      hipStreamSynchronize(st);
    }
    printf("launch (writes pinned) + sync   : %8.2f us\n", (now() - t0) / reps * 1e6);
  }
  return 0;
}
