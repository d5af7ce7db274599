// Probe: cost of handing 8.8 MB of matrix values from a caller-owned (pageable) array to the device:
// (a) memcpy into a pinned staging buffer + async copy, (b) hipMemcpyAsync straight from the pageable array
// (the runtime pins in place or stages itself), (c) array registered once with hipHostRegister,
// (d) staging in 1 MB chunks so that the host copy overlaps the DMA.
// hipcc -O3 --offload-arch=gfx950 h2d_paths.hip -o h2d_paths && ./h2d_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (size_t bytes : {(size_t)1 << 20, (size_t)8800000, (size_t)64 << 20}) {
    double* src = (double*)malloc(bytes);
    for (size_t i = 0; i < bytes / 8; ++i) src[i] = (double)i;
    void *pin, *dev;
    hipHostMalloc(&pin, bytes, hipHostMallocDefault);
    hipMalloc(&dev, bytes);
    const int reps = 30;
    auto run = [&](const char* name, auto&& f) {
      for (int w = 0; w < 3; ++w) f();
      double best = 1e9, sum = 0;
      for (int r = 0; r < reps; ++r) {
        src[r] += 1.0;
        const double t0 = now();
        f();
        const double dt = now() - t0;
        best = dt < best ? dt : best;
        sum += dt;
      }
      printf("%9zu B  %-34s best %8.1f us  mean %8.1f us  (%.1f GB/s)\n", bytes, name, best * 1e6, sum / reps * 1e6,
             bytes / best * 1e-9);
    };
    run("memcpy to pinned + async + sync", [&] {
      memcpy(pin, src, bytes);
      hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st);
      hipStreamSynchronize(st);
    });
    run("async from pageable + sync", [&] {
      hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, st);
      hipStreamSynchronize(st);
    });
    run("chunked staging (1 MB) + sync", [&] {
      const size_t ch = 1 << 20;
      for (size_t o = 0; o < bytes; o += ch) {
        const size_t len = bytes - o < ch ? bytes - o : ch;
        memcpy((char*)pin + o, (char*)src + o, len);
        hipMemcpyAsync((char*)dev + o, (char*)pin + o, len, hipMemcpyHostToDevice, st);
      }
      hipStreamSynchronize(st);
    });
    run("register + async + sync + unregister", [&] {
      hipHostRegister(src, bytes, hipHostRegisterDefault);
      hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, st);
      hipStreamSynchronize(st);
      hipHostUnregister(src);
    });
    hipHostRegister(src, bytes, hipHostRegisterDefault);
    run("registered once: async + sync", [&] {
      hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, st);
      hipStreamSynchronize(st);
    });
    hipHostUnregister(src);
    hipFree(dev);
    hipHostFree(pin);
    free(src);
  }
  return 0;
}
