// Probe: latency of a posted value (agent-scope atomic store, polled with agent-scope atomic loads - the exchange
// of the single-launch solve and of the pivot -> panel hop) between two workgroups on the SAME XCD and on
// DIFFERENT XCDs, and the time to pull 100 KB that the other workgroup has just written (release / acquire at
// agent scope, as between a child's Schur items and its parent's pivot item).  Which XCD a workgroup runs on is
// read from the hardware (XCC_ID), not assumed.
// hipcc -O3 --offload-arch=gfx950 xcd_hop.hip -o xcd_hop && ./xcd_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// ping-pong: block a posts i into slot A, block b answers in slot B; nhops round trips
__global__ void k_pingpong(unsigned long long* slots, int a, int b, int nhops, long long* ticks, unsigned* xcc) {
  const int me = blockIdx.x;
  if (threadIdx.x == 0) xcc[me] = xcc_id();
  if (me != a && me != b) return;
  if (threadIdx.x != 0) return;
  unsigned long long* mine = slots + (me == a ? 0 : 16);
  unsigned long long* other = slots + (me == a ? 16 : 0);
  const long long t0 = wall_clock64();
  for (int i = 1; i <= nhops; ++i) {
    if (me == a) {
      __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)i) {}
    } else {
      while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)i) {}
      __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (me == a) *ticks = wall_clock64() - t0;
}

// bulk: block a writes n doubles (plain stores), releases, sets a flag; block b polls the flag, acquires, sums
__global__ void k_bulk(double* buf, int n, int* flag, int a, int b, int rep, long long* ticks, double* sink) {
  const int me = blockIdx.x;
  if (me != a && me != b) return;
  if (me == a) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = (double)(i + rep);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(flag, rep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    __shared__ long long t0s;
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rep + 1) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      t0s = wall_clock64();
    }
    __syncthreads();
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += buf[i];
    sink[threadIdx.x] = s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *ticks = wall_clock64() - t0s;
  }
}

int main() {
  const int NB = 64;
  unsigned long long* slots;
  long long* ticks;
  unsigned* xcc;
  double *buf, *sink;
  int* flag;
  hipMalloc(&slots, 4096);
  hipMalloc(&ticks, 8);
  hipMalloc(&xcc, NB * 4);
  hipMalloc(&buf, 1 << 20);
  hipMalloc(&sink, 8192);
  hipMalloc(&flag, 4);
  std::vector<unsigned> hx(NB);
  hipMemset(slots, 0, 4096);
  hipLaunchKernelGGL(k_pingpong, dim3(NB), dim3(64), 0, 0, slots, 0, 1, 1, ticks, xcc);
  hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %u", hx[i]);
  printf("\n");
  int same = -1, diff = -1;
  for (int i = 1; i < NB && (same < 0 || diff < 0); ++i) {
    if (hx[i] == hx[0] && same < 0) same = i;
    if (hx[i] != hx[0] && diff < 0) diff = i;
  }
  printf("partner on the same XCD: workgroup %d, on another XCD: workgroup %d\n", same, diff);
  for (int pass = 0; pass < 2; ++pass)
    for (int b : {same, diff}) {
      if (b < 0) continue;
      const int nh = 2000;
      hipMemset(slots, 0, 4096);
      hipLaunchKernelGGL(k_pingpong, dim3(NB), dim3(64), 0, 0, slots, 0, b, nh, ticks, xcc);
      long long t;
      hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      hipMemcpy(hx.data(), xcc, NB * 4, hipMemcpyDeviceToHost);
      printf("posted value, workgroup 0 (XCD %u) <-> %2d (XCD %u): %.3f us per one-way hop\n", hx[0], b, hx[b],
             t / 100.0 / nh / 2);
    }
  for (int n : {12800, 131072})
    for (int b : {same, diff}) {
      if (b < 0) continue;
      double best = 1e9;
      for (int rep = 0; rep < 20; ++rep) {
        hipMemset(flag, 0, 4);
        hipLaunchKernelGGL(k_bulk, dim3(NB), dim3(512), 0, 0, buf, n, flag, 0, b, 0, ticks, sink);
        long long t;
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        if (t / 100.0 < best) best = t / 100.0;
      }
      printf("%7zu bytes written by workgroup 0, read by %2d (%s XCD) behind release / acquire: %.2f us (%.1f GB/s)\n",
             (size_t)n * 8, b, b == same ? "same" : "other", best, n * 8 / best * 1e-3);
    }
  return 0;
}
