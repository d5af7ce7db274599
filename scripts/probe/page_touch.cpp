// First-touch cost of a large host array: malloc vs mmap + MADV_HUGEPAGE vs MAP_POPULATE (what the product lists of a
// cold symbolic analysis pay).  g++ -O2 -pthread page_touch.cpp -o page_touch
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void touch(char* p, size_t n, int nt) {
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=] {
      const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
      for (size_t i = lo; i < hi; i += 4096) p[i] = 1;
    });
  for (auto& x : th) x.join();
}
int main() {
  const size_t n = 160u << 20;
  for (int nt : {1, 8}) {
    double t0 = now();
    char* a = (char*)malloc(n);
    touch(a, n, nt);
    double t1 = now();
    free(a);
    char* b = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(b, n, MADV_HUGEPAGE);
    double t2 = now();
    touch(b, n, nt);
    double t3 = now();
    munmap(b, n);
    double t4 = now();
    char* c = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    double t5 = now();
    munmap(c, n);
    printf("threads %d: malloc + touch %.1f ms | mmap + MADV_HUGEPAGE + touch %.1f ms | MAP_POPULATE %.1f ms\n", nt, 1e3 * (t1 - t0),
           1e3 * (t3 - t2), 1e3 * (t5 - t4));
  }
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char buf[128] = {0};
  if (f) { fgets(buf, 127, f); fclose(f); }
  printf("THP: %s", buf);
  return 0;
}
