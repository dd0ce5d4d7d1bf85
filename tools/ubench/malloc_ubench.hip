// hipMalloc / hipFree cost by size, with and without other allocations alive (round 4: arenas that grow during the C5 rounds)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  void* warm; hipMalloc(&warm, 1 << 20);
  for (int pass = 0; pass < 2; pass++) {
    std::vector<void*> held;
    if (pass == 1) for (int k = 0; k < 16; k++) { void* p; hipMalloc(&p, (size_t)512 << 20); hipMemset(p, 0, (size_t)512 << 20); held.push_back(p); }
    hipDeviceSynchronize();
    printf(pass ? "with 8 GB held:\n" : "fresh process:\n");
    for (size_t mb : {1, 16, 64, 256, 1024, 2048}) {
      double t0 = now(); void* p = nullptr; hipError_t e = hipMalloc(&p, mb << 20); double t1 = now();
      hipMemsetAsync(p, 0, 4096, 0); hipDeviceSynchronize(); double t2 = now();
      hipFree(p); double t3 = now();
      printf("  %5zu MB: malloc %.3f ms (%s), first touch %.3f ms, free %.3f ms\n", mb, 1e3 * (t1 - t0), hipGetErrorString(e), 1e3 * (t2 - t1), 1e3 * (t3 - t2));
    }
    for (void* p : held) hipFree(p);
  }
  return 0;
}
