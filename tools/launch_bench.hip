#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
  hipStream_t s; hipStreamCreate(&s);
  for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
  hipStreamSynchronize(s);
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    const int N = 2000;
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    printf("hipLaunchKernelGGL: %.2f us/launch host side (enqueue), %.2f us/launch incl. drain\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
  }
  return 0;
}
