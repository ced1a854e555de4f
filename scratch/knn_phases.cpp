// scratch micro-benchmark: time knn_dense phases by compiling the kernel file with phase switches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#ifndef KNN_SKIP_DIST
#define KNN_SKIP_DIST 0
#endif
#ifndef KNN_SKIP_SELECT
#define KNN_SKIP_SELECT 0
#endif
#include "../deep_gcns_torch_amd/csrc/knn_dense.hip"

int main(int argc, char** argv) {
  int B = 8, N = 4096, k = 16, d = argc > 1 ? atoi(argv[1]) : 1; int C = argc > 2 ? atoi(argv[2]) : 64;
  int K = k * d;
  std::vector<float> h((size_t)B * C * N);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX;
  float* x; int64_t* nn;
  hipMalloc(&x, h.size() * 4); hipMalloc(&nn, (size_t)B * N * k * 8);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) dgcn_knn_dense_f32(x, (int64_t)C * N, N, 1, B, C, N, K, d, nn, nullptr, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(a);
  int iters = 20;
  for (int it = 0; it < iters; ++it) dgcn_knn_dense_f32(x, (int64_t)C * N, N, 1, B, C, N, K, d, nn, nullptr, nullptr);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("skip_dist=%d skip_select=%d d=%d K=%d C=%d : %.3f ms/launch\n", KNN_SKIP_DIST, KNN_SKIP_SELECT, d, K, C, ms / iters);
  return 0;
}
