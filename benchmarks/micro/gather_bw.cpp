// micro-benchmark: ceiling of random row gathers (row bytes 64..512) out of an N-row table, indices streamed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LPR, int U>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                     long long E, int C, float* __restrict__ out) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, cl = lane % LPR;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * 4;
  float4 acc = {0, 0, 0, 0};
  // each wave takes contiguous blocks of 64 indices
  for (long long blk = wave * 64; blk < E; blk += nw * 64) {
    const int my = (blk + lane < E) ? idx[blk + lane] : 0;
    for (int s0 = 0; s0 < 64; s0 += G * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = __shfl(my, (s0 + u * G + g) & 63);
        v[u] = *reinterpret_cast<const float4*>(x + (long long)src * C + cl * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

__global__ void fill_idx(int* idx, long long E, int N) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    idx[i] = (int)(h % (unsigned long long)N);
  }
}

template <int LPR, int U>
void run(const float* x, const int* idx, long long E, int C, float* out, int grid) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gather_kernel<LPR, U>), dim3(grid), dim3(256), 0, 0, x, idx, E, C, out);
  CK(hipEventRecord(a));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather_kernel<LPR, U>), dim3(grid), dim3(256), 0, 0, x, idx, E, C, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  printf("row %4d B  U=%d grid=%5d : %.3f ms  %.2f TB/s rows (+idx %.2f TB/s)\n", C * 4, U, grid, ms,
         E * (double)C * 4 / ms / 1e9, E * (double)(C * 4 + 4) / ms / 1e9);
}

int main() {
  const int N = 2449029; const long long E = 126167309;
  float* x; int* idx; float* out;
  CK(hipMalloc(&x, (size_t)N * 128 * 4)); CK(hipMemset(x, 0, (size_t)N * 128 * 4));
  CK(hipMalloc(&idx, E * 4)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(fill_idx, dim3(4096), dim3(256), 0, 0, idx, E, N);
  CK(hipDeviceSynchronize());
  for (int grid : {2048, 8192}) {
    run<4, 4>(x, idx, E, 16, out, grid);
    run<4, 8>(x, idx, E, 16, out, grid);
    run<8, 4>(x, idx, E, 32, out, grid);
    run<8, 8>(x, idx, E, 32, out, grid);
    run<16, 4>(x, idx, E, 64, out, grid);
    run<32, 4>(x, idx, E, 128, out, grid);
    run<32, 8>(x, idx, E, 128, out, grid);
  }
  return 0;
}
