// does a compute phase between load batches cost bandwidth at 6 waves/SIMD, and does double buffering get it back?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int WORK>
__device__ __forceinline__ void consume(float4 (&v)[4], float4& acc, float4& mx) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float a[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
    float* ac = &acc.x; float* m = &mx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = a[j];
#pragma unroll
      for (int k = 0; k < WORK; ++k) s = __builtin_amdgcn_exp2f(s * 0.1f - m[j]) + s;   // dependent chain
      m[j] = fmaxf(m[j], s * 1e-9f);
      ac[j] += s;
    }
  }
}

template <int WORK, bool DB>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const int* __restrict__ idx, long long E, int C,
                                         float* __restrict__ out) {
  extern __shared__ float lds[];
  constexpr int LPR = 32, G = 2, U = 4;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, cl = lane % LPR;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * 4;
  float4 acc = {0, 0, 0, 0}, mx = {0, 0, 0, 0};
  for (long long blk = wave * 64; blk < E; blk += nw * 64) {
    const int my = (blk + lane < E) ? idx[blk + lane] : 0;
    if constexpr (!DB) {
      for (int s0 = 0; s0 < 64; s0 += G * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int src = __shfl(my, (s0 + u * G + g) & 63);
          v[u] = *reinterpret_cast<const float4*>(x + (long long)src * C + cl * 4);
        }
        consume<WORK>(v, acc, mx);
      }
    } else {
      float4 vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = __shfl(my, (u * G + g) & 63);
        vb[u] = *reinterpret_cast<const float4*>(x + (long long)src * C + cl * 4);
      }
      for (int s0 = 0; s0 < 64; s0 += G * U) {
        float4 va[U];
#pragma unroll
        for (int u = 0; u < U; ++u) va[u] = vb[u];
        if (s0 + G * U < 64) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int src = __shfl(my, (s0 + G * U + u * G + g) & 63);
            vb[u] = *reinterpret_cast<const float4*>(x + (long long)src * C + cl * 4);
          }
        }
        consume<WORK>(va, acc, mx);
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w + mx.x == 123.456f) out[0] = acc.x + lds[threadIdx.x];
}

__global__ void fill_idx(int* idx, long long E, int N) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (long long)gridDim.x * blockDim.x) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    idx[i] = (int)(h % (unsigned long long)N);
  }
}

template <int WORK, bool DB>
void run(const float* x, const int* idx, long long E, float* out, int wg_per_cu) {
  const int C = 128;
  const size_t lds = wg_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wg_per_cu) - 512;
  CK(hipFuncSetAttribute((const void*)k<WORK, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = 8192;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<WORK, DB>), dim3(grid), dim3(256), lds, 0, x, idx, E, C, out);
  CK(hipEventRecord(a));
  const int reps = 4;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<WORK, DB>), dim3(grid), dim3(256), lds, 0, x, idx, E, C, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  printf("work=%d db=%d wg/cu=%d : %.3f ms  %.2f TB/s\n", WORK, (int)DB, wg_per_cu, ms, E * 516.0 / ms / 1e9);
}

int main() {
  const int N = 2449029; const long long E = 126167309;
  float* x; int* idx; float* out;
  CK(hipMalloc(&x, (size_t)N * 128 * 4)); CK(hipMemset(x, 0, (size_t)N * 128 * 4));
  CK(hipMalloc(&idx, E * 4)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(fill_idx, dim3(4096), dim3(256), 0, 0, idx, E, N);
  CK(hipDeviceSynchronize());
  for (int wg : {8, 6, 5, 4}) {
    run<0, false>(x, idx, E, out, wg);
    run<0, true>(x, idx, E, out, wg);
    run<2, false>(x, idx, E, out, wg);
    run<2, true>(x, idx, E, out, wg);
    run<4, false>(x, idx, E, out, wg);
    run<4, true>(x, idx, E, out, wg);
  }
  return 0;
}
