// micro-benchmark 4: does staging output rows (per-wave bursts of RB consecutive rows) cut the write penalty?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// RB = consecutive rows per wave chunk; STAGE = 0: store each row as it completes, 1: stage RB rows in LDS, burst out
// WGSYNC = 1: the 4 waves of a workgroup own 4*RB consecutive rows and flush together after a barrier
template <int RB, int STAGE, int WGSYNC>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const int* __restrict__ rowptr,
                                         const int* __restrict__ col, int n_rows, int C, float* __restrict__ out) {
  constexpr int LPR = 32, G = 2, U = 4;
  __shared__ float4 stage[4][RB][32];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int g = lane / LPR, cl = lane % LPR;
  const int wave = blockIdx.x * 4 + wv;
  const int nw = gridDim.x * 4;
  const int n_chunks = (n_rows + RB - 1) / RB;
  const int n_iter = (n_chunks + nw - 1) / nw;
  for (int it = 0; it < n_iter; ++it) {
    const int chunk = it * nw + wave;            // consecutive waves -> consecutive chunks
    const int r0 = chunk * RB;
    for (int rr = 0; rr < RB; ++rr) {
      const int row = r0 + rr;
      if (row >= n_rows) break;
      const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
      const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
      float4 acc = {0, 0, 0, 0};
      for (int blk = beg; blk < end; blk += 64) {
        const int nb = min(64, end - blk);
        const int my = (lane < nb) ? col[blk + lane] : 0;
        for (int s0 = 0; s0 < nb; s0 += G * U) {
          float4 v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int ei = s0 + u * G + g;
            const int src = __shfl(my, ei & 63);
            v[u] = make_float4(0, 0, 0, 0);
            if (ei < nb) v[u] = *reinterpret_cast<const float4*>(x + (long long)src * C + cl * 4);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
      }
      acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
      if constexpr (STAGE) {
        if (g == 0) stage[wv][rr][cl] = acc;
      } else {
        if (g == 0) *reinterpret_cast<float4*>(out + (long long)row * C + cl * 4) = acc;
      }
    }
    if constexpr (STAGE) {
      if constexpr (WGSYNC) {
        __syncthreads();
        // 4*RB consecutive rows of this workgroup: 4*RB*32 float4, all 256 threads
        const int wr0 = (it * nw + blockIdx.x * 4) * RB;
        const float4* sp = &stage[0][0][0];
        for (int i = threadIdx.x; i < 4 * RB * 32; i += 256) {
          const int row = wr0 + i / 32;
          if (row < n_rows) *reinterpret_cast<float4*>(out + (long long)row * C + (i % 32) * 4) = sp[i];
        }
        __syncthreads();
      } else {
        const float4* sp = &stage[wv][0][0];
        for (int i = lane; i < RB * 32; i += 64) {
          const int row = r0 + i / 32;
          if (row < n_rows) *reinterpret_cast<float4*>(out + (long long)row * C + (i % 32) * 4) = sp[i];
        }
      }
    }
  }
}

template <int RB, int STAGE, int WGSYNC>
void run(const float* x, const int* rowptr, const int* col, int n, long long E, float* out, int grid) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float tot = 0;
  const int reps = 4;
  for (int i = 0; i < reps + 1; ++i) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<RB, STAGE, WGSYNC>), dim3(grid), dim3(256), 0, 0, x, rowptr, col, n, 128, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (i) tot += ms;
  }
  const float ms = tot / reps;
  printf("RB=%2d stage=%d wgsync=%d grid=%5d : %.3f ms  %.2f TB/s\n", RB, STAGE, WGSYNC, grid, ms, (E * 516.0 + n * 516.0) / ms / 1e9);
}

int main() {
  const int N = 2449029;
  std::vector<int> rp(N + 1);
  std::mt19937 rng(1); std::poisson_distribution<int> pd(51.5);
  rp[0] = 0; for (int i = 0; i < N; ++i) rp[i + 1] = rp[i] + pd(rng);
  const long long E = rp[N];
  std::vector<int> col(E);
  for (long long i = 0; i < E; ++i) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; col[i] = (int)(h % (unsigned long long)N); }
  float* x; int *drp, *dcol; float* out;
  CK(hipMalloc(&x, (size_t)N * 128 * 4)); CK(hipMemset(x, 0, (size_t)N * 128 * 4));
  CK(hipMalloc(&out, (size_t)N * 128 * 4));
  CK(hipMalloc(&drp, (N + 1) * 4)); CK(hipMalloc(&dcol, E * 4));
  CK(hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol, col.data(), E * 4, hipMemcpyHostToDevice));
  for (int grid : {8192, 2048}) {
    run<1, 0, 0>(x, drp, dcol, N, E, out, grid);
    run<8, 0, 0>(x, drp, dcol, N, E, out, grid);
    run<8, 1, 0>(x, drp, dcol, N, E, out, grid);
    run<16, 1, 0>(x, drp, dcol, N, E, out, grid);
    run<8, 1, 1>(x, drp, dcol, N, E, out, grid);
    run<16, 1, 1>(x, drp, dcol, N, E, out, grid);
  }
  return 0;
}
