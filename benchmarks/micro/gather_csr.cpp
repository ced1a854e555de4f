// micro-benchmark 3: row-structured (CSR) gather-sum, to find what costs the aggregation kernel 14% vs a flat gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE bits: 1 = write the output row, 2 = masked partial batches handled with per-load predicates (else clamp index),
//            4 = dynamic row assignment (atomic counter, chunks of 16 rows per wave)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const int* __restrict__ rowptr,
                                         const int* __restrict__ col, int n_rows, int C, float* __restrict__ out,
                                         int* __restrict__ counter) {
  constexpr int LPR = 32, G = 2, U = 4;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, cl = lane % LPR;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nw = gridDim.x * 4;
  float4 keep = {0, 0, 0, 0};
  int row = wave, row_end = n_rows, step = nw;
  if constexpr (MODE & 4) { row = 0; row_end = 0; step = 1; }
  while (true) {
    if constexpr (MODE & 4) {
      if (row >= row_end) {
        int r0 = 0;
        if (lane == 0) r0 = atomicAdd(counter, 16);
        r0 = __builtin_amdgcn_readfirstlane(r0);
        if (r0 >= n_rows) break;
        row = r0; row_end = min(n_rows, r0 + 16);
      }
    } else {
      if (row >= n_rows) break;
    }
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
          if constexpr (MODE & 2) {
            const int src = __shfl(my, ei & 63);
            v[u] = make_float4(0, 0, 0, 0);
            if (ei < nb) v[u] = *reinterpret_cast<const float4*>(x + (long long)src * C + cl * 4);
          } else {
            const int src = __shfl(my, min(ei, nb - 1) & 63);   // clamp: re-reads a valid row (cached), no predicate
            v[u] = *reinterpret_cast<const float4*>(x + (long long)src * C + cl * 4);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
    }
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    if constexpr (MODE & 1) {
      const long long orow = (MODE & 16) ? (row & 1023) : row;
      float4* dst = reinterpret_cast<float4*>(out + orow * C + cl * 4);
      if (g == 0) {
        if constexpr (MODE & 8) {
          __builtin_nontemporal_store(acc.x, &dst->x); __builtin_nontemporal_store(acc.y, &dst->y);
          __builtin_nontemporal_store(acc.z, &dst->z); __builtin_nontemporal_store(acc.w, &dst->w);
        } else {
          *dst = acc;
        }
      }
    } else {
      keep.x += acc.x + acc.y + acc.z + acc.w;
    }
    row += step;
  }
  if (keep.x == 123.456f) out[0] = keep.x;
}

template <int MODE>
void run(const float* x, const int* rowptr, const int* col, int n, long long E, float* out, int* counter, int grid) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float tot = 0;
  const int reps = 4;
  for (int i = 0; i < reps + 1; ++i) {
    CK(hipMemsetAsync(counter, 0, 4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, x, rowptr, col, n, 128, out, counter);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (i) tot += ms;
  }
  const float ms = tot / reps;
  printf("write=%d pred=%d dynamic=%d nt=%d small=%d grid=%5d : %.3f ms  %.2f TB/s\n", MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, (MODE >> 3) & 1, (MODE >> 4) & 1, grid, ms,
         (E * 516.0 + n * 516.0 * (MODE & 1)) / ms / 1e9);
}

int main() {
  const int N = 2449029;
  std::vector<int> rp(N + 1);
  std::mt19937 rng(1); std::poisson_distribution<int> pd(51.5);
  rp[0] = 0; for (int i = 0; i < N; ++i) rp[i + 1] = rp[i] + pd(rng);
  const long long E = rp[N];
  std::vector<int> col(E);
  for (long long i = 0; i < E; ++i) { unsigned long long h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; col[i] = (int)(h % (unsigned long long)N); }
  float* x; int *drp, *dcol, *counter; float* out;
  CK(hipMalloc(&x, (size_t)N * 128 * 4)); CK(hipMemset(x, 0, (size_t)N * 128 * 4));
  CK(hipMalloc(&out, (size_t)N * 128 * 4));
  CK(hipMalloc(&drp, (N + 1) * 4)); CK(hipMalloc(&dcol, E * 4)); CK(hipMalloc(&counter, 4));
  CK(hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol, col.data(), E * 4, hipMemcpyHostToDevice));
  printf("E = %lld\n", E);
  for (int grid : {8192, 2048}) {
    run<0>(x, drp, dcol, N, E, out, counter, grid);
    run<1>(x, drp, dcol, N, E, out, counter, grid);
    run<1 + 8>(x, drp, dcol, N, E, out, counter, grid);
    run<1 + 16>(x, drp, dcol, N, E, out, counter, grid);
    run<5>(x, drp, dcol, N, E, out, counter, grid);
    run<5 + 8>(x, drp, dcol, N, E, out, counter, grid);
  }
  return 0;
}
