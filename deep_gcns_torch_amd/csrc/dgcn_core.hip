// Library-level entry points: version, error strings, and the runtime self-test.
#include "dgcn_common.h"

namespace {

__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    y[i] = fmaf(a, x[i], y[i]);
  }
}

}  // namespace

extern "C" int dgcn_version(void) { return DGCN_VERSION; }

extern "C" const char* dgcn_strerror(int rc) {
  switch (rc) {
    case DGCN_OK: return "ok";
    case DGCN_E_NULL: return "dgcn: required pointer is NULL";
    case DGCN_E_SHAPE: return "dgcn: size/shape out of the supported range";
    case DGCN_E_ALIGN: return "dgcn: pointer or stride not aligned as required";
    case DGCN_E_MODE: return "dgcn: unknown mode or flag combination";
    case DGCN_E_WORKSPACE: return "dgcn: workspace too small";
    default: break;
  }
  if (rc > 0) return hipGetErrorString(static_cast<hipError_t>(rc));
  return "dgcn: unknown error";
}

extern "C" int dgcn_selftest_axpy_f32(float a, const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y) return DGCN_E_NULL;
  if (n < 0) return DGCN_E_SHAPE;
  if (n == 0) return DGCN_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(axpy_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a, x, y, n);
  return dgcn::launch_status();
}
