// Guard-page device allocator for the memory-safety tests (test infrastructure, not product code).
//
// torch.cuda.memory.CUDAPluggableAllocator entry points.  Every allocation gets its own physical block, mapped into a
// reserved virtual range with an UNMAPPED guard region on either side, and the user pointer is placed so that the buffer
// ENDS exactly (up to 16-byte alignment) at the end of the mapping (DGCN_GUARD_MODE=front: starts at its beginning).  A
// kernel that reads or writes past the end of ANY buffer it was handed -- input, output, saved-for-backward, workspace
// -- takes a page fault ("Memory access fault by GPU ... Reason: Page not present") instead of silently touching the
// neighbouring allocation, which is what torch's caching allocator gives it.  A free synchronises the device, unmaps and
// releases at once, so a use after free faults too.  Allocations made while a stream capture is in progress are never
// unmapped (the replayed graph keeps using them, as torch's private graph pools guarantee).
//
// Every allocation / free is appended to $DGCN_GUARD_LOG (default /tmp/dgcn_guard.log) as "A <ptr> <bytes> <map_lo>
// <map_hi>" / "F <ptr>", flushed line by line: tests/guard_alloc/lookup.py maps a fault address to the buffer whose guard
// was hit.
//
//   hipcc -O2 -fPIC -shared -o libguard_alloc.so guard_alloc.cpp

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {

struct Block {
  void* base;         // reserved range
  size_t reserved;    // bytes reserved (guard + mapping + guard)
  void* map;          // mapped range
  size_t mapped;
  hipMemGenericAllocationHandle_t handle;
  bool immortal;      // allocated during a stream capture: never unmapped
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_blocks;
size_t g_gran[16] = {0};
FILE* g_log = nullptr;
bool g_front = false;
bool g_plain = false;      // DGCN_GUARD_MODE=plain: one hipMalloc per allocation, no guards (A/B for the harness itself)
size_t g_align = 16;       // DGCN_GUARD_ALIGN: alignment of the user pointer in "back" mode
bool g_poison = false;
int g_free_mode = 1;       // DGCN_GUARD_FREE: 1 ("keepva", default) = unmap + release, the address range stays reserved and is
                           // never handed out again; 0 ("unmap") = the range is freed as well; 2 ("never") = nothing is
                           // unmapped.  Measured on ROCm 7.2 / MI355X: with "unmap" a range that is reserved and mapped again
                           // right after hipMemAddressFree serves stale translations (rocPRIM's radix sort and torch ops
                           // return wrong values within a few allocations; tests/guard_alloc/selftest.py), "keepva" is clean
bool g_init = false;
uint64_t g_allocs = 0, g_frees = 0;
int g_capture_depth = 0;   // > 0 while the test process captures a hipGraph (dgcn_guard_set_capturing; conftest wraps
                           // torch.cuda.graph): nothing may synchronise then, and what is allocated or freed stays mapped

void init_once() {
  if (g_init) return;
  g_init = true;
  const char* path = getenv("DGCN_GUARD_LOG");
  g_log = fopen(path ? path : "/tmp/dgcn_guard.log", "w");
  const char* mode = getenv("DGCN_GUARD_MODE");
  g_front = mode && strcmp(mode, "front") == 0;
  g_plain = mode && strcmp(mode, "plain") == 0;
  const char* al = getenv("DGCN_GUARD_ALIGN");
  if (al && atoi(al) >= 4) g_align = static_cast<size_t>(atoi(al));
  const char* fm = getenv("DGCN_GUARD_FREE");
  if (fm && strcmp(fm, "keepva") == 0) g_free_mode = 1;
  if (fm && strcmp(fm, "unmap") == 0) g_free_mode = 0;
  if (fm && strcmp(fm, "never") == 0) g_free_mode = 2;
  const char* poison = getenv("DGCN_GUARD_POISON");
  g_poison = poison && poison[0] == '1';
}

void die(const char* what, hipError_t e) {
  fprintf(stderr, "[guard_alloc] %s failed: %s\n", what, hipGetErrorString(e));
  fflush(stderr);
  abort();
}

#define GA_CHECK(call)                     \
  do {                                     \
    const hipError_t e_ = (call);          \
    if (e_ != hipSuccess) die(#call, e_);  \
  } while (0)

size_t granularity(int device) {
  if (device < 0 || device >= 16) device = 0;
  if (g_gran[device]) return g_gran[device];
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t g = 0;
  GA_CHECK(hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum));
  if (g < 4096) g = 4096;
  g_gran[device] = g;
  return g;
}

bool capturing(hipStream_t stream) {
  if (g_capture_depth > 0) return true;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

}  // namespace

extern "C" void* dgcn_guard_malloc(ssize_t size, int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  init_once();
  if (size <= 0) size = 1;
  int prev = 0;
  GA_CHECK(hipGetDevice(&prev));
  if (prev != device) GA_CHECK(hipSetDevice(device));
  if (g_plain) {
    void* p = nullptr;
    GA_CHECK(hipMalloc(&p, static_cast<size_t>(size)));
    Block pb;
    pb.base = nullptr; pb.reserved = 0; pb.map = p; pb.mapped = static_cast<size_t>(size); pb.immortal = capturing(stream);
    g_blocks[p] = pb;
    ++g_allocs;
    if (prev != device) GA_CHECK(hipSetDevice(prev));
    return p;
  }
  const size_t g = granularity(device);
  const size_t mapped = (static_cast<size_t>(size) + g - 1) / g * g;
  Block b;
  b.reserved = mapped + 2 * g;
  b.mapped = mapped;
  b.immortal = capturing(stream);
  GA_CHECK(hipMemAddressReserve(&b.base, b.reserved, g, nullptr, 0));
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  GA_CHECK(hipMemCreate(&b.handle, mapped, &prop, 0));
  b.map = static_cast<char*>(b.base) + g;
  GA_CHECK(hipMemMap(b.map, mapped, 0, b.handle, 0));
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  GA_CHECK(hipMemSetAccess(b.map, mapped, &acc, 1));
  char* user;
  if (g_front) {
    user = static_cast<char*>(b.map);
  } else {
    const size_t padded = (static_cast<size_t>(size) + g_align - 1) / g_align * g_align;
    user = static_cast<char*>(b.map) + mapped - padded;
  }
  if (g_poison && !b.immortal) {
    // 0xFF bytes = NaN floats / -1 ints: reads of never-written memory show up in the results
    GA_CHECK(hipMemsetAsync(b.map, 0xFF, mapped, stream));
  }
  g_blocks[user] = b;
  ++g_allocs;
  if (g_log) {
    fprintf(g_log, "A %p %zd %p %p%s\n", static_cast<void*>(user), size, b.map,
            static_cast<void*>(static_cast<char*>(b.map) + mapped), b.immortal ? " capture" : "");
    fflush(g_log);
  }
  if (prev != device) GA_CHECK(hipSetDevice(prev));
  return user;
}

extern "C" void dgcn_guard_free(void* ptr, ssize_t size, int device, hipStream_t stream) {
  (void)size;
  if (!ptr) return;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_blocks.find(ptr);
  if (it == g_blocks.end()) {
    fprintf(stderr, "[guard_alloc] free of unknown pointer %p\n", ptr);
    return;
  }
  Block& b = it->second;
  if (b.immortal || capturing(stream)) {
    b.immortal = true;       // a graph may replay kernels that use it: keep it mapped for the life of the process
    if (g_log) { fprintf(g_log, "K %p\n", ptr); fflush(g_log); }
    return;
  }
  int prev = 0;
  GA_CHECK(hipGetDevice(&prev));
  if (prev != device) GA_CHECK(hipSetDevice(device));
  GA_CHECK(hipDeviceSynchronize());     // kernels still queued on any stream may use the block
  if (g_plain) {
    GA_CHECK(hipFree(b.map));
    g_blocks.erase(it);
    ++g_frees;
    if (prev != device) GA_CHECK(hipSetDevice(prev));
    return;
  }
  if (g_free_mode < 2) {
    GA_CHECK(hipMemUnmap(b.map, b.mapped));
    GA_CHECK(hipMemRelease(b.handle));
    if (g_free_mode == 0) GA_CHECK(hipMemAddressFree(b.base, b.reserved));
  }
  if (g_log) { fprintf(g_log, "F %p\n", ptr); fflush(g_log); }
  g_blocks.erase(it);
  ++g_frees;
  if (prev != device) GA_CHECK(hipSetDevice(prev));
}

extern "C" void dgcn_guard_set_capturing(int on) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_capture_depth += on ? 1 : -1;
  if (g_capture_depth < 0) g_capture_depth = 0;
}

extern "C" void dgcn_guard_stats(uint64_t* allocs, uint64_t* frees, uint64_t* live) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (allocs) *allocs = g_allocs;
  if (frees) *frees = g_frees;
  if (live) *live = g_blocks.size();
}
