// Small-message SUM all-reduce between the ranks of one node, device side (include/unet_hip.h: unet_comm_*).
//
// The reference is single-process (no tf.distribute / Horovod anywhere): data parallelism is new, and what it adds to the step's critical path are
// 17 tiny fp64 reductions (8 BatchNorm statistics + the loss sums forward, 8 BatchNorm-backward sums; <= 2 x 1024 doubles each) that the next op reads.
// Through a collective library each costs a host call, a proxy hand-off and a ring / tree walk: tens of microseconds for 16 KB.  Here every rank owns a
// receive area in fine-grained (uncached) HBM that its peers map through HIP IPC; ONE kernel on the compute stream
//   (1) pushes its values into every peer's area -- xGMI is point to point: W - 1 independent links carry W - 1 copies at once, no ring --
//   (2) polls its own area until every rank's values of THIS call have landed, (3) adds them in rank order (every rank gets the same bits).
// A value travels as two 8-byte words (sequence number << 32 | half of the double): an aligned 8-byte store is a single-copy atomic on the fabric, so a word whose
// sequence number matches is complete -- no flag behind the data, no fence, one fabric latency per all-reduce.  Two areas alternate by call parity: a
// rank can be at most one call ahead of a peer (finishing call k needs the peer's words of call k, which it sends after finishing k - 1).
// The poll is bounded (timeout_ms): a missing peer sets the communicator's error word and turns the result into NaN instead of hanging the GPU; unet_comm_status reads the word.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "common.h"

namespace {

constexpr int MAXW = UNET_COMM_MAX_WORLD, MAXD = UNET_COMM_MAX_DOUBLES;
constexpr size_t AREA_WORDS = (size_t)MAXW * 2 * MAXD;          // one parity: [source rank][2 * MAXD] words
constexpr size_t AREA_BYTES = 2 * AREA_WORDS * 8;

struct peer_ptrs { unsigned long long* p[MAXW]; };

__device__ __forceinline__ void put(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long get(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(256) void small_allreduce_kernel(double* __restrict__ buf, int n, peer_ptrs peers, int rank, int world, unsigned seq,
                                                               long long timeout_ticks, int* __restrict__ err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t par = (size_t)(seq & 1) * AREA_WORDS;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(buf[i]), tag = (unsigned long long)seq << 32;
  const unsigned long long w0 = tag | (bits & 0xFFFFFFFFull), w1 = tag | (bits >> 32);
  for (int k = 0; k < world; ++k) {                          // own area last: the remote words are on their way while the local ones are written
    const int p = (rank + 1 + k) % world;
    unsigned long long* dst = peers.p[p] + par + (size_t)rank * 2 * MAXD + 2 * i;
    put(dst, w0); put(dst + 1, w1);
  }
  const unsigned long long* mine = peers.p[rank] + par + 2 * i;
  const long long t0 = wall_clock64();
  double sum = 0.0;
  for (int s = 0; s < world; ++s) {
    unsigned long long a, b;
    int spins = 0;
    for (;;) {
      a = get(mine + (size_t)s * 2 * MAXD); b = get(mine + (size_t)s * 2 * MAXD + 1);
      if ((unsigned)(a >> 32) == seq && (unsigned)(b >> 32) == seq) break;
      if ((++spins & 63) == 0) {
        if (wall_clock64() - t0 > timeout_ticks || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          // a missing rank: latch 1 + its number AND poison the result -- every value that would have been a cross-rank sum becomes NaN, so a loss / statistic built on
          // rank-local sums cannot pass for a global one even where nobody reads unet_comm_status (keras_like.check_comm does, once per epoch / evaluate call)
          atomicExch(err, 1 + s); buf[i] = __longlong_as_double(0x7FF8000000000000ll); return;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    sum += __longlong_as_double((long long)((a & 0xFFFFFFFFull) | (b << 32)));
  }
  buf[i] = sum;
}

}  // namespace

struct unet_comm {
  unet_ctx* ctx = nullptr;
  int device = 0;                            // (kept here: destroy may run after the context is gone)
  int rank = 0, world = 1;
  unsigned seq = 0;
  int timeout_ms = 60000;
  unsigned long long* area = nullptr;        // own receive area (fine-grained)
  peer_ptrs peers{};                         // [rank] = area, the others IPC-mapped
  bool mapped[MAXW] = {};
  int* err = nullptr;                        // device word: 0, or 1 + the source rank whose words did not arrive in time
  bool connected = false;
};

extern "C" {

int32_t unet_comm_create(unet_ctx* ctx, int32_t rank, int32_t world, unet_comm** out, unsigned char* handle_out) {
  if (!ctx || !out || !handle_out) return UNET_E_ARG;
  if (world < 1 || world > MAXW || rank < 0 || rank >= world) UNET_FAIL(ctx, UNET_E_ARG, "unet_comm_create: rank %d of %d (at most %d ranks: one node)", rank, world, MAXW);
  static_assert(sizeof(hipIpcMemHandle_t) <= UNET_COMM_HANDLE_BYTES, "handle size");
  (void)hipSetDevice(ctx->device);
  unet_comm* c = new unet_comm;
  c->ctx = ctx; c->device = ctx->device; c->rank = rank; c->world = world;
  // uncached: a peer's store must be visible to the polling loads of a RUNNING kernel (plain device memory is only coherent at kernel boundaries)
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->area), AREA_BYTES, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->area), AREA_BYTES, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) { delete c; UNET_FAIL(ctx, UNET_E_HIP, "unet_comm_create: fine-grained allocation: %s", hipGetErrorString(e)); }
  e = hipMalloc(reinterpret_cast<void**>(&c->err), sizeof(int));
  if (e == hipSuccess) e = hipMemset(c->area, 0, AREA_BYTES);          // sequence numbers start at 1: a zeroed word never matches
  if (e == hipSuccess) e = hipMemset(c->err, 0, sizeof(int));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, c->area);
  if (e != hipSuccess) { (void)hipFree(c->area); (void)hipFree(c->err); delete c; UNET_FAIL(ctx, UNET_E_HIP, "unet_comm_create: %s", hipGetErrorString(e)); }
  memset(handle_out, 0, UNET_COMM_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  c->peers.p[rank] = c->area;
  *out = c;
  return UNET_OK;
}

int32_t unet_comm_connect(unet_comm* c, const unsigned char* handles) {
  if (!c || !handles) return UNET_E_ARG;
  if (c->connected) UNET_FAIL(c->ctx, UNET_E_STATE, "unet_comm_connect: already connected");
  (void)hipSetDevice(c->ctx->device);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * UNET_COMM_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); UNET_FAIL(c->ctx, UNET_E_HIP, "unet_comm_connect: mapping rank %d's area: %s", r, hipGetErrorString(e)); }
    c->peers.p[r] = static_cast<unsigned long long*>(p); c->mapped[r] = true;
  }
  c->connected = true;
  return UNET_OK;
}

int32_t unet_comm_set_timeout_ms(unet_comm* c, int32_t ms) {
  if (!c || ms < 1) return UNET_E_ARG;
  c->timeout_ms = ms;
  return UNET_OK;
}

int32_t unet_comm_allreduce_f64(unet_comm* c, double* buf, int32_t count, void* stream) {
  if (!c || !buf) return UNET_E_ARG;
  if (!c->connected) UNET_FAIL(c->ctx, UNET_E_STATE, "unet_comm_allreduce_f64: not connected");
  if (count < 1) UNET_FAIL(c->ctx, UNET_E_ARG, "unet_comm_allreduce_f64: %d doubles", count);
  const long long ticks = (long long)c->timeout_ms * 100000;          // wall_clock64: 100 MHz
  for (int off = 0; off < count; off += MAXD) {              // (longer vectors: one call per UNET_COMM_MAX_DOUBLES, each with its own sequence number)
    const int n = count - off < MAXD ? count - off : MAXD;
    if (++c->seq == 0) c->seq = 2;                           // (a wrap after 2^32 calls: 0 stays the never-written value, the parities keep alternating)
    hipLaunchKernelGGL(small_allreduce_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), buf + off, n, c->peers, c->rank, c->world, c->seq,
                       ticks, c->err);
  }
  UNET_CHECK_LAUNCH(c->ctx, "small_allreduce");
  return UNET_OK;
}

/* err_out: 0, or 1 + the rank whose contribution did not arrive within the timeout (sticky).  Synchronises the device copy of one word on `stream`. */
int32_t unet_comm_status(unet_comm* c, int32_t* err_out, void* stream) {
  if (!c || !err_out) return UNET_E_ARG;
  int v = 0;
  hipError_t e = hipMemcpyAsync(&v, c->err, sizeof(int), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
  if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (e != hipSuccess) UNET_FAIL(c->ctx, UNET_E_HIP, "unet_comm_status: %s", hipGetErrorString(e));
  *err_out = v;
  return UNET_OK;
}

void unet_comm_destroy(unet_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (c->mapped[r]) (void)hipIpcCloseMemHandle(c->peers.p[r]);
  (void)hipFree(c->area); (void)hipFree(c->err);
  delete c;
}

}  // extern "C"
