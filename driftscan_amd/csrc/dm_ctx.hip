// dm_ctx.hip — context, workspace arena and host<->device staging for libdriftmi.
#include "dm_common.h"

#include <algorithm>
#include <initializer_list>
#include "dm_kernels.h"
#include "../../include/driftmi.h"

namespace {
constexpr size_t kAlign = 256;
constexpr size_t kPinned = 64u << 20;  // staging ring for descriptor uploads
inline size_t align_up(size_t x) { return (x + kAlign - 1) & ~(kAlign - 1); }
}  // namespace

// Replace the arena by one of at least `bytes` (preferably `prefer`).  Live allocations of the old arena
// stay valid: it is retired, not freed, until the outermost mark is released.
static int ws_grow(dm_ctx* ctx, size_t prefer, size_t bytes) {
  // growing: kernels in flight may still use the old arena -> drain first
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->ws_used == 0 && ctx->ws) {
    (void)hipFree(ctx->ws);
    ctx->ws = nullptr;
    ctx->ws_cap = 0;
  }
  void* p = nullptr;
  hipError_t e = hipErrorOutOfMemory;
  size_t got = 0;
  // the preferred size doubles the arena (few growth steps); at hundreds of GB that may not exist while the
  // old arena is still alive, so fall back to what is needed now
  for (size_t want : {prefer, bytes + bytes / 8, bytes}) {
    want = align_up(want);
    if (want < bytes) continue;
    e = hipMalloc(&p, want);
    if (e == hipSuccess) { got = want; break; }
    (void)hipGetLastError();
    p = nullptr;
  }
  if (e != hipSuccess) {
    ctx->err = std::string("hipMalloc workspace: ") + hipGetErrorString(e);
    return DM_ENOMEM;  // the old arena (if any) is untouched
  }
  if (ctx->ws) ctx->retired.push_back(ctx->ws);  // live allocations remain valid
  ctx->ws = reinterpret_cast<char*>(p);
  ctx->ws_cap = got;
  ctx->ws_used = 0;
  return DM_OK;
}

int dm_ws_reserve(dm_ctx* ctx, size_t bytes) {
  bytes = align_up(bytes);
  if (bytes <= ctx->ws_cap) return DM_OK;
  return ws_grow(ctx, bytes, bytes);
}

void* dm_ws_alloc(dm_ctx* ctx, size_t bytes) {
  bytes = align_up(bytes ? bytes : 1);
  if (ctx->ws_used + bytes > ctx->ws_cap) {
    // the new arena starts empty (earlier allocations live on in the retired one): it has to hold this
    // request, and preferably as much again as everything handed out so far
    size_t prefer = std::max(ctx->ws_cap * 2, (ctx->ws_used + bytes) * 2);
    if (prefer < (256u << 20)) prefer = 256u << 20;
    if (ws_grow(ctx, prefer, bytes) != DM_OK) return nullptr;
  }
  void* p = ctx->ws + ctx->ws_used;
  ctx->ws_used += bytes;
  return p;
}

size_t dm_ws_mark(dm_ctx* ctx) { return ctx->ws_used; }

void dm_ws_release(dm_ctx* ctx, size_t mark) {
  // Re-use is stream-ordered: every producer and consumer of workspace memory,
  // including the H2D descriptor copies, runs on ctx->stream.
  if (mark <= ctx->ws_used) ctx->ws_used = mark;
  if (!ctx->retired.empty() && mark == 0) {
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : ctx->retired) (void)hipFree(p);
    ctx->retired.clear();
  }
}

int dm_upload(dm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return DM_OK;
  if (bytes > kPinned / 2) {  // big: synchronous path
    DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DM_HIP(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return DM_OK;
  }
  size_t need = align_up(bytes);
  if (ctx->hpin_used + need > ctx->hpin_cap) {
    // wrap the ring: make sure every earlier staged copy has been consumed
    DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->hpin_used = 0;
  }
  char* stage = ctx->hpin + ctx->hpin_used;
  ctx->hpin_used += need;
  std::memcpy(stage, src, bytes);
  DM_HIP(ctx, hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
  return DM_OK;
}

// Device -> host through a page-locked landing buffer: the callers hand in pageable memory (std::vector), for which
// the runtime stages small copies itself at several hundred microseconds apiece — and every one of these calls sits
// on the critical path (ranks, eigenvalues, status words the host decides on).
int dm_download(dm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return DM_OK;
  constexpr size_t kLand = 8u << 20;
  if (!ctx->hpin_dl) {
    void* hp = nullptr;
    if (hipHostMalloc(&hp, kLand, hipHostMallocDefault) == hipSuccess) ctx->hpin_dl = reinterpret_cast<char*>(hp);
    else (void)hipGetLastError();
  }
  if (!ctx->hpin_dl) {
    DM_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DM_OK;
  }
  for (size_t off = 0; off < bytes; off += kLand) {
    const size_t n = std::min(kLand, bytes - off);
    DM_HIP(ctx, hipMemcpyAsync(ctx->hpin_dl, static_cast<const char*>(src) + off, n, hipMemcpyDeviceToHost, ctx->stream));
    DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(static_cast<char*>(dst) + off, ctx->hpin_dl, n);
  }
  return DM_OK;
}

hipEvent_t dm_prof_event(dm_ctx* ctx) {
  if (!ctx->ev_pool.empty()) {
    hipEvent_t e = ctx->ev_pool.back();
    ctx->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

extern "C" {

int dm_prof_reset(dm_ctx* ctx, int enable) {
  if (!ctx) return DM_EARG;
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->prof) { ctx->ev_pool.push_back(r.a); ctx->ev_pool.push_back(r.b); }
  ctx->prof.clear();
  for (auto& q : ctx->prof_seq) q = 0;
  if (!ctx->prof_dev) DM_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->prof_dev), sizeof(unsigned long long) * DM_PROF_NCLASS));
  DM_HIP(ctx, hipMemset(ctx->prof_dev, 0, sizeof(unsigned long long) * DM_PROF_NCLASS));
  ctx->prof_on = enable != 0;
  ctx->prof_level = enable >= 2 ? 2 : 1;
  return DM_OK;
}

// ms[c], flops[c], launches[c] for c < DM_PROF_NCLASS (= 24): summed event time, algorithmic
// flops and launch count of each kernel class since dm_prof_reset.
int dm_prof_trd_stride(void) { return DM_PROF_TRD_STRIDE; }

int dm_prof_report(dm_ctx* ctx, double* ms, double* flops, long long* launches) {
  if (!ctx || !ms || !flops || !launches) return DM_EARG;
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int c = 0; c < DM_PROF_NCLASS; ++c) { ms[c] = 0.0; flops[c] = 0.0; launches[c] = 0; }
  for (auto& r : ctx->prof) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) ms[r.cls] += r.weight * t;
    flops[r.cls] += r.weight * r.flops;
    launches[r.cls] += (long long)r.weight;
  }
  if (ctx->prof_dev) {
    unsigned long long h[DM_PROF_NCLASS];
    DM_HIP(ctx, hipMemcpy(h, ctx->prof_dev, sizeof(h), hipMemcpyDeviceToHost));
    for (int c = 0; c < DM_PROF_NCLASS; ++c) flops[c] += (double)h[c];
  }
  return DM_OK;
}

int dm_ctx_create(int device, size_t workspace_bytes, void* stream, dm_ctx** out) {
  if (!out) return DM_EARG;
  *out = nullptr;
  dm_ctx* ctx = new dm_ctx();
  ctx->device = device;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    delete ctx;
    return DM_EHIP;
  }
  if (stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    ctx->own_stream = false;
  } else {
    e = hipStreamCreate(&ctx->stream);
    if (e != hipSuccess) {
      delete ctx;
      return DM_EHIP;
    }
    ctx->own_stream = true;
  }
  void* hp = nullptr;
  e = hipHostMalloc(&hp, kPinned, hipHostMallocDefault);
  if (e != hipSuccess) {
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return DM_ENOMEM;
  }
  ctx->hpin = reinterpret_cast<char*>(hp);
  ctx->hpin_cap = kPinned;
  if (workspace_bytes) {
    int rc = dm_ws_reserve(ctx, workspace_bytes);
    if (rc != DM_OK) {
      (void)hipHostFree(hp);
      if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
      delete ctx;
      return rc;
    }
  }
  *out = ctx;
  return DM_OK;
}

int dm_ctx_destroy(dm_ctx* ctx) {
  if (!ctx) return DM_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (void* p : ctx->retired) (void)hipFree(p);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->hpin) (void)hipHostFree(ctx->hpin);
  if (ctx->hpin_dl) (void)hipHostFree(ctx->hpin_dl);
  if (ctx->prof_dev) (void)hipFree(ctx->prof_dev);
  for (auto& r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return DM_OK;
}

const char* dm_last_error(dm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dm_ctx_sync(dm_ctx* ctx) {
  if (!ctx) return DM_EARG;
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DM_OK;
}

size_t dm_ctx_workspace_bytes(dm_ctx* ctx) { return ctx ? ctx->ws_cap : 0; }

int dm_ctx_workspace_reset(dm_ctx* ctx, size_t bytes) {
  if (!ctx) return DM_EARG;
  DM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  DM_ARG(ctx, ctx->ws_used == 0);  // nothing may be live in the arena
  for (void* p : ctx->retired) (void)hipFree(p);
  ctx->retired.clear();
  if (ctx->ws) (void)hipFree(ctx->ws);
  ctx->ws = nullptr;
  ctx->ws_cap = 0;
  ctx->ws_used = 0;
  return bytes ? ws_grow(ctx, bytes, bytes) : DM_OK;
}

int dm_version(void) { return 100; }

}  // extern "C"
