// Host runtime behind the device half of include/zkir_hip.h: contexts, tape
// upload, launches.  Nothing here computes field arithmetic on the CPU; if no
// gfx950 device is present every call fails with ZKIR_E_NODEVICE / ZKIR_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/zkir_hip.h"
#include "device_structs.h"
#include "tape_check.h"

namespace zkir {
hipError_t launch_interp(int variant, const TapeDesc* tapes_dev, uint32_t n_tapes, uint32_t passes_per_tape,
                         uint64_t n_vectors, const uint32_t* inputs, uint32_t* outputs, void* gslot_pool,
                         uint32_t gslots_per_block, uint32_t lslots, uint64_t* unsat_per_tape, uint64_t* stats,
                         uint32_t* work_counter, uint32_t grid, uint32_t strands, uint32_t xslots, hipStream_t stream);
hipError_t interp_blocks_per_cu(int variant, uint32_t lslots, uint32_t strands, uint32_t xslots, int* blocks);
hipError_t launch_consts_to_mont(uint32_t* consts_dev, uint32_t n, hipStream_t stream);
hipError_t launch_int_mul_rate(int which, uint32_t* out, uint32_t blocks, int iters, hipStream_t stream);
hipError_t launch_fr_mul_rate(uint32_t* out, uint32_t blocks, int iters, hipStream_t stream);
}  // namespace zkir

using zkir::TapeDesc;

struct DevTape {
  zkir_tape_hdr hdr{};
  bool full = false;       // uses an operator beyond + - * : needs the full interpreter
  void* ops = nullptr;     // device
  void* consts = nullptr;  // device, Montgomery form
  int slab = -1;           // which slab of the context's tape arena holds ops and consts
};

// Device memory for uploaded tapes: slabs with bump allocation, so that an upload costs one
// copy and one small kernel instead of two hipMalloc calls.  A slab is reused from its start
// once every tape in it has been released; slabs are freed with the context.
struct TapeSlab {
  char* base = nullptr;
  size_t size = 0, used = 0;
  int live = 0;  // tapes currently allocated in this slab
};

struct zkir_ctx {
  int device = 0;
  int n_cu = 0;
  std::string err;
  std::map<zkir_tape_id, DevTape> tapes;
  zkir_tape_id next_id = 1;
  // growable device buffers owned by the context
  void* gslot_pool = nullptr;
  size_t gslot_pool_bytes = 0;
  void* desc_buf = nullptr;
  size_t desc_buf_bytes = 0;
  void* result_buf = nullptr;  // 4 x uint64 for the host-pointer entry points
  void* work_counter = nullptr;  // uint32 the workgroups of a launch draw work items from
  void* io_in = nullptr;
  size_t io_in_bytes = 0;
  void* io_out = nullptr;
  size_t io_out_bytes = 0;
  void* io_unsat = nullptr;
  size_t io_unsat_bytes = 0;
  uint64_t last_faults = 0;
  // the host-pointer entry point's copy pipeline for large results
  hipStream_t pipe_stream = nullptr;
  hipEvent_t pipe_done[2] = {nullptr, nullptr};
  void* pipe_stage[2] = {nullptr, nullptr};  // pinned host memory: a piece of the result on its way to the caller
  std::vector<TapeSlab> slabs;
  hipEvent_t upload_done = nullptr;  // recorded after the latest upload's device work
  bool upload_pending = false;       // an evaluation on another stream must wait for upload_done
  bool launched_since_sync = false;  // a launch may still be reading tapes: release must wait for the device
};

namespace {

int32_t fail(zkir_ctx* ctx, int32_t code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}
int32_t hip_fail(zkir_ctx* ctx, hipError_t e, const char* what) {
  return fail(ctx, ZKIR_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, expr)                                  \
  do {                                                      \
    hipError_t e_ = (expr);                                 \
    if (e_ != hipSuccess) return hip_fail(ctx, e_, #expr);  \
  } while (0)

// {unsat count, first unsat pair, faults, spare} = {0, all ones, 0, 0}, written on the stream by the
// device (no host memory involved, so nothing has to outlive the call)
hipError_t init_result(uint64_t* result_dev, hipStream_t s) {
  hipError_t e = hipMemsetAsync(result_dev, 0, 4 * sizeof(uint64_t), s);
  if (e == hipSuccess) e = hipMemsetAsync(result_dev + 1, 0xff, sizeof(uint64_t), s);
  return e;
}

int32_t ensure(zkir_ctx* ctx, void** buf, size_t* have, size_t need) {
  if (*have >= need && *buf) return ZKIR_OK;
  if (*buf) {
    HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
  }
  size_t sz = std::max<size_t>(need, 256);
  hipError_t e = hipMalloc(buf, sz);
  if (e != hipSuccess) return fail(ctx, ZKIR_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  *have = sz;
  return ZKIR_OK;
}

// `bytes` of 256-byte-aligned device memory from the tape arena; *slab_out names the slab
int32_t arena_alloc(zkir_ctx* ctx, size_t bytes, void** out, int* slab_out) {
  const size_t need = (std::max<size_t>(bytes, 256) + 255) & ~(size_t)255;
  for (size_t i = 0; i < ctx->slabs.size(); i++) {
    TapeSlab& sl = ctx->slabs[i];
    if (sl.live == 0) sl.used = 0;  // everything in it was released
    if (sl.size - sl.used >= need) {
      *out = sl.base + sl.used;
      sl.used += need;
      sl.live++;
      *slab_out = (int)i;
      return ZKIR_OK;
    }
  }
  TapeSlab sl;
  sl.size = std::max<size_t>(need, (size_t)32 << 20);
  hipError_t e = hipMalloc((void**)&sl.base, sl.size);
  if (e != hipSuccess) return fail(ctx, ZKIR_E_NOMEM, std::string("hipMalloc (tape arena): ") + hipGetErrorString(e));
  sl.used = need;
  sl.live = 1;
  *out = sl.base;
  ctx->slabs.push_back(sl);
  *slab_out = (int)ctx->slabs.size() - 1;
  return ZKIR_OK;
}

// Grid: as many workgroups as the machine holds at once (the occupancy the kernel's
// registers and its LDS allow), never more than there is work; workgroups draw work items from a
// device-wide counter and reuse their slot region from item to item.
uint32_t pick_grid(const zkir_ctx* ctx, uint64_t work_items, int variant, uint32_t lslots, uint32_t strands,
                   uint32_t xslots) {
  int per_cu = 2;
  if (zkir::interp_blocks_per_cu(variant, lslots, strands, xslots, &per_cu) != hipSuccess) per_cu = 2;
  if (const char* cap = std::getenv("ZKIR_BLOCKS_PER_CU"))  // measurement aid: fewer resident workgroups
    per_cu = std::max(1, std::min(per_cu, std::atoi(cap)));
  const uint64_t g = (uint64_t)ctx->n_cu * (uint64_t)per_cu;
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(g, work_items));
}

// Which interpreter kernel runs a launch: 1 (compiled, every operator) when a tape uses an
// operator beyond + - *; otherwise 2, the generated instruction stream, which needs 32*N to
// fit 32 bits; otherwise 0, the compiled + - * kernel.  ZKIR_INTERP=compiled in the
// environment forces the compiled kernels (a debugging aid: all three give the same bits).
int pick_variant(bool full, uint64_t N) {
  if (full) return 1;
  const char* want = std::getenv("ZKIR_INTERP");
  if (want && std::strcmp(want, "compiled") == 0) return 0;
  return N * 32 < (1ull << 32) ? 2 : 0;
}

// The tapes of one call, split by kind: plain tapes run 256 vectors per workgroup, strand tapes 64 (their
// four wavefronts are the four strands), so a call that holds both becomes two launches on the same
// stream.  Input / output offsets and the unsat index of a tape are those of its place in the call.
struct Launch {
  std::vector<TapeDesc> descs;
  bool full = false;
  uint32_t max_gslots = 0, max_lslots = 0, max_xslots = 0;
};

int32_t build_launches(zkir_ctx* ctx, const zkir_tape_id* ids, uint32_t n, uint64_t N, Launch L[2]) {
  uint64_t in_off = 0, out_off = 0;
  for (uint32_t i = 0; i < n; i++) {
    auto it = ctx->tapes.find(ids[i]);
    if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
    const DevTape& t = it->second;
    const bool strands = (t.hdr.flags & ZKIR_TAPE_F_STRANDS) != 0;
    Launch& G = L[strands ? 1 : 0];
    G.descs.emplace_back();
    TapeDesc& d = G.descs.back();
    std::memset(&d, 0, sizeof d);
    d.ops = (const zkir_op*)t.ops;
    d.consts = (const uint32_t*)t.consts;
    d.in_word_off = in_off;
    d.out_word_off = out_off;
    d.n_ops = t.hdr.n_ops;
    d.n_inputs = t.hdr.n_inputs;
    d.n_outputs = t.hdr.n_outputs;
    d.n_constraints = t.hdr.n_constraints;
    d.n_consts = t.hdr.n_consts;
    d.unsat_index = i;
    if (strands) {
      d.n_xslots = t.hdr.n_xslots;
      uint32_t begin = 0;
      for (uint32_t s = 0; s < ZKIR_STRANDS; s++) {
        d.strand_begin[s] = begin;
        d.strand_ops[s] = t.hdr.strand_ops[s];
        begin += t.hdr.strand_ops[s];
      }
      G.max_xslots = std::max(G.max_xslots, t.hdr.n_xslots);
    }
    in_off += (uint64_t)t.hdr.n_inputs * 8 * N;
    out_off += (uint64_t)t.hdr.n_outputs * 8 * N;
    G.full = G.full || t.full;
    G.max_gslots = std::max(G.max_gslots, t.hdr.n_gslots);
    G.max_lslots = std::max(G.max_lslots, t.hdr.n_lslots);
  }
  return ZKIR_OK;
}

// Launch the interpreter on device buffers.  unsat_dev has n entries, stats_dev 2
// entries ({packed first unsat, faults}); both must already be initialised.
int32_t run_tapes(zkir_ctx* ctx, const zkir_tape_id* ids, uint32_t n, const uint32_t* in_dev, uint64_t N,
                  uint32_t* out_dev, uint64_t* unsat_dev, uint64_t* stats_dev, hipStream_t stream) {
  if (N == 0 || n == 0) return ZKIR_OK;
  if (N >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "N must be below 2^32");
  Launch L[2];
  int32_t rc = build_launches(ctx, ids, n, N, L);
  if (rc) return rc;
  struct Plan {
    int variant;
    uint32_t strands, passes, grid;
  } plan[2];
  const size_t slot_bytes = (size_t)2 * ZKIR_BLOCK_THREADS * 16;
  size_t pool_bytes = 0;
  for (int k = 0; k < 2; k++) {
    const Launch& G = L[k];
    if (G.descs.empty()) continue;
    Plan& P = plan[k];
    P.strands = (uint32_t)k;
    // a pass is one workgroup's worth of vectors: 256, or 64 when its four wavefronts are the four strands
    const uint32_t per_pass = k ? 64u : (uint32_t)ZKIR_BLOCK_THREADS;
    P.passes = (uint32_t)((N + per_pass - 1) / per_pass);
    const uint64_t work = (uint64_t)G.descs.size() * P.passes;
    if (work >= (1ull << 32) - (1u << 20)) return fail(ctx, ZKIR_E_LIMIT, "too many (tape, pass) work items for one launch");
    P.variant = pick_variant(G.full, N);
    if (k) {  // only the generated stream runs strand tapes (upload refused any other operator)
      if (N * 32 >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "strand tapes take fewer than 2^27 vectors per launch");
      P.variant = 2;
    }
    P.grid = pick_grid(ctx, work, P.variant, G.max_lslots, P.strands, G.max_xslots);
    pool_bytes = std::max(pool_bytes, (size_t)P.grid * G.max_gslots * slot_bytes);
  }
  rc = ensure(ctx, &ctx->gslot_pool, &ctx->gslot_pool_bytes, pool_bytes);
  if (rc) return rc;
  rc = ensure(ctx, &ctx->desc_buf, &ctx->desc_buf_bytes, (size_t)n * sizeof(TapeDesc));
  if (rc) return rc;
  if (ctx->upload_pending) {  // constants of freshly uploaded tapes are converted on the null stream
    HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->upload_done, 0));
    ctx->upload_pending = false;
  }
  // the two launches draw work items from two counters (64 bytes apart) zeroed here
  HIP_TRY(ctx, hipMemsetAsync(ctx->work_counter, 0, 256, stream));
  size_t desc_at = 0;
  for (int k = 0; k < 2; k++) {
    const Launch& G = L[k];
    if (G.descs.empty()) continue;
    TapeDesc* descs_dev = (TapeDesc*)ctx->desc_buf + desc_at;
    // G.descs is pageable host memory: hipMemcpyAsync has consumed it when it returns,
    // and the device-side staging buffer is only rewritten by later work on `stream`.
    HIP_TRY(ctx, hipMemcpyAsync(descs_dev, G.descs.data(), G.descs.size() * sizeof(TapeDesc), hipMemcpyHostToDevice, stream));
    hipError_t e = zkir::launch_interp(plan[k].variant, descs_dev, (uint32_t)G.descs.size(), plan[k].passes, N, in_dev, out_dev,
                                       ctx->gslot_pool, G.max_gslots, G.max_lslots, unsat_dev, stats_dev,
                                       (uint32_t*)ctx->work_counter + 16 * k, plan[k].grid, plan[k].strands, G.max_xslots, stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch zkir_interp_kernel");
    ctx->launched_since_sync = true;
    desc_at += G.descs.size();
  }
  return ZKIR_OK;
}

}  // namespace

extern "C" {

int32_t zkir_ctx_create(int32_t device, zkir_ctx** out) {
  if (!out) return ZKIR_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZKIR_E_NODEVICE;
  if (device < 0 || device >= count) return ZKIR_E_INVALID;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ZKIR_E_HIP;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ZKIR_E_NODEVICE;  // kernels are gfx950 only
  zkir_ctx* ctx = new (std::nothrow) zkir_ctx();
  if (!ctx) return ZKIR_E_NOMEM;
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&ctx->result_buf, 4 * sizeof(uint64_t)) != hipSuccess ||
      hipMalloc(&ctx->work_counter, 256) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->upload_done, hipEventDisableTiming) != hipSuccess) {
    if (ctx->result_buf) (void)hipFree(ctx->result_buf);
    if (ctx->work_counter) (void)hipFree(ctx->work_counter);
    delete ctx;
    return ZKIR_E_HIP;
  }
  *out = ctx;
  return ZKIR_OK;
}

int32_t zkir_ctx_destroy(zkir_ctx* ctx) {
  if (!ctx) return ZKIR_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (TapeSlab& sl : ctx->slabs) (void)hipFree(sl.base);
  if (ctx->upload_done) (void)hipEventDestroy(ctx->upload_done);
  void* bufs[] = {ctx->gslot_pool, ctx->desc_buf, ctx->result_buf, ctx->work_counter, ctx->io_in, ctx->io_out, ctx->io_unsat};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (void* b : ctx->pipe_stage)
    if (b) (void)hipHostFree(b);
  for (hipEvent_t ev : ctx->pipe_done)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->pipe_stream) (void)hipStreamDestroy(ctx->pipe_stream);
  delete ctx;
  return ZKIR_OK;
}

const char* zkir_last_error(const zkir_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t zkir_tape_upload(zkir_ctx* ctx, const zkir_tape_hdr* hdr, const void* blob, size_t bytes, zkir_tape_id* out) {
  if (!ctx || !hdr || !out) return ZKIR_E_INVALID;
  std::string why;
  if (!zkir::check_tape(*hdr, blob, bytes, &why)) return fail(ctx, ZKIR_E_INVALID, "bad tape: " + why);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  DevTape t;
  t.hdr = *hdr;
  for (uint32_t i = 0; i < hdr->n_ops; i++) {
    const uint32_t c = ((const zkir_op*)blob)[i].code & 0xffu;
    if (c == ZKIR_OP_BNOT || c == ZKIR_OP_BNOT256 || (c >= ZKIR_OP_POW && c <= ZKIR_OP_SHL)) t.full = true;
  }
  if (t.full && (hdr->flags & ZKIR_TAPE_F_STRANDS))
    return fail(ctx, ZKIR_E_INVALID, "bad tape: strand tapes are made of + - * and data movement only");
  // ops and constants are adjacent in the blob and stay adjacent on the device: one copy
  const size_t ops_bytes = (size_t)hdr->n_ops * sizeof(zkir_op);
  const size_t consts_bytes = (size_t)hdr->n_consts * 32;
  void* base = nullptr;
  int32_t rc = arena_alloc(ctx, ops_bytes + consts_bytes, &base, &t.slab);
  if (rc) return rc;
  t.ops = base;
  t.consts = (char*)base + ops_bytes;
  hipError_t e = hipSuccess;
  if (ops_bytes + consts_bytes) e = hipMemcpy(base, blob, ops_bytes + consts_bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = zkir::launch_consts_to_mont((uint32_t*)t.consts, hdr->n_consts, nullptr);
  if (e == hipSuccess) e = hipEventRecord(ctx->upload_done, nullptr);
  if (e != hipSuccess) {
    ctx->slabs[t.slab].live--;
    return hip_fail(ctx, e, "tape upload");
  }
  ctx->upload_pending = true;
  *out = ctx->next_id++;
  ctx->tapes[*out] = t;
  return ZKIR_OK;
}

int32_t zkir_tape_upload_many(zkir_ctx* ctx, uint32_t n, const zkir_tape_hdr* hdrs, const void* const* blobs,
                              const size_t* bytes, zkir_tape_id* out) {
  if (!ctx || (n && (!hdrs || !blobs || !bytes || !out))) return ZKIR_E_INVALID;
  if (n == 0) return ZKIR_OK;
  std::string why;
  // device layout of the batch: every tape's ops, then (256-byte aligned) every tape's constants,
  // so that one kernel converts all constants
  std::vector<DevTape> tapes(n);
  size_t ops_total = 0, consts_total = 0;
  for (uint32_t k = 0; k < n; k++) {
    if (!zkir::check_tape(hdrs[k], blobs[k], bytes[k], &why))
      return fail(ctx, ZKIR_E_INVALID, "bad tape " + std::to_string(k) + ": " + why);
    DevTape& t = tapes[k];
    t.hdr = hdrs[k];
    for (uint32_t i = 0; i < hdrs[k].n_ops; i++) {
      const uint32_t c = ((const zkir_op*)blobs[k])[i].code & 0xffu;
      if (c == ZKIR_OP_BNOT || c == ZKIR_OP_BNOT256 || (c >= ZKIR_OP_POW && c <= ZKIR_OP_SHL)) t.full = true;
    }
    if (t.full && (hdrs[k].flags & ZKIR_TAPE_F_STRANDS))
      return fail(ctx, ZKIR_E_INVALID, "bad tape: strand tapes are made of + - * and data movement only");
    ops_total += (size_t)hdrs[k].n_ops * sizeof(zkir_op);
    consts_total += (size_t)hdrs[k].n_consts * 32;
  }
  if (consts_total / 32 >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "too many constants in one upload");
  const size_t consts_off = (ops_total + 255) & ~(size_t)255;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<char> staging;
  try {
    staging.resize(consts_off + consts_total);
  } catch (const std::bad_alloc&) {
    return fail(ctx, ZKIR_E_NOMEM, "host staging buffer");
  }
  void* base = nullptr;
  int slab = -1;
  int32_t rc = arena_alloc(ctx, consts_off + consts_total, &base, &slab);
  if (rc) return rc;
  size_t o = 0, c = consts_off;
  for (uint32_t k = 0; k < n; k++) {
    const size_t ob = (size_t)hdrs[k].n_ops * sizeof(zkir_op), cb = (size_t)hdrs[k].n_consts * 32;
    if (ob) std::memcpy(staging.data() + o, blobs[k], ob);
    if (cb) std::memcpy(staging.data() + c, (const char*)blobs[k] + ob, cb);
    tapes[k].ops = (char*)base + o;
    tapes[k].consts = (char*)base + c;
    tapes[k].slab = slab;
    o += ob;
    c += cb;
  }
  hipError_t e = hipSuccess;
  if (!staging.empty()) e = hipMemcpy(base, staging.data(), staging.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = zkir::launch_consts_to_mont((uint32_t*)((char*)base + consts_off), (uint32_t)(consts_total / 32), nullptr);
  if (e == hipSuccess) e = hipEventRecord(ctx->upload_done, nullptr);
  if (e != hipSuccess) {
    ctx->slabs[slab].live--;
    return hip_fail(ctx, e, "tape upload");
  }
  ctx->slabs[slab].live += (int)n - 1;  // arena_alloc counted one; every tape is released on its own
  ctx->upload_pending = true;
  for (uint32_t k = 0; k < n; k++) {
    out[k] = ctx->next_id++;
    ctx->tapes[out[k]] = tapes[k];
  }
  return ZKIR_OK;
}

int32_t zkir_tape_release(zkir_ctx* ctx, zkir_tape_id id) {
  if (!ctx) return ZKIR_E_INVALID;
  auto it = ctx->tapes.find(id);
  if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->launched_since_sync) {  // no launch may still be reading the tape
    HIP_TRY(ctx, hipDeviceSynchronize());
    ctx->launched_since_sync = false;
  }
  ctx->slabs[it->second.slab].live--;
  ctx->tapes.erase(it);
  return ZKIR_OK;
}

int32_t zkir_eval_batch_dev(zkir_ctx* ctx, zkir_tape_id tape, const uint32_t* inputs_soa_dev, uint64_t N,
                            uint32_t* outputs_soa_dev, uint64_t* result_dev, void* stream) {
  if (!ctx || !result_dev) return ZKIR_E_INVALID;
  auto it = ctx->tapes.find(tape);
  if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
  if (it->second.hdr.n_inputs && N && !inputs_soa_dev) return fail(ctx, ZKIR_E_INVALID, "inputs required");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(ctx, init_result(result_dev, s));
  return run_tapes(ctx, &tape, 1, inputs_soa_dev, N, outputs_soa_dev, result_dev, result_dev + 1, s);
}

namespace {

// `bytes` from pinned memory into the caller's (pageable) memory on a few host threads: one thread copies
// at about 20 GB/s, the link delivers close to 60
void parallel_copy(char* dst, const char* src, size_t bytes) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t n_threads = std::max<size_t>(1, std::min<size_t>((size_t)6, (size_t)(hw ? hw : 1)));
  if (n_threads == 1 || bytes < ((size_t)8 << 20)) {
    std::memcpy(dst, src, bytes);
    return;
  }
  const size_t per = ((bytes + n_threads - 1) / n_threads + 4095) & ~(size_t)4095;
  std::vector<std::thread> pool;
  for (size_t t = 1; t < n_threads; t++)
    if (t * per < bytes) pool.emplace_back([=] { std::memcpy(dst + t * per, src + t * per, std::min(per, bytes - t * per)); });
  std::memcpy(dst, src, std::min(per, bytes));
  for (std::thread& th : pool) th.join();
}

// Device to pageable host memory for large results: pieces go to pinned memory by DMA (two buffers),
// and while piece k is on the link, host threads move piece k-1 from pinned memory into the caller's
// buffer.  A plain hipMemcpy into pageable memory does the same with one thread.
int32_t copy_out_pipelined(zkir_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  const size_t piece = (size_t)64 << 20;
  if (!ctx->pipe_stream) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pipe_stream, hipStreamNonBlocking));
    for (hipEvent_t& ev : ctx->pipe_done) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  for (void*& b : ctx->pipe_stage)
    if (!b && hipHostMalloc(&b, piece, hipHostMallocDefault) != hipSuccess) return fail(ctx, ZKIR_E_NOMEM, "pinned staging buffer");
  hipStream_t s = ctx->pipe_stream;
  const size_t n = (bytes + piece - 1) / piece;
  auto size_of = [&](size_t k) { return std::min(piece, bytes - k * piece); };
  for (size_t k = 0; k < n; k++) {
    // buffer k&1 was emptied into the caller's memory before piece k-1 was waited for (below)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->pipe_stage[k & 1], (const char*)src_dev + k * piece, size_of(k), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipEventRecord(ctx->pipe_done[k & 1], s));
    if (k > 0) {
      HIP_TRY(ctx, hipEventSynchronize(ctx->pipe_done[(k - 1) & 1]));
      parallel_copy((char*)dst_host + (k - 1) * piece, (const char*)ctx->pipe_stage[(k - 1) & 1], size_of(k - 1));
    }
  }
  HIP_TRY(ctx, hipEventSynchronize(ctx->pipe_done[(n - 1) & 1]));
  parallel_copy((char*)dst_host + (n - 1) * piece, (const char*)ctx->pipe_stage[(n - 1) & 1], size_of(n - 1));
  return ZKIR_OK;
}

}  // namespace

int32_t zkir_eval_batch(zkir_ctx* ctx, zkir_tape_id tape, const uint32_t* inputs_soa, uint64_t N, uint32_t* outputs_soa,
                        uint64_t* unsat_count, uint32_t* first_unsat) {
  if (!ctx || !unsat_count) return ZKIR_E_INVALID;
  auto it = ctx->tapes.find(tape);
  if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
  const zkir_tape_hdr& h = it->second.hdr;
  if (h.n_inputs && N && !inputs_soa) return fail(ctx, ZKIR_E_INVALID, "inputs required");
  if (N >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "N must be below 2^32");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t in_bytes = (size_t)h.n_inputs * 32 * N, out_bytes = (size_t)h.n_outputs * 32 * N;
  int32_t rc = ensure(ctx, &ctx->io_in, &ctx->io_in_bytes, in_bytes);
  if (rc) return rc;
  if (outputs_soa) {
    rc = ensure(ctx, &ctx->io_out, &ctx->io_out_bytes, out_bytes);
    if (rc) return rc;
  }
  if (in_bytes) HIP_TRY(ctx, hipMemcpy(ctx->io_in, inputs_soa, in_bytes, hipMemcpyHostToDevice));
  rc = zkir_eval_batch_dev(ctx, tape, (const uint32_t*)ctx->io_in, N, outputs_soa ? (uint32_t*)ctx->io_out : nullptr,
                           (uint64_t*)ctx->result_buf, nullptr);
  if (rc) return rc;
  uint64_t res[4];
  HIP_TRY(ctx, hipMemcpy(res, ctx->result_buf, sizeof res, hipMemcpyDeviceToHost));
  if (outputs_soa && out_bytes) {
    // ZKIR_COPY_PIPELINE=0 / 1 forces the plain copy / the pipelined one (test aid); by default the
    // pipeline takes results of 128 MiB and more
    const char* force = std::getenv("ZKIR_COPY_PIPELINE");
    const bool piped = force ? force[0] == '1' : out_bytes >= ((size_t)128 << 20);
    if (piped) {
      rc = copy_out_pipelined(ctx, outputs_soa, ctx->io_out, out_bytes);  // res above was copied after the launch: it has finished
      if (rc) return rc;
    } else {
      HIP_TRY(ctx, hipMemcpy(outputs_soa, ctx->io_out, out_bytes, hipMemcpyDeviceToHost));
    }
  }
  *unsat_count = res[0];
  ctx->last_faults = res[2];
  if (first_unsat) {
    first_unsat[0] = res[1] == ~0ull ? 0xffffffffu : (uint32_t)(res[1] >> 32);
    first_unsat[1] = res[1] == ~0ull ? 0xffffffffu : (uint32_t)(res[1] & 0xffffffffu);
  }
  return ZKIR_OK;
}

int32_t zkir_eval_multi_dev(zkir_ctx* ctx, const zkir_tape_id* tapes, uint32_t n_tapes, const uint32_t* inputs_dev,
                            uint64_t N_per_tape, uint64_t* per_tape_unsat_dev, void* stream) {
  if (!ctx || (!tapes && n_tapes) || (!per_tape_unsat_dev && n_tapes)) return ZKIR_E_INVALID;
  if (n_tapes == 0) return ZKIR_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(ctx, hipMemsetAsync(per_tape_unsat_dev, 0, (size_t)n_tapes * sizeof(uint64_t), s));
  HIP_TRY(ctx, init_result((uint64_t*)ctx->result_buf, s));
  return run_tapes(ctx, tapes, n_tapes, inputs_dev, N_per_tape, nullptr, per_tape_unsat_dev,
                   (uint64_t*)ctx->result_buf + 1, s);
}

int32_t zkir_eval_multi(zkir_ctx* ctx, const zkir_tape_id* tapes, uint32_t n_tapes, const uint32_t* inputs,
                        uint64_t N_per_tape, uint64_t* per_tape_unsat) {
  if (!ctx || (!tapes && n_tapes) || (!per_tape_unsat && n_tapes)) return ZKIR_E_INVALID;
  if (n_tapes == 0) return ZKIR_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t in_bytes = 0;
  for (uint32_t i = 0; i < n_tapes; i++) {
    auto it = ctx->tapes.find(tapes[i]);
    if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
    in_bytes += (size_t)it->second.hdr.n_inputs * 32 * N_per_tape;
  }
  if (in_bytes && !inputs) return fail(ctx, ZKIR_E_INVALID, "inputs required");
  int32_t rc = ensure(ctx, &ctx->io_in, &ctx->io_in_bytes, in_bytes);
  if (rc) return rc;
  rc = ensure(ctx, &ctx->io_unsat, &ctx->io_unsat_bytes, (size_t)n_tapes * sizeof(uint64_t));
  if (rc) return rc;
  if (in_bytes) HIP_TRY(ctx, hipMemcpy(ctx->io_in, inputs, in_bytes, hipMemcpyHostToDevice));
  rc = zkir_eval_multi_dev(ctx, tapes, n_tapes, (const uint32_t*)ctx->io_in, N_per_tape, (uint64_t*)ctx->io_unsat,
                           nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(per_tape_unsat, ctx->io_unsat, (size_t)n_tapes * sizeof(uint64_t), hipMemcpyDeviceToHost));
  uint64_t res[4];
  HIP_TRY(ctx, hipMemcpy(res, ctx->result_buf, sizeof res, hipMemcpyDeviceToHost));
  ctx->last_faults = res[2];
  return ZKIR_OK;
}

int32_t zkir_last_faults(const zkir_ctx* ctx, uint64_t* faults) {
  if (!ctx || !faults) return ZKIR_E_INVALID;
  *faults = ctx->last_faults;
  return ZKIR_OK;
}

static int32_t time_probe(zkir_ctx* ctx, int which, bool fr, double* out) {
  if (!ctx || !out) return ZKIR_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t blocks = (uint32_t)ctx->n_cu * 8;  // 8 waves per SIMD
  const int iters = fr ? 64 : 2048;
  uint32_t* buf = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&buf, (size_t)blocks * 256 * 4));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    (void)hipFree(buf);
    return fail(ctx, ZKIR_E_HIP, "rate probe: hipEventCreate failed");
  }
  auto launch = [&]() {
    return fr ? zkir::launch_fr_mul_rate(buf, blocks, iters, nullptr)
              : zkir::launch_int_mul_rate(which, buf, blocks, iters, nullptr);
  };
  hipError_t e = launch();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  const int reps = 5;
  if (e == hipSuccess) {
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps && e == hipSuccess; r++) e = launch();
    (void)hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
  }
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  if (e != hipSuccess) return hip_fail(ctx, e, "rate probe");
  const double per_lane = fr ? 2.0 * iters : 16.0 * iters;
  *out = (double)blocks * 256 * per_lane * reps / (ms * 1e-3);
  return ZKIR_OK;
}

int32_t zkir_measure_int_mul_rate(zkir_ctx* ctx, int32_t which, double* lane_ops_per_s) {
  if (which < 0 || which > 2) return ZKIR_E_INVALID;
  return time_probe(ctx, which, false, lane_ops_per_s);
}

int32_t zkir_measure_fr_mul_rate(zkir_ctx* ctx, double* fr_mul_per_s) { return time_probe(ctx, 0, true, fr_mul_per_s); }

}  // extern "C"
