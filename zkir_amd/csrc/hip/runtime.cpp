// Host runtime behind the device half of include/zkir_hip.h: contexts, tape
// upload, launches.  Nothing here computes field arithmetic on the CPU; if no
// gfx950 device is present every call fails with ZKIR_E_NODEVICE / ZKIR_E_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/zkir_hip.h"
#include "device_structs.h"
#include "tape_check.h"
#include "../host/lower.hpp"  // zkir_tape: the handle zkir_lower and zkir_tape_from_blob hand out

namespace zkir {
hipError_t launch_interp(int variant, const TapeDesc* tapes_dev, uint32_t n_tapes, uint32_t passes_per_tape,
                         uint64_t n_vectors, const uint32_t* inputs, uint32_t* outputs, void* gslot_pool,
                         uint32_t gslots_per_block, uint32_t lslots, uint64_t* unsat_per_tape, uint64_t* stats,
                         uint32_t* work_counter, uint32_t grid, uint32_t strands, uint32_t xslots, uint32_t n_cu,
                         uint64_t* faults_per_tape, hipStream_t stream);
hipError_t interp_blocks_per_cu(int variant, uint32_t lslots, uint32_t strands, uint32_t xslots, int* blocks);
size_t interp_lds_need(int variant, uint32_t lslots, uint32_t strands, uint32_t xslots);
hipError_t launch_init(uint64_t* result_dev, uint32_t* counters_dev, hipStream_t stream);
hipError_t launch_consts_to_mont(uint32_t* consts_dev, uint32_t n, hipStream_t stream);
hipError_t launch_consts_to_mont29(const uint32_t* canonical_dev, uint32_t* dst_dev, uint32_t n, hipStream_t stream);
size_t interp_slot_bytes(int variant);
hipError_t launch_int_mul_rate(int which, uint32_t* out, uint32_t blocks, int iters, hipStream_t stream);
hipError_t launch_fr_mul_rate(uint32_t* out, uint32_t blocks, int iters, hipStream_t stream);
hipError_t launch_fr29_mul_rate(uint32_t* out, uint32_t blocks, int iters, hipStream_t stream);
}  // namespace zkir

using zkir::TapeDesc;

struct DevTape {
  zkir_tape_hdr hdr{};
  bool full = false;       // uses an operator beyond + - * : needs the full interpreter
  void* ops = nullptr;     // device
  void* consts = nullptr;  // device, Montgomery form (8 x 32-bit limbs, R = 2^256)
  void* consts29 = nullptr;  // device, the same constants as 9 x 29-bit limbs with R = 2^261, 12 words each
  void* desc = nullptr;    // device: the TapeDesc of this tape evaluated alone (offsets 0, unsat index 0)
  uint32_t one_word = 0xffffffffu;  // operand word of the constant 1 in the tape's pool, if it holds one
  int slab = -1;           // which slab of the context's tape arena holds ops, consts and desc
  uint64_t last_launch = 0;  // number of the latest chained launch that reads this tape (0: none yet)
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
  size_t lds_per_cu = 0;  // bytes of LDS a workgroup may take on this device
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
  bool upload_pending = false;       // evaluations must wait for upload_done (until the device has been synchronised)
  bool launched_since_sync = false;  // a launch may still be reading tapes: release must wait for the device
  // Launches of one context share its work counters, descriptor buffer and slot pool, so they run one
  // after the other whatever streams the caller uses: every launch waits for launch_done of the one before.
  // The events of the last kLaunchRing launches are kept (launch number n uses entry (n - 1) % kLaunchRing), so that
  // releasing a tape waits for the launch that read it and not for whatever was launched since.
  static constexpr int kLaunchRing = 8;
  hipEvent_t launch_ev[kLaunchRing] = {};
  uint64_t launch_seq = 0;    // chained launches recorded so far
  uint64_t done_seq = 0;      // launches known to have finished (chained launches finish in order)
  bool unchained = false;     // a launch went out without an event (ZKIR_NO_CHAIN, graph capture): only a device
                              // synchronisation says when it is over
  bool launch_recorded = false;
  // descriptors of a multi-tape call travel through pinned host memory the context owns (two buffers in
  // turn; a buffer is rewritten only after the copy that read it has finished)
  void* desc_stage[2] = {nullptr, nullptr};
  size_t desc_stage_bytes[2] = {0, 0};
  hipEvent_t desc_copied[2] = {nullptr, nullptr};
  bool desc_copy_recorded[2] = {false, false};
  int desc_turn = 0;
  unsigned char* static_scratch = nullptr;  // this device's copy of the file-scope scratch array below
  void* up_stage = nullptr;  // pinned host memory a batch of tapes is gathered in before its one copy to the device
  size_t up_stage_bytes = 0;
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

// `bytes` of 256-byte-aligned device memory from the tape arena; *slab_out names the slab,
// *mark_out what arena_rollback needs to give the bytes back
int32_t arena_alloc(zkir_ctx* ctx, size_t bytes, void** out, int* slab_out, size_t* mark_out) {
  const size_t need = (std::max<size_t>(bytes, 256) + 255) & ~(size_t)255;
  for (size_t i = 0; i < ctx->slabs.size(); i++) {
    TapeSlab& sl = ctx->slabs[i];
    if (sl.live == 0) sl.used = 0;  // everything in it was released
    if (sl.size - sl.used >= need) {
      *out = sl.base + sl.used;
      *mark_out = sl.used;
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
  *mark_out = 0;
  try {
    ctx->slabs.push_back(sl);
  } catch (const std::bad_alloc&) {
    (void)hipFree(sl.base);
    return fail(ctx, ZKIR_E_NOMEM, "tape arena bookkeeping");
  }
  *slab_out = (int)ctx->slabs.size() - 1;
  return ZKIR_OK;
}

// an upload failed after arena_alloc: the bytes go back when nothing was allocated behind them
void arena_rollback(zkir_ctx* ctx, int slab, size_t mark, size_t bytes) {
  TapeSlab& sl = ctx->slabs[slab];
  const size_t need = (std::max<size_t>(bytes, 256) + 255) & ~(size_t)255;
  sl.live--;
  if (sl.used == mark + need) sl.used = mark;
}

// does the tape use an operator beyond + - * (it then needs the kernel that knows every operator)
bool tape_is_full(const zkir_tape_hdr& h, const void* blob) {
  for (uint32_t i = 0; i < h.n_ops; i++) {
    const uint32_t c = ((const zkir_op*)blob)[i].code & 0xffu;
    if (c == ZKIR_OP_BNOT || c == ZKIR_OP_BNOT256 || (c >= ZKIR_OP_POW && c <= ZKIR_OP_SHL)) return true;
  }
  return false;
}

// descriptor of tape `t` at place `index` of a call whose blocks start at the given word offsets
TapeDesc make_desc(const DevTape& t, uint32_t index, uint64_t in_off, uint64_t out_off) {
  TapeDesc d;
  std::memset(&d, 0, sizeof d);
  d.ops = (const zkir_op*)t.ops;
  d.consts = (const uint32_t*)t.consts;
  d.consts29 = (const uint32_t*)t.consts29;
  d.one_word = t.one_word;
  d.in_word_off = in_off;
  d.out_word_off = out_off;
  d.n_ops = t.hdr.n_ops;
  d.n_inputs = t.hdr.n_inputs;
  d.n_outputs = t.hdr.n_outputs;
  d.n_constraints = t.hdr.n_constraints;
  d.n_consts = t.hdr.n_consts;
  d.unsat_index = index;
  if (t.hdr.flags & ZKIR_TAPE_F_STRANDS) {
    d.n_xslots = t.hdr.n_xslots;
    d.n_strands = t.hdr.n_strands;
    uint32_t begin = 0;
    for (uint32_t s = 0; s < t.hdr.n_strands && s < ZKIR_MAX_STRANDS; s++) {
      d.strand_begin[s] = begin;
      d.strand_ops[s] = t.hdr.strand_ops[s];
      begin += t.hdr.strand_ops[s];
    }
  }
  return d;
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

// one witness slot of one workgroup: 8 KiB, or 9 KiB in the kernel that keeps 9 limbs of 29 bits; the fixed
// scratch of the no-alloc entry point is counted in rows of the larger size
constexpr size_t kSlotBytesMax = 9216;

// Which interpreter kernel runs a launch: 1 (compiled, every operator) when a tape uses an
// operator beyond + - *; otherwise 3, the generated instruction stream on 9 x 29-bit limbs, which addresses
// a vector's data and a workgroup's slot region in 32 bits (32*N and 9216*gslots below 2^32);
// otherwise 0, the compiled + - * kernel.  Tapes that keep more than kSpillHeavySlots values in device
// memory take variant 2 instead, the generated stream on 8 x 32-bit limbs: such launches run at the rate
// device memory delivers their witness slots, and a 32-byte slot moves 11 % fewer bytes than a 36-byte one
// (measured on the 1 047-constraint circuit with 430 slots: 30.8 ms against 31.6; on the gadget, which
// keeps its values on chip, the 29-bit stream is 1.2 to 1.3 times faster).  ZKIR_INTERP in the
// environment forces a choice (a debugging and measurement aid: all four kernels give the same bits):
// "compiled", "lean32" (variant 2), "lean29" (variant 3).
constexpr uint32_t kSpillHeavySlots = 96;
int pick_variant(bool full, uint64_t N, uint32_t max_gslots) {
  if (full) return 1;
  const char* want = std::getenv("ZKIR_INTERP");
  if (want && std::strcmp(want, "compiled") == 0) return 0;
  int stream = max_gslots > kSpillHeavySlots ? 2 : 3;
  if (want && std::strcmp(want, "lean32") == 0) stream = 2;
  if (want && std::strcmp(want, "lean29") == 0) stream = 3;
  return (N * 32 < (1ull << 32) && (uint64_t)max_gslots * zkir::interp_slot_bytes(stream) < (1ull << 32)) ? stream : 0;
}

// The tapes of one kind in a call: plain tapes run 256 vectors per workgroup, strand tapes 64 (their
// four or eight wavefronts are the strands), so a call that holds several kinds becomes one launch per
// kind on the same stream.  Kinds: 0 plain, 1 four strands, 2 eight strands.
constexpr int kKinds = 3;
inline int kind_of(const zkir_tape_hdr& h) { return !(h.flags & ZKIR_TAPE_F_STRANDS) ? 0 : h.n_strands == ZKIR_MAX_STRANDS ? 2 : 1; }
inline uint32_t strands_of_kind(int kind) { return kind == 0 ? 0u : kind == 1 ? ZKIR_STRANDS : ZKIR_MAX_STRANDS; }
// sets of slot rows a workgroup keeps (a row holds four wavefronts)
inline uint32_t row_sets_of_kind(int kind) { return kind == 2 ? 2u : 1u; }
struct Group {
  uint32_t count = 0;
  bool full = false;
  uint32_t max_gslots = 0, max_lslots = 0, max_xslots = 0;
  void add(const DevTape& t) {
    count++;
    full = full || t.full;
    max_gslots = std::max(max_gslots, t.hdr.n_gslots);
    max_lslots = std::max(max_lslots, t.hdr.n_lslots);
    if (t.hdr.flags & ZKIR_TAPE_F_STRANDS) max_xslots = std::max(max_xslots, t.hdr.n_xslots);
  }
};
struct Plan {
  int variant = 0;
  uint32_t strands = 0, passes = 0, grid = 0;
};

// pool_rows_cap: 0, or the slot rows (of 9 KiB, the larger of the two streams' rows) the caller's fixed pool holds; the grid shrinks to fit
int32_t plan_group(zkir_ctx* ctx, const Group& G, int kind, uint64_t N, size_t pool_rows_cap, Plan* P) {
  P->strands = strands_of_kind(kind);
  P->variant = pick_variant(G.full, N, G.max_gslots);
  if (kind && P->variant == 0) {  // ZKIR_INTERP=compiled: strand tapes still take a generated stream
    const char* want = std::getenv("ZKIR_INTERP");
    if (want && std::strcmp(want, "compiled") == 0 && (uint64_t)G.max_gslots * kSlotBytesMax < (1ull << 32)) P->variant = 3;
  }
  // a pass is one workgroup's worth of vectors: 256, or 64 when its wavefronts are the strands.  (A wavefront of a
  // plain tape taking 64-vector items on its own was tried, to shorten the end of a launch: no gain on the
  // benchmark's launch, 7 % slower on the 29-bit stream, whose four wavefronts then stop reading neighbouring rows.)
  const uint32_t per_pass = kind ? 64u : (uint32_t)ZKIR_BLOCK_THREADS;
  P->passes = (uint32_t)((N + per_pass - 1) / per_pass);
  const uint64_t work = (uint64_t)G.count * P->passes;
  if (work >= (1ull << 32) - (1u << 20)) return fail(ctx, ZKIR_E_LIMIT, "too many (tape, pass) work items for one launch");
  if (kind) {  // only the generated stream runs strand tapes (upload refused any other operator)
    if (N * 32 >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "strand tapes take fewer than 2^27 vectors per launch");
    if (P->variant != 2 && P->variant != 3)
      return fail(ctx, ZKIR_E_LIMIT, "strand tapes take fewer than 466 033 device slots per strand");
  }
  size_t lds = zkir::interp_lds_need(P->variant, G.max_lslots, P->strands, G.max_xslots);
  // the 29-bit stream's rows are 9 KiB and it keeps a table: a launch that does not fit with them may fit the 32-bit stream's 8 KiB rows
  if (lds > ctx->lds_per_cu && P->variant == 3 && zkir::interp_lds_need(2, G.max_lslots, P->strands, G.max_xslots) <= ctx->lds_per_cu) {
    P->variant = 2;
    lds = zkir::interp_lds_need(2, G.max_lslots, P->strands, G.max_xslots);
  }
  if (lds > ctx->lds_per_cu)
    return fail(ctx, ZKIR_E_LIMIT, "the tapes of this launch need " + std::to_string(lds) + " bytes of LDS per workgroup (" +
                                       std::to_string(G.max_lslots) + " window rows, " + std::to_string(G.max_xslots) +
                                       " exchange slots); a CU has " + std::to_string(ctx->lds_per_cu));
  P->grid = pick_grid(ctx, work, P->variant, G.max_lslots, P->strands, G.max_xslots);
  if (pool_rows_cap && G.max_gslots) {
    const uint64_t rows = (uint64_t)G.max_gslots * row_sets_of_kind(kind);
    if (rows > pool_rows_cap)
      return fail(ctx, ZKIR_E_LIMIT, "the tape needs " + std::to_string(rows) + " device slot rows; the fixed scratch holds " +
                                         std::to_string(pool_rows_cap));
    P->grid = (uint32_t)std::min<uint64_t>(P->grid, pool_rows_cap / rows);
  }
  return ZKIR_OK;
}

// is the caller recording `stream` into a HIP graph?  Events recorded outside the capture may not be waited for
// inside it, and an event recorded inside it means nothing to launches outside
bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// what every launch of a context does first on its stream: wait for the constants of fresh uploads and for
// the launch before it (which may be on another stream and shares the counters, descriptors and slots).
// While the stream is being captured into a graph neither is done: the caller has synchronised with its uploads
// before the capture began and orders the graph's replays against the context's other launches itself.
int32_t chain_stream(zkir_ctx* ctx, hipStream_t stream) {
  if (stream_is_capturing(stream)) return ZKIR_OK;
  // the constants of an upload are converted on the null stream: a launch on any other stream has to wait for them
  if (ctx->upload_pending) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->upload_done, 0));
  static const bool off = std::getenv("ZKIR_NO_CHAIN") != nullptr;  // measurement aid: the caller keeps to one stream
  if (off) return ZKIR_OK;
  if (ctx->launch_recorded) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->launch_ev[(ctx->launch_seq - 1) % zkir_ctx::kLaunchRing], 0));
  return ZKIR_OK;
}
int32_t chain_done(zkir_ctx* ctx, hipStream_t stream) {
  static const bool off = std::getenv("ZKIR_NO_CHAIN") != nullptr;
  if (off || stream_is_capturing(stream)) {
    ctx->launched_since_sync = true;
    ctx->unchained = true;
    return ZKIR_OK;
  }
  HIP_TRY(ctx, hipEventRecord(ctx->launch_ev[ctx->launch_seq % zkir_ctx::kLaunchRing], stream));
  ctx->launch_seq++;
  ctx->launch_recorded = true;
  ctx->launched_since_sync = true;
  return ZKIR_OK;
}
bool make_launch_events(zkir_ctx* ctx) {
  for (hipEvent_t& ev : ctx->launch_ev)
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return false;
  return true;
}
// the device has been synchronised: nothing is in flight any more
void note_device_idle(zkir_ctx* ctx) {
  ctx->upload_pending = false;
  ctx->launched_since_sync = false;
  ctx->launch_recorded = false;
  ctx->unchained = false;
  ctx->done_seq = ctx->launch_seq;
}

int32_t launch_group(zkir_ctx* ctx, const Group& G, const Plan& P, const TapeDesc* descs_dev, const uint32_t* in_dev, uint64_t N,
                     uint32_t* out_dev, void* pool, uint32_t* counter, uint64_t* unsat_dev, uint64_t* stats_dev,
                     uint64_t* faults_per_tape_dev, hipStream_t stream) {
  hipError_t e = zkir::launch_interp(P.variant, descs_dev, G.count, P.passes, N, in_dev, out_dev, pool, G.max_gslots,
                                     G.max_lslots, unsat_dev, stats_dev, counter, P.grid, P.strands, G.max_xslots, (uint32_t)ctx->n_cu,
                                     faults_per_tape_dev, stream);
  if (e != hipSuccess) return hip_fail(ctx, e, "launch zkir_interp_kernel");
  return ZKIR_OK;
}

// One tape on device buffers.  unsat_dev points at the caller's four result words {unsat, packed first unsat,
// faults, spare} (stats_dev at the second), which this call initialises on the stream.  The tape's own descriptor (written at upload) is used, so nothing is
// copied.  fixed_scratch: null (the context's growable pool and counters) or a fixed region of
// `fixed_bytes` whose first 9 KiB hold the work counter and the rest slot rows; with it the call
// allocates nothing.
int32_t run_one(zkir_ctx* ctx, const DevTape& t, const uint32_t* in_dev, uint64_t N, uint32_t* out_dev, uint64_t* unsat_dev,
                uint64_t* stats_dev, hipStream_t stream, unsigned char* fixed_scratch, size_t fixed_bytes) {
  if (N == 0) {  // nothing to launch: the result words still start from {0, all ones, 0, 0}; the work counters are
                 // left alone (an earlier launch of this context on another stream may be drawing from them)
    HIP_TRY(ctx, zkir::launch_init(unsat_dev, nullptr, stream));
    return ZKIR_OK;
  }
  if (N >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "N must be below 2^32");
  Group G;
  G.add(t);
  Plan P;
  const int kind = kind_of(t.hdr);
  const size_t rows_cap = fixed_scratch ? (fixed_bytes - kSlotBytesMax) / kSlotBytesMax : 0;
  int32_t rc = plan_group(ctx, G, kind, N, rows_cap, &P);
  if (rc) return rc;
  void* pool = nullptr;
  uint32_t* counter = nullptr;
  if (fixed_scratch) {
    counter = (uint32_t*)fixed_scratch;
    pool = fixed_scratch + kSlotBytesMax;
  } else {
    rc = ensure(ctx, &ctx->gslot_pool, &ctx->gslot_pool_bytes,
                (size_t)P.grid * G.max_gslots * row_sets_of_kind(kind) * zkir::interp_slot_bytes(P.variant));
    if (rc) return rc;
    counter = (uint32_t*)ctx->work_counter;
    pool = ctx->gslot_pool;
  }
  rc = chain_stream(ctx, stream);
  if (rc) return rc;
  HIP_TRY(ctx, zkir::launch_init(unsat_dev, counter, stream));  // the result words (unsat_dev points at the first) and the counters
  rc = launch_group(ctx, G, P, (const TapeDesc*)t.desc, in_dev, N, out_dev, pool, counter, unsat_dev, stats_dev, nullptr, stream);
  if (rc) return rc;
  return chain_done(ctx, stream);
}

// pinned staging for `bytes` of descriptors; the buffer returned is free to be written
int32_t desc_stage_acquire(zkir_ctx* ctx, size_t bytes, int* which) {
  const int k = ctx->desc_turn;
  ctx->desc_turn ^= 1;
  if (!ctx->desc_copied[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->desc_copied[k], hipEventDisableTiming));
  if (ctx->desc_copy_recorded[k]) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->desc_copied[k]));
    ctx->desc_copy_recorded[k] = false;
  }
  if (ctx->desc_stage_bytes[k] < bytes) {
    if (ctx->desc_stage[k]) HIP_TRY(ctx, hipHostFree(ctx->desc_stage[k]));
    ctx->desc_stage[k] = nullptr;
    ctx->desc_stage_bytes[k] = 0;
    const size_t sz = std::max<size_t>(bytes + bytes / 2, (size_t)64 << 10);
    if (hipHostMalloc(&ctx->desc_stage[k], sz, hipHostMallocDefault) != hipSuccess)
      return fail(ctx, ZKIR_E_NOMEM, "pinned descriptor staging buffer");
    ctx->desc_stage_bytes[k] = sz;
  }
  *which = k;
  return ZKIR_OK;
}

// Launch the interpreter on device buffers for the tapes of a call.  unsat_dev has n entries, stats_dev 2
// entries ({packed first unsat, faults}); both must already be initialised.  Input / output offsets and
// the unsat index of a tape are those of its place in the call.
int32_t run_tapes(zkir_ctx* ctx, const zkir_tape_id* ids, uint32_t n, const uint32_t* in_dev, uint64_t N,
                  uint32_t* out_dev, uint64_t* unsat_dev, uint64_t* stats_dev, uint64_t* faults_per_tape_dev, hipStream_t stream) {
  if (N == 0 || n == 0) return ZKIR_OK;
  if (N >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "N must be below 2^32");
  Group G[kKinds];
  // descriptors of one pinned buffer: the plain tapes in call order, then the four-strand tapes, then the eight-strand ones
  int stage = 0;
  int32_t rc = desc_stage_acquire(ctx, (size_t)n * sizeof(TapeDesc), &stage);
  if (rc) return rc;
  TapeDesc* host_descs = (TapeDesc*)ctx->desc_stage[stage];
  uint32_t n_of_kind[kKinds] = {0, 0, 0};
  for (uint32_t i = 0; i < n; i++) {
    auto it = ctx->tapes.find(ids[i]);
    if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
    n_of_kind[kind_of(it->second.hdr)]++;
  }
  uint64_t in_off = 0, out_off = 0;
  uint32_t at[kKinds] = {0, n_of_kind[0], n_of_kind[0] + n_of_kind[1]};
  for (uint32_t i = 0; i < n; i++) {
    const DevTape& t = ctx->tapes.find(ids[i])->second;
    const int kind = kind_of(t.hdr);
    G[kind].add(t);
    host_descs[at[kind]++] = make_desc(t, i, in_off, out_off);
    in_off += (uint64_t)t.hdr.n_inputs * 8 * N;
    out_off += (uint64_t)t.hdr.n_outputs * 8 * N;
  }
  Plan plan[kKinds];
  size_t pool_bytes = 0;
  for (int k = 0; k < kKinds; k++) {
    if (!G[k].count) continue;
    rc = plan_group(ctx, G[k], k, N, 0, &plan[k]);
    if (rc) return rc;
    pool_bytes = std::max(pool_bytes, (size_t)plan[k].grid * G[k].max_gslots * row_sets_of_kind(k) *
                                          zkir::interp_slot_bytes(plan[k].variant));
  }
  rc = ensure(ctx, &ctx->gslot_pool, &ctx->gslot_pool_bytes, pool_bytes);
  if (rc) return rc;
  rc = ensure(ctx, &ctx->desc_buf, &ctx->desc_buf_bytes, (size_t)n * sizeof(TapeDesc));
  if (rc) return rc;
  rc = chain_stream(ctx, stream);
  if (rc) return rc;
  // the launches draw work items from counters of their own (64 bytes apart) zeroed here
  HIP_TRY(ctx, hipMemsetAsync(ctx->work_counter, 0, 256, stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->desc_buf, host_descs, (size_t)n * sizeof(TapeDesc), hipMemcpyHostToDevice, stream));
  HIP_TRY(ctx, hipEventRecord(ctx->desc_copied[stage], stream));
  ctx->desc_copy_recorded[stage] = true;
  size_t desc_at = 0;
  for (int k = 0; k < kKinds; k++) {
    if (!G[k].count) continue;
    rc = launch_group(ctx, G[k], plan[k], (const TapeDesc*)ctx->desc_buf + desc_at, in_dev, N, out_dev, ctx->gslot_pool,
                      (uint32_t*)ctx->work_counter + 16 * k, unsat_dev, stats_dev, faults_per_tape_dev, stream);
    if (rc) return rc;
    ctx->launched_since_sync = true;
    desc_at += G[k].count;
  }
  rc = chain_done(ctx, stream);
  if (rc) return rc;
  for (uint32_t i = 0; i < n; i++) ctx->tapes[ids[i]].last_launch = ctx->launch_seq;
  return ZKIR_OK;
}

}  // namespace

// Scratch of zkir_eval_batch_dev_noalloc: the work counter (in the first 9 KiB) and witness-slot rows of 9 KiB.
// Every GPU the library runs on holds its own copy until the process ends.  In host code the array's name
// is only a handle for hipGetSymbolAddress; the address kernels and memsets use is zkir_ctx::static_scratch.
static __device__ __align__(256) unsigned char scratch[ZKIR_NOALLOC_SCRATCH_BYTES];

namespace {
// Who launched on a device's scratch array last, and the event behind that launch: a call of ANOTHER context that
// arrives while that launch is still running would share its rows and its counter, so it is refused (ZKIR_E_BUSY)
// instead of racing.  Calls of one context are chained by its own events.  Under graph capture nothing is recorded.
constexpr int kMaxDevices = 64;
struct ScratchOwner {
  const zkir_ctx* ctx = nullptr;
  hipEvent_t done = nullptr;
  bool recorded = false;
};
std::mutex g_scratch_lock;
ScratchOwner g_scratch_owner[kMaxDevices];
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
  ctx->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor ? (size_t)prop.maxSharedMemoryPerMultiProcessor : (size_t)160 * 1024;
  // with `device` current: the address of this device's copy of the scratch array
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&ctx->result_buf, 4 * sizeof(uint64_t)) != hipSuccess ||
      hipMalloc(&ctx->work_counter, 256) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->upload_done, hipEventDisableTiming) != hipSuccess ||
      !make_launch_events(ctx) ||
      hipGetSymbolAddress((void**)&ctx->static_scratch, HIP_SYMBOL(scratch)) != hipSuccess || !ctx->static_scratch) {
    if (ctx->result_buf) (void)hipFree(ctx->result_buf);
    if (ctx->work_counter) (void)hipFree(ctx->work_counter);
    if (ctx->upload_done) (void)hipEventDestroy(ctx->upload_done);
    for (hipEvent_t ev : ctx->launch_ev)
      if (ev) (void)hipEventDestroy(ev);
    delete ctx;
    return ZKIR_E_HIP;
  }
  if (device < kMaxDevices) {  // the no-alloc entry point's per-device event is made here: that call creates nothing
    std::lock_guard<std::mutex> g(g_scratch_lock);
    if (!g_scratch_owner[device].done) (void)hipEventCreateWithFlags(&g_scratch_owner[device].done, hipEventDisableTiming);
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
  for (hipEvent_t ev : ctx->launch_ev)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->desc_copied)
    if (ev) (void)hipEventDestroy(ev);
  for (void* b : ctx->desc_stage)
    if (b) (void)hipHostFree(b);
  void* bufs[] = {ctx->gslot_pool, ctx->desc_buf, ctx->result_buf, ctx->work_counter, ctx->io_in, ctx->io_out, ctx->io_unsat};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (void* b : ctx->pipe_stage)
    if (b) (void)hipHostFree(b);
  if (ctx->up_stage) (void)hipHostFree(ctx->up_stage);
  for (hipEvent_t ev : ctx->pipe_done)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->pipe_stream) (void)hipStreamDestroy(ctx->pipe_stream);
  delete ctx;
  return ZKIR_OK;
}

const char* zkir_last_error(const zkir_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t zkir_tape_upload(zkir_ctx* ctx, const zkir_tape_hdr* hdr, const void* blob, size_t bytes, zkir_tape_id* out) {
  if (!ctx || !hdr || !out) return ZKIR_E_INVALID;
  const void* blobs[1] = {blob};
  return zkir_tape_upload_many(ctx, 1, hdr, blobs, &bytes, out);
}

static int32_t upload_impl(zkir_ctx* ctx, uint32_t n, const zkir_tape_hdr* hdrs, const void* const* blobs,
                           const size_t* bytes, zkir_tape_id* out, bool validated);

int32_t zkir_tape_upload_many(zkir_ctx* ctx, uint32_t n, const zkir_tape_hdr* hdrs, const void* const* blobs,
                              const size_t* bytes, zkir_tape_id* out) {
  if (!ctx || (n && (!hdrs || !blobs || !bytes || !out))) return ZKIR_E_INVALID;
  if (n == 0) return ZKIR_OK;
  return upload_impl(ctx, n, hdrs, blobs, bytes, out, false);
}

int32_t zkir_tape_upload_handles(zkir_ctx* ctx, uint32_t n, const zkir_tape* const* tapes, zkir_tape_id* out) {
  if (!ctx || (n && (!tapes || !out))) return ZKIR_E_INVALID;
  if (n == 0) return ZKIR_OK;
  std::vector<zkir_tape_hdr> hdrs;
  std::vector<const void*> blobs;
  std::vector<size_t> bytes;
  try {
    hdrs.resize(n);
    blobs.resize(n);
    bytes.resize(n);
  } catch (const std::bad_alloc&) {
    return fail(ctx, ZKIR_E_NOMEM, "host bookkeeping");
  }
  for (uint32_t k = 0; k < n; k++) {
    if (!tapes[k]) return fail(ctx, ZKIR_E_INVALID, "null tape handle");
    hdrs[k] = tapes[k]->tape.hdr;
    blobs[k] = tapes[k]->tape.blob.data();
    bytes[k] = tapes[k]->tape.blob.size();
  }
  // a tape handle comes from lower or from zkir_tape_from_blob, which validated it: no second look at every op
  return upload_impl(ctx, n, hdrs.data(), blobs.data(), bytes.data(), out, true);
}

static int32_t upload_impl(zkir_ctx* ctx, uint32_t n, const zkir_tape_hdr* hdrs, const void* const* blobs,
                           const size_t* bytes, zkir_tape_id* out, bool validated) {
  std::string why;
  // device layout of the batch: every tape's ops, then (256-byte aligned) every tape's constants, so that
  // one kernel converts all constants, then (256-byte aligned) one descriptor per tape: the tape evaluated alone
  std::vector<DevTape> tapes;
  size_t ops_total = 0, consts_total = 0;
  try {
    tapes.resize(n);
  } catch (const std::bad_alloc&) {
    return fail(ctx, ZKIR_E_NOMEM, "host bookkeeping");
  }
  for (uint32_t k = 0; k < n; k++) {
    if (!validated && !zkir::check_tape(hdrs[k], blobs[k], bytes[k], &why))
      return fail(ctx, ZKIR_E_INVALID, (n > 1 ? "bad tape " + std::to_string(k) + ": " : std::string("bad tape: ")) + why);
    DevTape& t = tapes[k];
    t.hdr = hdrs[k];
    t.full = tape_is_full(hdrs[k], blobs[k]);
    {  // which constant, if any, is 1 (the pool of a lowered tape holds every value once; the first match is taken)
      const uint32_t* cw = (const uint32_t*)((const char*)blobs[k] + (size_t)hdrs[k].n_ops * sizeof(zkir_op));
      for (uint32_t c = 0; c < hdrs[k].n_consts && t.one_word == 0xffffffffu; c++) {
        uint32_t rest = 0;
        for (int l = 1; l < 8; l++) rest |= cw[(size_t)c * 8 + l];
        if (cw[(size_t)c * 8] == 1u && rest == 0) t.one_word = ZKIR_OPERAND(ZKIR_KIND_CONST, c);
      }
    }
    if (t.full && (hdrs[k].flags & ZKIR_TAPE_F_STRANDS))
      return fail(ctx, ZKIR_E_INVALID, "bad tape: strand tapes are made of + - * and data movement only");
    ops_total += (size_t)hdrs[k].n_ops * sizeof(zkir_op);
    consts_total += (size_t)hdrs[k].n_consts * 32;
  }
  if (consts_total / 32 >= (1ull << 32)) return fail(ctx, ZKIR_E_LIMIT, "too many constants in one upload");
  const size_t consts_off = (ops_total + 255) & ~(size_t)255;
  const size_t descs_off = (consts_off + consts_total + 255) & ~(size_t)255;
  const size_t staged = descs_off + (size_t)n * sizeof(TapeDesc);  // what the host stages and copies
  const size_t consts29_off = (staged + 255) & ~(size_t)255;       // written by the device: 48 bytes per constant
  const size_t total = consts29_off + consts_total / 32 * 48;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the batch is gathered in pinned memory the context keeps (a copy from it runs at the link's rate and needs no
  // second staging inside the runtime); the previous upload's copy was synchronous, so the buffer is free
  if (ctx->up_stage_bytes < staged) {
    if (ctx->up_stage) (void)hipHostFree(ctx->up_stage);
    ctx->up_stage = nullptr;
    ctx->up_stage_bytes = 0;
    const size_t sz = std::max<size_t>(staged + staged / 4, (size_t)1 << 20);
    if (hipHostMalloc(&ctx->up_stage, sz, hipHostMallocDefault) != hipSuccess) return fail(ctx, ZKIR_E_NOMEM, "pinned staging buffer");
    ctx->up_stage_bytes = sz;
  }
  char* const staging = (char*)ctx->up_stage;
  void* base = nullptr;
  int slab = -1;
  size_t mark = 0;
  int32_t rc = arena_alloc(ctx, total, &base, &slab, &mark);
  if (rc) return rc;
  std::vector<size_t> ops_at, consts_at;
  try {
    ops_at.resize(n);
    consts_at.resize(n);
  } catch (const std::bad_alloc&) {
    arena_rollback(ctx, slab, mark, total);
    return fail(ctx, ZKIR_E_NOMEM, "host bookkeeping");
  }
  size_t o = 0, c = consts_off;
  for (uint32_t k = 0; k < n; k++) {
    ops_at[k] = o;
    consts_at[k] = c;
    tapes[k].ops = (char*)base + o;
    tapes[k].consts = (char*)base + c;
    tapes[k].consts29 = (char*)base + consts29_off + (c - consts_off) / 32 * 48;
    tapes[k].desc = (char*)base + descs_off + (size_t)k * sizeof(TapeDesc);
    tapes[k].slab = slab;
    o += (size_t)hdrs[k].n_ops * sizeof(zkir_op);
    c += (size_t)hdrs[k].n_consts * 32;
  }
  auto gather = [&](uint32_t k0, uint32_t k1) {
    for (uint32_t k = k0; k < k1; k++) {
      const size_t ob = (size_t)hdrs[k].n_ops * sizeof(zkir_op), cb = (size_t)hdrs[k].n_consts * 32;
      if (ob) std::memcpy(staging + ops_at[k], blobs[k], ob);
      if (cb) std::memcpy(staging + consts_at[k], (const char*)blobs[k] + ob, cb);
      const TapeDesc d = make_desc(tapes[k], 0, 0, 0);
      std::memcpy(staging + descs_off + (size_t)k * sizeof(TapeDesc), &d, sizeof d);
    }
  };
  {  // big batches on a few threads: one thread copies at about 10 GB/s
    const unsigned hw = std::thread::hardware_concurrency();
    const uint32_t n_threads = staged < ((size_t)16 << 20) ? 1u : std::max(1u, std::min({4u, hw ? hw : 1u, n}));
    std::vector<std::thread> pool;
    const uint32_t per = (n + n_threads - 1) / n_threads;
    try {
      pool.reserve(n_threads);
      for (uint32_t t = 1; t < n_threads; t++)
        if (t * per < n) pool.emplace_back(gather, t * per, std::min(n, (t + 1) * per));
    } catch (const std::exception&) {  // fewer helpers than asked for (thread limit, memory): the rest is done here
    }
    const uint32_t started = (uint32_t)pool.size();
    gather(0, std::min(n, per));
    for (uint32_t t = started + 1; t < n_threads; t++)
      if (t * per < n) gather(t * per, std::min(n, (t + 1) * per));
    for (std::thread& th : pool) th.join();
  }
  // synchronous copy: the staging buffer is free again when it returns
  hipError_t e = hipMemcpy(base, staging, staged, hipMemcpyHostToDevice);
  // both forms of the constants: the 29-bit one is made from the canonical words first, then those are
  // converted in place
  if (e == hipSuccess)
    e = zkir::launch_consts_to_mont29((const uint32_t*)((char*)base + consts_off), (uint32_t*)((char*)base + consts29_off),
                                      (uint32_t)(consts_total / 32), nullptr);
  if (e == hipSuccess)
    e = zkir::launch_consts_to_mont((uint32_t*)((char*)base + consts_off), (uint32_t)(consts_total / 32), nullptr);
  if (e == hipSuccess) e = hipEventRecord(ctx->upload_done, nullptr);
  if (e != hipSuccess) {
    arena_rollback(ctx, slab, mark, total);
    return hip_fail(ctx, e, "tape upload");
  }
  try {
    for (uint32_t k = 0; k < n; k++) ctx->tapes[ctx->next_id + k] = tapes[k];
  } catch (const std::bad_alloc&) {
    for (uint32_t k = 0; k < n; k++) ctx->tapes.erase(ctx->next_id + k);
    arena_rollback(ctx, slab, mark, total);
    return fail(ctx, ZKIR_E_NOMEM, "host bookkeeping");
  }
  ctx->slabs[slab].live += (int)n - 1;  // arena_alloc counted one; every tape is released on its own
  ctx->upload_pending = true;
  for (uint32_t k = 0; k < n; k++) out[k] = ctx->next_id++;
  return ZKIR_OK;
}

int32_t zkir_tape_release(zkir_ctx* ctx, zkir_tape_id id) {
  if (!ctx) return ZKIR_E_INVALID;
  auto it = ctx->tapes.find(id);
  if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->launched_since_sync) {  // no launch may still be reading the tape
    const uint64_t used = it->second.last_launch;
    if (!ctx->unchained && used <= ctx->done_seq) {
      // never launched, or its last launch is known to be over
    } else if (!ctx->unchained && ctx->launch_seq - used < (uint64_t)zkir_ctx::kLaunchRing) {
      // wait for the launch that read it: launches sent since (the next group of a pipelined caller) keep running
      HIP_TRY(ctx, hipEventSynchronize(ctx->launch_ev[(used - 1) % zkir_ctx::kLaunchRing]));
      ctx->done_seq = std::max(ctx->done_seq, used);
    } else {
      HIP_TRY(ctx, hipDeviceSynchronize());
      note_device_idle(ctx);
    }
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
  const int32_t rc = run_one(ctx, it->second, inputs_soa_dev, N, outputs_soa_dev, result_dev, result_dev + 1, s, nullptr, 0);
  if (rc == ZKIR_OK && N) it->second.last_launch = ctx->launch_seq;
  return rc;
}

int32_t zkir_eval_batch_dev_noalloc(zkir_ctx* ctx, zkir_tape_id tape, const uint32_t* inputs_soa_dev, uint64_t N,
                                    uint32_t* outputs_soa_dev, uint64_t* result_dev, void* stream) {
  if (!ctx || !result_dev) return ZKIR_E_INVALID;
  auto it = ctx->tapes.find(tape);
  if (it == ctx->tapes.end()) return fail(ctx, ZKIR_E_INVALID, "unknown tape id");
  if (it->second.hdr.n_inputs && N && !inputs_soa_dev) return fail(ctx, ZKIR_E_INVALID, "inputs required");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  const bool capturing = stream_is_capturing(s);
  ScratchOwner* owner = ctx->device < kMaxDevices ? &g_scratch_owner[ctx->device] : nullptr;
  std::unique_lock<std::mutex> guard(g_scratch_lock, std::defer_lock);
  if (owner && !capturing) {
    guard.lock();
    if (owner->recorded && owner->ctx != ctx && hipEventQuery(owner->done) == hipErrorNotReady)
      return fail(ctx, ZKIR_E_BUSY, "another context's zkir_eval_batch_dev_noalloc launch is still running on this GPU's scratch array");
  }
  // the result words and the counter are initialised by run_one; slot rows need no clearing (a tape writes a slot before it reads it)
  const int32_t rc = run_one(ctx, it->second, inputs_soa_dev, N, outputs_soa_dev, result_dev, result_dev + 1, s, ctx->static_scratch,
                             (size_t)ZKIR_NOALLOC_SCRATCH_BYTES);
  if (rc == ZKIR_OK && N) it->second.last_launch = ctx->launch_seq;
  if (rc == ZKIR_OK && owner && !capturing && owner->done) {
    if (hipEventRecord(owner->done, s) == hipSuccess) {
      owner->ctx = ctx;
      owner->recorded = true;
    }
  }
  return rc;
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
                            uint64_t N_per_tape, uint64_t* per_tape_unsat_dev, uint64_t* result_dev, void* stream) {
  return zkir_eval_multi_dev_faults(ctx, tapes, n_tapes, inputs_dev, N_per_tape, per_tape_unsat_dev, nullptr, result_dev, stream);
}

int32_t zkir_eval_multi_dev_faults(zkir_ctx* ctx, const zkir_tape_id* tapes, uint32_t n_tapes, const uint32_t* inputs_dev,
                                   uint64_t N_per_tape, uint64_t* per_tape_unsat_dev, uint64_t* per_tape_faults_dev,
                                   uint64_t* result_dev, void* stream) {
  if (!ctx || (!tapes && n_tapes) || (!per_tape_unsat_dev && n_tapes)) return ZKIR_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream;
  // without a caller buffer the statistics land in the context's own words (read by the host-pointer form)
  uint64_t* res = result_dev ? result_dev : (uint64_t*)ctx->result_buf;
  HIP_TRY(ctx, init_result(res, s));
  if (n_tapes == 0) return ZKIR_OK;
  HIP_TRY(ctx, hipMemsetAsync(per_tape_unsat_dev, 0, (size_t)n_tapes * sizeof(uint64_t), s));
  if (per_tape_faults_dev) HIP_TRY(ctx, hipMemsetAsync(per_tape_faults_dev, 0, (size_t)n_tapes * sizeof(uint64_t), s));
  return run_tapes(ctx, tapes, n_tapes, inputs_dev, N_per_tape, nullptr, per_tape_unsat_dev, res + 1, per_tape_faults_dev, s);
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
                           nullptr, nullptr);
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

// which: the multiply instruction of the integer probe; fr: 0 = that probe, 1 = the 8 x 32-bit multiplier, 2 = the 9 x 29-bit one
static int32_t time_probe(zkir_ctx* ctx, int which, int fr, double* out) {
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
    return fr == 2 ? zkir::launch_fr29_mul_rate(buf, blocks, iters, nullptr)
           : fr  ? zkir::launch_fr_mul_rate(buf, blocks, iters, nullptr)
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
  return time_probe(ctx, which, 0, lane_ops_per_s);
}

int32_t zkir_measure_fr_mul_rate(zkir_ctx* ctx, double* fr_mul_per_s) { return time_probe(ctx, 0, 1, fr_mul_per_s); }

int32_t zkir_measure_fr_mul_rate29(zkir_ctx* ctx, double* fr_mul_per_s) { return time_probe(ctx, 0, 2, fr_mul_per_s); }

}  // extern "C"
