// The campaign's one exchange step behind the C ABI, for a host process that drives several GPUs itself (the shape
// of the reference's only caller, src/main.rs:81-86, is one process): RCCL communicators made by ncclCommInitAll over
// the given devices, one all-reduce (sum) of the counters and one all-gather of the padded divergent-id lists, count
// first.  RCCL has no OR reduction; the shards of a campaign are disjoint, so sums do.  The payload is a few hundred
// bytes to some kilobytes: the exchange is bound by latency, and nothing about it is tuned.
// RCCL is loaded when the first reducer is made (dlopen), so the library itself does not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../../include/zkir_hip.h"

namespace {

// the part of the RCCL API used here (rccl.h: ncclCommInitAll :236, ncclAllReduce :611, ncclAllGather :678)
typedef void* comm_t;
struct Rccl {
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
constexpr int kNcclSum = 0, kNcclUint32 = 3, kNcclUint64 = 5;  // rccl.h:448, :462, :464

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    // a copy the process already holds first (a host that imported PyTorch has its wheel's RCCL, bound to the same
    // HIP runtime this library is bound to), then the system's
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return x;
    x.CommInitAll = (decltype(x.CommInitAll))dlsym(h, "ncclCommInitAll");
    x.CommDestroy = (decltype(x.CommDestroy))dlsym(h, "ncclCommDestroy");
    x.AllReduce = (decltype(x.AllReduce))dlsym(h, "ncclAllReduce");
    x.AllGather = (decltype(x.AllGather))dlsym(h, "ncclAllGather");
    x.GroupStart = (decltype(x.GroupStart))dlsym(h, "ncclGroupStart");
    x.GroupEnd = (decltype(x.GroupEnd))dlsym(h, "ncclGroupEnd");
    x.GetErrorString = (decltype(x.GetErrorString))dlsym(h, "ncclGetErrorString");
    x.ok = x.CommInitAll && x.CommDestroy && x.AllReduce && x.AllGather && x.GroupStart && x.GroupEnd && x.GetErrorString;
    return x;
  }();
  return r;
}

}  // namespace

struct zkir_reducer {
  std::vector<int> devices;
  std::vector<comm_t> comms;
  std::vector<hipStream_t> streams;
  // per device: counters in / out, ids in ([1 + capacity]: count, then the padded payload) / out ([n][1 + capacity])
  std::vector<uint64_t*> cnt_in, cnt_out;
  std::vector<uint32_t*> ids_in, ids_out;
  uint32_t n_counters = 0, capacity = 0;
  std::string err;
};

namespace {
int32_t fail(zkir_reducer* r, int32_t code, const std::string& m) {
  if (r) r->err = m;
  return code;
}
void release(zkir_reducer* r) {
  const Rccl& R = rccl();
  for (size_t k = 0; k < r->devices.size(); k++) {
    (void)hipSetDevice(r->devices[k]);
    if (k < r->comms.size() && r->comms[k] && R.ok) (void)R.CommDestroy(r->comms[k]);
    if (k < r->streams.size() && r->streams[k]) (void)hipStreamDestroy(r->streams[k]);
    for (auto* v : {(void*)(k < r->cnt_in.size() ? r->cnt_in[k] : nullptr), (void*)(k < r->cnt_out.size() ? r->cnt_out[k] : nullptr),
                    (void*)(k < r->ids_in.size() ? r->ids_in[k] : nullptr), (void*)(k < r->ids_out.size() ? r->ids_out[k] : nullptr)})
      if (v) (void)hipFree(v);
  }
  delete r;
}
}  // namespace

extern "C" {

int32_t zkir_reducer_create(const int32_t* devices, uint32_t n_devices, uint32_t n_counters, uint32_t id_capacity, zkir_reducer** out) {
  if (!out) return ZKIR_E_INVALID;
  *out = nullptr;
  if (!devices || n_devices == 0 || n_devices > 64 || n_counters == 0 || n_counters > 4096 || id_capacity > (1u << 20)) return ZKIR_E_INVALID;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZKIR_E_NODEVICE;
  for (uint32_t k = 0; k < n_devices; k++) {
    if (devices[k] < 0 || devices[k] >= count) return ZKIR_E_INVALID;
    for (uint32_t j = 0; j < k; j++)
      if (devices[j] == devices[k]) return ZKIR_E_INVALID;  // one rank per device
  }
  const Rccl& R = rccl();
  if (!R.ok) return ZKIR_E_UNSUPPORTED;  // no RCCL library to load
  zkir_reducer* r = new (std::nothrow) zkir_reducer();
  if (!r) return ZKIR_E_NOMEM;
  try {
    r->devices.assign(devices, devices + n_devices);
    r->comms.assign(n_devices, nullptr);
    r->streams.assign(n_devices, nullptr);
    r->cnt_in.assign(n_devices, nullptr);
    r->cnt_out.assign(n_devices, nullptr);
    r->ids_in.assign(n_devices, nullptr);
    r->ids_out.assign(n_devices, nullptr);
  } catch (const std::bad_alloc&) {
    delete r;
    return ZKIR_E_NOMEM;
  }
  r->n_counters = n_counters;
  r->capacity = id_capacity;
  for (uint32_t k = 0; k < n_devices; k++) {
    const size_t row = (size_t)(1 + id_capacity) * sizeof(uint32_t);
    if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&r->streams[k], hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&r->cnt_in[k], n_counters * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void**)&r->cnt_out[k], n_counters * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void**)&r->ids_in[k], row) != hipSuccess || hipMalloc((void**)&r->ids_out[k], row * n_devices) != hipSuccess) {
      release(r);
      return ZKIR_E_HIP;
    }
  }
  const int rc = R.CommInitAll(r->comms.data(), (int)n_devices, r->devices.data());
  if (rc != 0) {
    release(r);
    return ZKIR_E_HIP;
  }
  *out = r;
  return ZKIR_OK;
}

int32_t zkir_reducer_destroy(zkir_reducer* r) {
  if (r) release(r);
  return ZKIR_OK;
}

const char* zkir_reducer_last_error(const zkir_reducer* r) { return r ? r->err.c_str() : "null reducer"; }

int32_t zkir_reducer_reduce(zkir_reducer* r, const uint64_t* counters, const uint32_t* ids, const uint32_t* n_ids, uint64_t* totals,
                            uint32_t* all_ids, uint32_t* n_all_ids) {
  if (!r || !counters || !n_ids || !totals || !n_all_ids || (r->capacity && (!ids || !all_ids))) return ZKIR_E_INVALID;
  const Rccl& R = rccl();
  const uint32_t n = (uint32_t)r->devices.size(), cap = r->capacity, row = 1 + cap;
  std::vector<uint32_t> stage;
  try {
    stage.assign((size_t)row * n, 0xffffffffu);  // the padding of a list is all ones
  } catch (const std::bad_alloc&) {
    return fail(r, ZKIR_E_NOMEM, "host staging");
  }
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    r->err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
  };
  for (uint32_t k = 0; k < n; k++) {
    if (n_ids[k] > cap) return fail(r, ZKIR_E_LIMIT, "more divergent ids on one device than the reducer's capacity");
    stage[(size_t)k * row] = n_ids[k];
    if (n_ids[k]) std::memcpy(&stage[(size_t)k * row + 1], ids + (size_t)k * cap, (size_t)n_ids[k] * sizeof(uint32_t));
    if (!hip_ok(hipSetDevice(r->devices[k]), "hipSetDevice") ||
        !hip_ok(hipMemcpyAsync(r->cnt_in[k], counters + (size_t)k * r->n_counters, r->n_counters * sizeof(uint64_t), hipMemcpyHostToDevice, r->streams[k]), "copy counters") ||
        !hip_ok(hipMemcpyAsync(r->ids_in[k], &stage[(size_t)k * row], row * sizeof(uint32_t), hipMemcpyHostToDevice, r->streams[k]), "copy ids"))
      return ZKIR_E_HIP;
  }
  auto nccl_ok = [&](int rc, const char* what) {
    if (rc == 0) return true;
    r->err = std::string(what) + ": " + R.GetErrorString(rc);
    return false;
  };
  // one rank per device in one process: the calls of one collective go between GroupStart and GroupEnd
  if (!nccl_ok(R.GroupStart(), "ncclGroupStart")) return ZKIR_E_HIP;
  for (uint32_t k = 0; k < n; k++)
    if (!nccl_ok(R.AllReduce(r->cnt_in[k], r->cnt_out[k], r->n_counters, kNcclUint64, kNcclSum, r->comms[k], r->streams[k]), "ncclAllReduce")) {
      (void)R.GroupEnd();
      return ZKIR_E_HIP;
    }
  if (!nccl_ok(R.GroupEnd(), "ncclGroupEnd")) return ZKIR_E_HIP;
  if (!nccl_ok(R.GroupStart(), "ncclGroupStart")) return ZKIR_E_HIP;
  for (uint32_t k = 0; k < n; k++)
    if (!nccl_ok(R.AllGather(r->ids_in[k], r->ids_out[k], row, kNcclUint32, r->comms[k], r->streams[k]), "ncclAllGather")) {
      (void)R.GroupEnd();
      return ZKIR_E_HIP;
    }
  if (!nccl_ok(R.GroupEnd(), "ncclGroupEnd")) return ZKIR_E_HIP;
  // every device holds the same totals and the same gathered lists: the first one's are read back, the others are
  // waited for (their streams may not be left running into the next call's copies)
  std::vector<uint32_t> gathered;
  try {
    gathered.resize((size_t)row * n);
  } catch (const std::bad_alloc&) {
    return fail(r, ZKIR_E_NOMEM, "host staging");
  }
  for (uint32_t k = 0; k < n; k++) {
    if (!hip_ok(hipSetDevice(r->devices[k]), "hipSetDevice")) return ZKIR_E_HIP;
    if (k == 0 && (!hip_ok(hipMemcpyAsync(totals, r->cnt_out[0], r->n_counters * sizeof(uint64_t), hipMemcpyDeviceToHost, r->streams[0]), "read totals") ||
                   !hip_ok(hipMemcpyAsync(gathered.data(), r->ids_out[0], gathered.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, r->streams[0]), "read ids")))
      return ZKIR_E_HIP;
    if (!hip_ok(hipStreamSynchronize(r->streams[k]), "hipStreamSynchronize")) return ZKIR_E_HIP;
  }
  std::vector<uint32_t> all;
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t c = std::min(gathered[(size_t)k * row], cap);
    all.insert(all.end(), gathered.begin() + (size_t)k * row + 1, gathered.begin() + (size_t)k * row + 1 + c);
  }
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  *n_all_ids = (uint32_t)all.size();
  if (!all.empty()) std::memcpy(all_ids, all.data(), all.size() * sizeof(uint32_t));
  return ZKIR_OK;
}

}  // extern "C"
