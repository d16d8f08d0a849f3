// qinco_gather_codes -- the end-of-job collective of the path for hosts that are not Python: every rank's
// (n_r, M) codes to one root over RCCL send / recv (point-to-point: on MI355X each peer reaches the root over its own xGMI
// link).  The reference has no payload collective at all (per-rank part files, search_tasks.py:119-134); the Python host uses
// torch.distributed for the same exchange (qinco_amd/encode_db.py: gather_codes).
//
// RCCL is not linked: a process must use ONE RCCL (PyTorch wheels bundle their own librccl.so next to /opt/rocm's), and the
// communicator comes from the host, so the entry points are looked up at call time -- first among the libraries the process
// has already loaded (whoever created the communicator), then librccl.so.1 by name.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstddef>
#include <cstdint>

#include "abi_util.hpp"

namespace {
typedef int (*nccl_p2p_fn)(void*, size_t, int, int, void*, hipStream_t);   // ncclSend / ncclRecv (const void* / void* buffer)
typedef int (*nccl_void_fn)();
typedef const char* (*nccl_err_fn)(int);

struct Rccl {
  nccl_p2p_fn send = nullptr, recv = nullptr;
  nccl_void_fn group_start = nullptr, group_end = nullptr;
  nccl_err_fn err = nullptr;
  bool ok() const { return send && recv && group_start && group_end; }
};

void* lookup(const char* name) {
  if (void* p = dlsym(RTLD_DEFAULT, name)) return p;
  static void* lib = [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return h ? h : dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  }();
  return lib ? dlsym(lib, name) : nullptr;
}

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl t;
    t.send = reinterpret_cast<nccl_p2p_fn>(lookup("ncclSend"));
    t.recv = reinterpret_cast<nccl_p2p_fn>(lookup("ncclRecv"));
    t.group_start = reinterpret_cast<nccl_void_fn>(lookup("ncclGroupStart"));
    t.group_end = reinterpret_cast<nccl_void_fn>(lookup("ncclGroupEnd"));
    t.err = reinterpret_cast<nccl_err_fn>(lookup("ncclGetErrorString"));
    return t;
  }();
  return r;
}
constexpr int kNcclUint8 = 1;   // rccl.h: ncclUint8
}  // namespace

#define fail qinco::abi_fail
#define RCCL_TRY(expr)                                                                                        \
  do {                                                                                                        \
    const int _r = (expr);                                                                                    \
    if (_r != 0) return fail(QINCO_ERR_HIP, "%s failed: %s", #expr, R.err ? R.err(_r) : "RCCL error");        \
  } while (0)

extern "C" int qinco_gather_codes(const void* codes_local, int64_t n_local, int32_t M, int code_dtype, void* out,
                                  const int64_t* counts, int32_t world, int32_t rank, int32_t root, void* nccl_comm, void* stream) {
  if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || M < 1 || !counts)
    return fail(QINCO_ERR_INVALID, "qinco_gather_codes: bad world / rank / root / M / counts");
  if (code_dtype < 0 || code_dtype > 2) return fail(QINCO_ERR_INVALID, "qinco_gather_codes: bad code dtype %d", code_dtype);
  if (n_local != counts[rank]) return fail(QINCO_ERR_INVALID, "qinco_gather_codes: n_local = %ld but counts[%d] = %ld", (long)n_local, rank, (long)counts[rank]);
  const size_t esz = code_dtype == QINCO_CODE_I64 ? 8 : code_dtype == QINCO_CODE_I32 ? 4 : 1;
  const size_t rowb = (size_t)M * esz;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n_local > 0 && !codes_local) return fail(QINCO_ERR_INVALID, "qinco_gather_codes: null codes_local");
  int64_t total = 0;
  for (int r = 0; r < world; ++r) {
    if (counts[r] < 0) return fail(QINCO_ERR_INVALID, "qinco_gather_codes: counts[%d] = %ld", r, (long)counts[r]);
    total += counts[r];
  }
  if (rank == root && !out && total > 0) return fail(QINCO_ERR_INVALID, "qinco_gather_codes: the root needs an output buffer");
  if (world == 1 && !nccl_comm) {   // a job of one rank without a communicator: a device copy
    if (n_local > 0) HIP_TRY(hipMemcpyAsync(out, codes_local, (size_t)n_local * rowb, hipMemcpyDeviceToDevice, st));
    return QINCO_OK;
  }
  if (!nccl_comm) return fail(QINCO_ERR_INVALID, "qinco_gather_codes: null communicator");
  const Rccl& R = rccl();
  if (!R.ok()) return fail(QINCO_ERR_UNSUPPORTED, "qinco_gather_codes: no RCCL in this process (ncclSend / ncclRecv not found; librccl.so.1 not loadable)");
  if (world == 1) {
    // A communicator of ONE rank was handed in: the shard still travels through RCCL -- one grouped ncclSend to self +
    // ncclRecv from self, the same group / send / recv / stream sequence as the many-rank path below.  This is what a 1-GPU
    // box can execute of that path on the real library (tests/test_multi_gpu.py::test_rccl_world_of_one).
    if (n_local > 0) {
      RCCL_TRY(R.group_start());
      int grc = R.send(const_cast<void*>(codes_local), (size_t)n_local * rowb, kNcclUint8, 0, nccl_comm, st);
      if (grc == 0) grc = R.recv(out, (size_t)n_local * rowb, kNcclUint8, 0, nccl_comm, st);
      if (grc != 0) {
        (void)R.group_end();
        RCCL_TRY(grc);
      }
      RCCL_TRY(R.group_end());
    }
    return QINCO_OK;
  }
  RCCL_TRY(R.group_start());
  // (an error inside the group must still CLOSE it -- an open group on this thread would swallow every later collective of the process)
  int grc = 0;
  if (rank != root) {
    if (n_local > 0) grc = R.send(const_cast<void*>(codes_local), (size_t)n_local * rowb, kNcclUint8, root, nccl_comm, st);
  } else {
    size_t off = 0;
    for (int r = 0; r < world && grc == 0; ++r) {
      const size_t bytes = (size_t)counts[r] * rowb;
      if (r != root && bytes) grc = R.recv(static_cast<char*>(out) + off, bytes, kNcclUint8, r, nccl_comm, st);
      off += bytes;
    }
  }
  if (grc != 0) {
    (void)R.group_end();
    RCCL_TRY(grc);
  }
  RCCL_TRY(R.group_end());
  if (rank == root && n_local > 0) {   // the root's own shard: a device copy into its place
    size_t off = 0;
    for (int r = 0; r < root; ++r) off += (size_t)counts[r] * rowb;
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(out) + off, codes_local, (size_t)n_local * rowb, hipMemcpyDeviceToDevice, st));
  }
  return QINCO_OK;
}

// Which RCCL this process's qinco_gather_codes calls: the path of the shared object that ncclSend resolved to (dladdr) --
// PyTorch wheels carry their own librccl.so next to /opt/rocm's, and a communicator must be used with the library that made it.
extern "C" int qinco_rccl_library(char* path, size_t cap) {
  if (!path || cap == 0) return fail(QINCO_ERR_INVALID, "qinco_rccl_library: null buffer");
  const Rccl& R = rccl();
  if (!R.ok()) return fail(QINCO_ERR_UNSUPPORTED, "qinco_rccl_library: no RCCL in this process (ncclSend / ncclRecv not found; librccl.so.1 not loadable)");
  Dl_info info;
  if (!dladdr(reinterpret_cast<void*>(R.send), &info) || !info.dli_fname) return fail(QINCO_ERR_UNSUPPORTED, "qinco_rccl_library: dladdr could not place ncclSend");
  size_t n = 0;
  while (info.dli_fname[n] && n + 1 < cap) { path[n] = info.dli_fname[n]; ++n; }
  path[n] = 0;
  return QINCO_OK;
}
