// RCCL from inside the library (include/amg_hip.h "communicator"): a C / C++ caller
// of the drop-in gets the sharded V-cycle without Python.  librccl is opened at run time (dlopen:
// libamg_hip.so itself links the HIP runtime only, and a process that already carries an RCCL --
// PyTorch bundles one -- keeps using that copy); the communicator is bootstrapped from a 128-byte
// unique id that rank 0 makes and the caller distributes (MPI_Bcast, torch.distributed, a file).
// One sharded V-cycle is then ONE call that enqueues, on the solver's stream and with no host
// synchronisation: a grouped ncclSend / ncclRecv pair per neighbour (halo of the level-0
// solution), the captured down-legs, one ncclAllGather (right-hand side of the first replicated
// level), the replicated rest, the captured up-legs.
//
// Only the public C ABI of the solver is used here (amg_hip_slab_run, amg_hip_window_run,
// amg_hip_vec_dev_ptr, ...): this file is an in-library CLIENT of the same interface
// window_vcycle.py / slab_vcycle.py drive through torch.distributed.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/amg_hip.h"

namespace amg_hip {
amg_hip_status fail_status(amg_hip_status st, const std::string& msg);  // solver.cpp: sets amg_hip_last_error
}

namespace {

// the handful of RCCL entry points (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220,
// ncclCommDestroy :260, ncclAllGather :678, ncclSend :700, ncclRecv :722, ncclGroupStart/End :923-933)
typedef struct { char internal[128]; } rcclUniqueId;
typedef void* rcclComm_t;
enum { RCCL_FLOAT64 = 8 };  // ncclDouble / ncclFloat64
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId*) = nullptr;
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl R;
  static bool tried = false;
  if (tried) return &R;
  tried = true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  // a copy the process already carries first (RTLD_NOLOAD), else a fresh one
  for (const char* n : names)
    if ((R.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!R.lib)
    for (const char* n : names)
      if ((R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!R.lib) {
    R.why = std::string("librccl.so could not be opened: ") + (dlerror() ? dlerror() : "?");
    return &R;
  }
  auto sym = [&](const char* n) -> void* {
    void* p = dlsym(R.lib, n);
    if (!p && R.why.empty()) R.why = std::string("librccl.so lacks ") + n;
    return p;
  };
  R.GetUniqueId = (int (*)(rcclUniqueId*))sym("ncclGetUniqueId");
  R.CommInitRank = (int (*)(rcclComm_t*, int, rcclUniqueId, int))sym("ncclCommInitRank");
  R.CommDestroy = (int (*)(rcclComm_t))sym("ncclCommDestroy");
  R.AllGather = (int (*)(const void*, void*, size_t, int, rcclComm_t, hipStream_t))sym("ncclAllGather");
  R.Send = (int (*)(const void*, size_t, int, int, rcclComm_t, hipStream_t))sym("ncclSend");
  R.Recv = (int (*)(void*, size_t, int, int, rcclComm_t, hipStream_t))sym("ncclRecv");
  R.GroupStart = (int (*)())sym("ncclGroupStart");
  R.GroupEnd = (int (*)())sym("ncclGroupEnd");
  R.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  if (!R.why.empty()) {
    dlclose(R.lib);
    R.lib = nullptr;
  }
  return &R;
}

amg_hip_status rccl_fail(const char* what, int rc) {
  Rccl* R = rccl();
  return amg_hip::fail_status(AMG_HIP_ECOMM, std::string(what) + ": " +
                                                 (R->GetErrorString ? R->GetErrorString(rc) : "RCCL error"));
}
#define RCCL_TRY(call)                                                  \
  do {                                                                  \
    const int rc_ = (call);                                             \
    if (rc_ != 0) return rccl_fail(#call, rc_);                         \
  } while (0)
#define HIP_TRY_C(call)                                                                       \
  do {                                                                                        \
    const hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess)                                                                     \
      return amg_hip::fail_status(AMG_HIP_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

}  // namespace

struct amg_hip_comm {
  rcclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  double* stage = nullptr;     // window cycle: this rank's block of f_k, then every rank's
  double* gathered = nullptr;
  size_t stage_doubles = 0, gathered_doubles = 0;
};

extern "C" {

amg_hip_status amg_hip_comm_unique_id(uint8_t id[AMG_HIP_COMM_ID_BYTES]) {
  if (!id) return amg_hip::fail_status(AMG_HIP_EINVAL, "null id");
  Rccl* R = rccl();
  if (!R->lib) return amg_hip::fail_status(AMG_HIP_EUNSUPPORTED, R->why);
  rcclUniqueId u;
  RCCL_TRY(R->GetUniqueId(&u));
  static_assert(sizeof(u) == AMG_HIP_COMM_ID_BYTES, "unique id size");
  std::memcpy(id, &u, sizeof(u));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_comm_create(const uint8_t id[AMG_HIP_COMM_ID_BYTES], int32_t rank, int32_t world,
                                   int32_t device, amg_hip_comm** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world)
    return amg_hip::fail_status(AMG_HIP_EINVAL, "amg_hip_comm_create: bad argument");
  *out = nullptr;
  Rccl* R = rccl();
  if (!R->lib) return amg_hip::fail_status(AMG_HIP_EUNSUPPORTED, R->why);
  if (device >= 0) HIP_TRY_C(hipSetDevice(device));
  else HIP_TRY_C(hipGetDevice(&device));
  rcclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  amg_hip_comm* c = new amg_hip_comm;
  c->rank = rank;
  c->world = world;
  c->device = device;
  const int rc = R->CommInitRank(&c->comm, world, u, rank);
  if (rc != 0) {
    delete c;
    return rccl_fail("ncclCommInitRank", rc);
  }
  *out = c;
  return AMG_HIP_OK;
}

void amg_hip_comm_destroy(amg_hip_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stage) (void)hipFree(c->stage);
  if (c->gathered) (void)hipFree(c->gathered);
  Rccl* R = rccl();
  if (c->comm && R->CommDestroy) (void)R->CommDestroy(c->comm);
  delete c;
}

int32_t amg_hip_comm_rank(const amg_hip_comm* c) { return c ? c->rank : -1; }
int32_t amg_hip_comm_world(const amg_hip_comm* c) { return c ? c->world : 0; }

// grouped send / recv with rank-1 and rank+1 (counts in doubles; a null pointer or a zero count
// skips that direction) -- the halo exchange both sharded cycles start with
amg_hip_status amg_hip_comm_neighbor_exchange(amg_hip_comm* c, const double* send_prev, int64_t n_send_prev,
                                              double* recv_prev, int64_t n_recv_prev, const double* send_next,
                                              int64_t n_send_next, double* recv_next, int64_t n_recv_next,
                                              void* stream) {
  if (!c) return amg_hip::fail_status(AMG_HIP_EINVAL, "null communicator");
  Rccl* R = rccl();
  hipStream_t st = (hipStream_t)stream;
  const bool p = c->rank > 0, n = c->rank + 1 < c->world;
  if (!p && !n) return AMG_HIP_OK;
  RCCL_TRY(R->GroupStart());
  int rc = 0;
  if (p && recv_prev && n_recv_prev > 0 && !rc) rc = R->Recv(recv_prev, (size_t)n_recv_prev, RCCL_FLOAT64, c->rank - 1, c->comm, st);
  if (p && send_prev && n_send_prev > 0 && !rc) rc = R->Send(send_prev, (size_t)n_send_prev, RCCL_FLOAT64, c->rank - 1, c->comm, st);
  if (n && recv_next && n_recv_next > 0 && !rc) rc = R->Recv(recv_next, (size_t)n_recv_next, RCCL_FLOAT64, c->rank + 1, c->comm, st);
  if (n && send_next && n_send_next > 0 && !rc) rc = R->Send(send_next, (size_t)n_send_next, RCCL_FLOAT64, c->rank + 1, c->comm, st);
  const int rc2 = R->GroupEnd();
  if (rc) return rccl_fail("ncclSend / ncclRecv", rc);
  if (rc2) return rccl_fail("ncclGroupEnd", rc2);
  return AMG_HIP_OK;
}

// out = [block of rank 0 | block of rank 1 | ...]; in may be out + rank * count (in place)
amg_hip_status amg_hip_comm_all_gather(amg_hip_comm* c, const double* in, double* out, int64_t count,
                                       void* stream) {
  if (!c || !in || !out || count < 0) return amg_hip::fail_status(AMG_HIP_EINVAL, "amg_hip_comm_all_gather: bad argument");
  Rccl* R = rccl();
  RCCL_TRY(R->AllGather(in, out, (size_t)count, RCCL_FLOAT64, c->comm, (hipStream_t)stream));
  return AMG_HIP_OK;
}

// ---- one slab-sharded V-cycle (amg_hip_slab_setup has run with this communicator's rank / world) ----
amg_hip_status amg_hip_slab_cycle(amg_hip_solver* s, amg_hip_comm* c, const amg_hip_slab_info* info) {
  if (!s || !c || !info) return amg_hip::fail_status(AMG_HIP_EINVAL, "amg_hip_slab_cycle: null argument");
  void* st = nullptr;
  amg_hip_status r = amg_hip_get_stream(s, &st);
  if (r != AMG_HIP_OK) return r;
  const int64_t m = info->pitch0, H = (int64_t)info->halo_lines * m;
  const int64_t a = info->line_begin * m, b = info->line_end * m;
  double* u = info->u0;
  if (c->world > 1) {
    // 1. halo lines of u0, in place in the full-size vector
    r = amg_hip_comm_neighbor_exchange(c, u + a, H, u + a - H, H, u + b - H, H, u + b, H, st);
    if (r != AMG_HIP_OK) return r;
  }
  if ((r = amg_hip_slab_run(s, 1)) != AMG_HIP_OK) return r;
  if (c->world > 1) {
    // 2. all-gather of the first replicated level's right-hand side, in place
    const int64_t blk = info->chunk_lines * info->gather_pitch;
    r = amg_hip_comm_all_gather(c, info->f_gather + (int64_t)c->rank * blk, info->f_gather, blk, st);
    if (r != AMG_HIP_OK) return r;
  }
  if ((r = amg_hip_slab_run(s, 2)) != AMG_HIP_OK) return r;
  return amg_hip_slab_run(s, 3);
}

// ---- one window-sharded V-cycle: window solver w (levels < k), tail solver t (levels >= k) ----
amg_hip_status amg_hip_window_cycle(amg_hip_solver* w, amg_hip_solver* t, amg_hip_comm* c,
                                    const amg_hip_window_plan* p) {
  if (!w || !t || !c || !p) return amg_hip::fail_status(AMG_HIP_EINVAL, "amg_hip_window_cycle: null argument");
  void *st = nullptr, *st2 = nullptr;
  amg_hip_status r = amg_hip_get_stream(w, &st);
  if (r != AMG_HIP_OK) return r;
  if ((r = amg_hip_get_stream(t, &st2)) != AMG_HIP_OK) return r;
  if (st != st2) return amg_hip::fail_status(AMG_HIP_EINVAL, "amg_hip_window_cycle: both solvers must share one stream");
  hipStream_t s = (hipStream_t)st;
  const int k = amg_hip_n_levels(w) - 1;
  void *u0v = nullptr, *fkv = nullptr, *ukv = nullptr, *tfv = nullptr, *tuv = nullptr;
  int64_t n0 = 0, nkw = 0, nk = 0;
  if ((r = amg_hip_vec_dev_ptr(w, 0, 0, &u0v, &n0)) != AMG_HIP_OK) return r;
  if ((r = amg_hip_vec_dev_ptr(w, k, 1, &fkv, &nkw)) != AMG_HIP_OK) return r;
  if ((r = amg_hip_vec_dev_ptr(w, k, 0, &ukv, nullptr)) != AMG_HIP_OK) return r;
  if ((r = amg_hip_vec_dev_ptr(t, 0, 1, &tfv, &nk)) != AMG_HIP_OK) return r;
  if ((r = amg_hip_vec_dev_ptr(t, 0, 0, &tuv, nullptr)) != AMG_HIP_OK) return r;
  double *u0 = (double*)u0v, *fk = (double*)fkv, *uk = (double*)ukv, *tf = (double*)tfv, *tu = (double*)tuv;
  if (p->own_k_off < 0 || p->own_k_cnt < 0 || p->own_k_off + p->own_k_cnt > nkw || p->own_k_cnt > p->block_k ||
      p->block_k * (int64_t)c->world < nk || p->uk_off < 0 || p->uk_off + nkw > nk ||
      p->recv_prev_cnt < 0 || p->recv_next_cnt < 0 || p->own0_off - p->recv_prev_cnt < 0 ||
      p->own0_end + p->recv_next_cnt > n0 || p->send_prev_cnt > p->own0_end - p->own0_off ||
      p->send_next_cnt > p->own0_end - p->own0_off)
    return amg_hip::fail_status(AMG_HIP_EINVAL, "amg_hip_window_cycle: the plan does not fit the solvers");
  HIP_TRY_C(hipSetDevice(c->device));
  const size_t need_s = (size_t)p->block_k, need_g = (size_t)p->block_k * (size_t)c->world;
  if (c->stage_doubles < need_s) {
    if (c->stage) (void)hipFree(c->stage);
    c->stage = nullptr;
    HIP_TRY_C(hipMalloc((void**)&c->stage, sizeof(double) * need_s));
    HIP_TRY_C(hipMemset(c->stage, 0, sizeof(double) * need_s));
    c->stage_doubles = need_s;
  }
  if (c->gathered_doubles < need_g) {
    if (c->gathered) (void)hipFree(c->gathered);
    c->gathered = nullptr;
    HIP_TRY_C(hipMalloc((void**)&c->gathered, sizeof(double) * need_g));
    c->gathered_doubles = need_g;
  }
  // 1. halo units of the level-0 solution, in place in the window
  if (c->world > 1) {
    r = amg_hip_comm_neighbor_exchange(c, u0 + p->own0_off, p->send_prev_cnt, u0 + p->own0_off - p->recv_prev_cnt,
                                       p->recv_prev_cnt, u0 + p->own0_end - p->send_next_cnt, p->send_next_cnt,
                                       u0 + p->own0_end, p->recv_next_cnt, st);
    if (r != AMG_HIP_OK) return r;
  }
  // 2. down-legs of the distributed levels (multigrid.hpp:265-283)
  if ((r = amg_hip_window_run(w, 1)) != AMG_HIP_OK) return r;
  // 3. owned rows of f_k from every rank -> the tail's right-hand side
  HIP_TRY_C(hipMemcpyAsync(c->stage, fk + p->own_k_off, sizeof(double) * (size_t)p->own_k_cnt,
                           hipMemcpyDeviceToDevice, s));
  if (c->world > 1) {
    if ((r = amg_hip_comm_all_gather(c, c->stage, c->gathered, p->block_k, st)) != AMG_HIP_OK) return r;
    HIP_TRY_C(hipMemcpyAsync(tf, c->gathered, sizeof(double) * (size_t)nk, hipMemcpyDeviceToDevice, s));
  } else {
    HIP_TRY_C(hipMemcpyAsync(tf, c->stage, sizeof(double) * (size_t)nk, hipMemcpyDeviceToDevice, s));
  }
  // 4. the replicated levels from a zero guess (multigrid.hpp:278), coarse solve included
  if ((r = amg_hip_zero_vec(t, 0, 0)) != AMG_HIP_OK) return r;
  if ((r = amg_hip_vcycle(t)) != AMG_HIP_OK) return r;
  // 5. the window of u_k, then the up-legs (:291-302)
  HIP_TRY_C(hipMemcpyAsync(uk, tu + p->uk_off, sizeof(double) * (size_t)nkw, hipMemcpyDeviceToDevice, s));
  return amg_hip_window_run(w, 3);
}

}  // extern "C"
