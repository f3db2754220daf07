// TEST INFRASTRUCTURE (tests/test_gpu_dp_loopback.py) -- not part of the product, never loaded by kprn_amd on its own.
//
// An in-process stand-in for librccl with the six entry points kprn_amd/csrc/kprn_api.hip binds (ncclGetUniqueId, ncclCommInitRank,
// ncclAllGather, ncclCommDestroy, ncclCommCount, ncclGetErrorString).  Handed to kprn_dp_unique_id / kprn_dp_init as `rccl_path`, it lets W
// engine handles of ONE process on ONE GPU run the engine's own world > 1 exchange -- kprn_dp_exchange_begin / _finish, the in-place slot
// offsets of the gathered buffer, the dp_comm_stream hand-over, k_union_adam against W real id lists -- exactly as they will run under RCCL
// on W GPUs (no multi-GPU node was available to any round of this build).
//
// Semantics of ncclAllGather here: a rendezvous of the group's W ranks (each driven by its own host thread); once every rank has arrived,
// each rank queues on ITS stream: wait for peer p's "slot complete" event, copy p's send slot into recvbuff + p * bytes (device to device),
// for every p; then a second rendezvous after which every rank's stream also waits for all peers' copies -- a rank's send slot is not
// reused (the next step's pack) before every peer has read it, which is what completion of the real collective guarantees.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
constexpr int kMaxRanks = 64;
struct Slot { const void* send; void* recv; size_t bytes; hipStream_t stream; hipEvent_t ready; hipEvent_t copied; };
struct Group {
  int world = 0, refs = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  Slot slot[kMaxRanks];
  uint64_t calls = 0;
};
struct Comm { Group* g; int rank; hipEvent_t ready, copied; };
std::mutex g_mu;
std::map<std::string, Group*> g_groups;
uint64_t g_next_id = 1;
uint64_t g_allgathers = 0;

// all ranks of the group meet here; false = a rank never came (the caller reports an error instead of hanging the test)
bool rendezvous(Group* g, std::unique_lock<std::mutex>& lk) {
  const uint64_t my_gen = g->gen;
  if (++g->arrived == g->world) {
    g->arrived = 0;
    ++g->gen;
    g->cv.notify_all();
    return true;
  }
  return g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->gen != my_gen; });
}
}  // namespace

extern "C" {

int ncclGetUniqueId(void* id128) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id128, 0, 128);
  const uint64_t v = g_next_id++;
  memcpy(id128, "KPRNLOOP", 8);
  memcpy((char*)id128 + 8, &v, sizeof(v));
  return 0;
}

struct Id128 { char b[128]; };
int ncclCommInitRank(void** comm, int world, Id128 id, int rank) {
  if (!comm || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return 4;   // ncclInvalidArgument
  std::lock_guard<std::mutex> lk(g_mu);
  Group*& g = g_groups[std::string(id.b, 128)];
  if (!g) { g = new Group(); g->world = world; }
  if (g->world != world) return 4;
  ++g->refs;
  Comm* c = new Comm{g, rank, nullptr, nullptr};
  if (hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->copied, hipEventDisableTiming) != hipSuccess) return 1;
  *comm = c;
  return 0;
}

int ncclCommCount(void* comm, int* n) {
  if (!comm || !n) return 4;
  *n = ((Comm*)comm)->g->world;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->copied);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    --c->g->refs;   // (groups are small and few: left in the map for the life of the test process)
  }
  delete c;
  return 0;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case 0: return "success";
    case 1: return "loopback: HIP call failed";
    case 2: return "loopback: a rank never reached the collective (60 s)";
    case 4: return "loopback: invalid argument";
    default: return "loopback: error";
  }
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || !sendbuff || !recvbuff) return 4;
  const size_t esz = (dtype == 0 || dtype == 1) ? 1 : ((dtype == 6 /*half*/ || dtype == 9 /*bf16*/) ? 2 : ((dtype == 4 || dtype == 5 || dtype == 8) ? 8 : 4));
  const size_t bytes = count * esz;
  Group* g = c->g;
  const int W = g->world, me = c->rank;
  if (hipEventRecord(c->ready, stream) != hipSuccess) return 1;   // everything queued before the collective (the pack) precedes this
  std::unique_lock<std::mutex> lk(g->m);
  g->slot[me] = Slot{sendbuff, recvbuff, bytes, stream, c->ready, c->copied};
  if (!rendezvous(g, lk)) return 2;
  Slot peers[kMaxRanks];
  for (int p = 0; p < W; ++p) peers[p] = g->slot[p];
  lk.unlock();
  for (int p = 0; p < W; ++p) {
    char* dst = (char*)recvbuff + (size_t)p * bytes;
    if (p == me) {
      if ((const void*)dst != sendbuff && hipMemcpyAsync(dst, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
      continue;
    }
    if (hipStreamWaitEvent(stream, peers[p].ready, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(dst, peers[p].send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
  }
  if (hipEventRecord(c->copied, stream) != hipSuccess) return 1;
  lk.lock();
  if (!rendezvous(g, lk)) return 2;
  if (me == 0) { ++g->calls; std::lock_guard<std::mutex> l2(g_mu); ++g_allgathers; }
  lk.unlock();
  for (int p = 0; p < W; ++p)
    if (p != me && hipStreamWaitEvent(stream, peers[p].copied, 0) != hipSuccess) return 1;
  return 0;
}

// (test hook) collectives completed so far, over all groups
uint64_t kprn_loopback_allgathers() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_allgathers;
}

}  // extern "C"
