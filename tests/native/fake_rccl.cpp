// tests/native/fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl whose "ranks" are host threads of ONE process that
// share ONE GPU.  RCCL refuses communicators with two ranks on one device, and the test box has one device, so the library's
// native exchange (lcpc_amd/csrc/shard.cpp: lcpc_commit_sharded_device, lcpc_prove_sharded_rccl) could never run with more than one
// rank.  Loaded through LCPC_RCCL_LIB, this file gives those code paths N > 1: the compact node layout (all-gather of node 0 +
// grouped broadcasts of the extra nodes), the node table of the finish step, column slices, async tails on several commitments of
// one communicator, the three all-gathers of the sharded prove.  It implements the SEMANTICS the library relies on and nothing of
// RCCL's transport:
//   * stream order: a collective starts when its stream reaches it and the stream continues when the rank's receive buffer is
//     complete;
//   * a rank's send buffer is read by its peers before that rank's stream goes on (they copy out of it);
//   * all ranks call the collectives of a communicator in the same order (a call blocks on the host until every rank has made it:
//     stricter than RCCL, which only enqueues -- a library that submitted in different orders on different ranks would deadlock here
//     as it would there).
// The prototypes are the ones shard.cpp binds (rccl.h, NCCL 2.x ABI).  Not part of the product; built by tests/test_gpu_fake_rccl.py.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Group {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0;
  uint64_t generation = 0;
  int joined = 0;
  // the collective in flight (all ranks are inside the same call between two barriers)
  std::vector<const void*> send;
  std::vector<hipEvent_t> ready, done;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t gen = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};
struct Comm {
  Group* g;
  int rank;
};
std::mutex g_mu;
std::map<uint64_t, Group*> g_groups;
uint64_t g_next_id = 1;

struct UniqueId {
  char internal[128];
};

int fail(hipError_t e) { return e == hipSuccess ? 0 : 1; }

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id->internal, 0, sizeof id->internal);
  const uint64_t v = g_next_id++;
  memcpy(id->internal, &v, 8);
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
  uint64_t key;
  memcpy(&key, id.internal, 8);
  Group* g;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_groups.find(key);
    if (it == g_groups.end()) {
      g = new Group();
      g->n = nranks;
      g->send.resize(nranks);
      g->ready.resize(nranks);
      g->done.resize(nranks);
      g_groups[key] = g;
    } else {
      g = it->second;
    }
    if (g->n != nranks || rank < 0 || rank >= nranks) return 4;       // ncclInvalidArgument
  }
  if (hipEventCreateWithFlags(&g->ready[rank], hipEventDisableTiming) != hipSuccess) return 1;
  if (hipEventCreateWithFlags(&g->done[rank], hipEventDisableTiming) != hipSuccess) return 1;
  Comm* c = new Comm{g, rank};
  g->barrier();                        // a collective, like the real one: returns when every rank has joined
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  delete static_cast<Comm*>(comm);     // (groups and their events live as long as the test process)
  return 0;
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "fake rccl error"; }
int ncclGroupStart() { return 0; }     // every operation runs when it is called; all ranks call in the same order
int ncclGroupEnd() { return 0; }

// every rank contributes `count` bytes (dtype is ncclUint8 in this library) and receives all contributions in rank order
int ncclAllGather(const void* send, void* recv, size_t count, int /*dtype*/, void* comm, hipStream_t st) {
  Comm* c = static_cast<Comm*>(comm);
  Group* g = c->g;
  const int r = c->rank, n = g->n;
  int rc = fail(hipEventRecord(g->ready[r], st));          // my contribution is complete at this point of my stream
  g->send[r] = send;
  g->barrier();                                            // everybody's (pointer, event) is published
  for (int p = 0; p < n && !rc; p++) {
    if (p != r) rc = fail(hipStreamWaitEvent(st, g->ready[p], 0));
    void* dst = static_cast<char*>(recv) + (size_t)p * count;
    if (!rc && dst != g->send[p]) rc = fail(hipMemcpyAsync(dst, g->send[p], count, hipMemcpyDeviceToDevice, st));
  }
  if (!rc) rc = fail(hipEventRecord(g->done[r], st));      // I have read every peer's buffer at this point of my stream
  g->barrier();
  for (int p = 0; p < n && !rc; p++)                      // my stream goes on (and may overwrite my send buffer) only after
    if (p != r) rc = fail(hipStreamWaitEvent(st, g->done[p], 0));      // every peer has read it
  g->barrier();                                            // the shared slots are free for the next collective
  return rc;
}

int ncclBroadcast(const void* send, void* recv, size_t count, int /*dtype*/, int root, void* comm, hipStream_t st) {
  Comm* c = static_cast<Comm*>(comm);
  Group* g = c->g;
  const int r = c->rank, n = g->n;
  int rc = 0;
  if (r == root) {
    rc = fail(hipEventRecord(g->ready[r], st));
    g->send[r] = send;
  }
  g->barrier();
  if (r != root) {
    rc = fail(hipStreamWaitEvent(st, g->ready[root], 0));
    if (!rc) rc = fail(hipMemcpyAsync(recv, g->send[root], count, hipMemcpyDeviceToDevice, st));
  } else if (recv != send) {
    rc = fail(hipMemcpyAsync(recv, send, count, hipMemcpyDeviceToDevice, st));
  }
  if (!rc) rc = fail(hipEventRecord(g->done[r], st));
  g->barrier();
  if (r == root)
    for (int p = 0; p < n && !rc; p++)
      if (p != r) rc = fail(hipStreamWaitEvent(st, g->done[p], 0));
  g->barrier();
  return rc;
}

}  // extern "C"
