// tests/native/fake_rccl_shm.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl whose ranks are PROCESSES that may share one GPU
// (tests/native/fake_rccl.cpp is the same for threads of one process).  `python bench.py --gpus N` and lcpc_amd.distributed's
// comm_init run one process per rank; RCCL refuses two ranks on one device and the test box has one, so that flow -- the id drawn
// on rank 0 and carried to the others by torch.distributed, lcpc_comm_init on every rank, the timed loop of async-tail commits on
// two commitments, the self-check against an unsharded commit -- could not run with the library's own exchange.  Loaded through
// LCPC_RCCL_LIB it can.  Collectives are staged through POSIX shared memory and fully host-synchronous (a call returns when the
// rank's receive buffer is complete): slower and stricter than RCCL, same results.  Not part of the product.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {

constexpr size_t SLOT = (size_t)64 << 20;      // bytes per rank and collective (the headline exchanges 8.4 MB per rank)
constexpr int MAX_RANKS = 16;

struct Shared {
  std::atomic<int> count;
  std::atomic<int> generation;
  char pad[56];
  unsigned char data[1];                       // MAX_RANKS slots of SLOT bytes follow
};
struct Comm {
  Shared* sh;
  int rank, n;
  void barrier() {
    const int gen = sh->generation.load(std::memory_order_acquire);
    if (sh->count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      sh->count.store(0, std::memory_order_relaxed);
      sh->generation.fetch_add(1, std::memory_order_acq_rel);
    } else {
      while (sh->generation.load(std::memory_order_acquire) == gen) usleep(20);
    }
  }
  unsigned char* slot(int r) { return sh->data + (size_t)r * SLOT; }
};
struct UniqueId {
  char internal[128];
};

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId* id) {
  static std::atomic<int> seq{0};
  memset(id->internal, 0, sizeof id->internal);
  snprintf(id->internal, sizeof id->internal, "/lcpc_fake_rccl_%d_%d", (int)getpid(), seq.fetch_add(1));
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') return 4;
  const size_t bytes = sizeof(Shared) + (size_t)MAX_RANKS * SLOT;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);          // a fresh segment is zero-filled: a valid barrier state
  if (fd < 0) return 2;
  if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return 2; }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  Comm* c = new Comm{static_cast<Shared*>(p), rank, nranks};
  c->barrier();                                // every rank has mapped the segment
  if (rank == 0) shm_unlink(id.internal);      // the name is no longer needed (the mappings keep the memory)
  c->barrier();
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  munmap(c->sh, sizeof(Shared) + (size_t)MAX_RANKS * SLOT);
  delete c;
  return 0;
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "fake rccl (shm) error"; }
int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }

int ncclAllGather(const void* send, void* recv, size_t count, int /*dtype*/, void* comm, hipStream_t st) {
  Comm* c = static_cast<Comm*>(comm);
  if (count > SLOT) return 4;
  if (hipMemcpyAsync(c->slot(c->rank), send, count, hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  c->barrier();                                // every contribution is in shared memory
  for (int p = 0; p < c->n; p++)
    if (hipMemcpyAsync(static_cast<char*>(recv) + (size_t)p * count, c->slot(p), count, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  c->barrier();                                // nobody overwrites a slot another rank is still reading
  return 0;
}

int ncclBroadcast(const void* send, void* recv, size_t count, int /*dtype*/, int root, void* comm, hipStream_t st) {
  Comm* c = static_cast<Comm*>(comm);
  if (count > SLOT) return 4;
  if (c->rank == root) {
    if (hipMemcpyAsync(c->slot(root), send, count, hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
  }
  c->barrier();
  if (c->rank != root || recv != send) {
    if (hipMemcpyAsync(recv, c->slot(root), count, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
  }
  c->barrier();
  return 0;
}

}  // extern "C"
