// TEST INFRASTRUCTURE: a stand-in for librccl.so exporting the six entry points libnidreg.so resolves with dlsym
// (csrc/nidreg_rccl.hip RcclApi), so that the in-library point-sharded evaluation (nidreg_shard_comm_init -> nidreg_eval: histogram ->
// ncclAllReduce(int64) -> entropy -> gradient -> ncclAllReduce(f64 x 7)) can run with TWO ranks -- two processes -- on a box
// with ONE GPU, where RCCL itself refuses two ranks on one device.  The "collective" goes through POSIX shared memory: every rank
// drains its stream, copies its buffer to its slot, meets the others at a barrier, reduces all slots in rank order and copies the
// result back.  Loaded through NIDREG_RCCL_LIB (tests/test_rccl_inlib.py).  Not a performance path, not shipped.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 1 << 20;  // >= nidreg_hist_words(256) * 8 = 528 KB
struct Shared {
  std::atomic<int> arrived[2];  // two alternating barrier counters
  std::atomic<int> phase;
  unsigned char slot[kMaxRanks][kSlotBytes];
};
struct Comm {
  Shared* sh;
  int nranks, rank;
  int round;
  char name[64];
};
bool barrier(Comm* c) {
  const int which = c->round & 1;
  c->round++;
  const int target = c->nranks * ((c->round + 1) / 2);  // counter `which` is used every other round
  c->sh->arrived[which].fetch_add(1, std::memory_order_acq_rel);
  const auto t0 = std::chrono::steady_clock::now();
  while (c->sh->arrived[which].load(std::memory_order_acquire) < target) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
    std::this_thread::yield();
  }
  return true;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "/nidreg_rccl_stub_%d_%lld", int(getpid()), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;  // ncclInvalidArgument
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 2;  // ncclSystemError
  if (ftruncate(fd, sizeof(Shared)) != 0) return 2;  // (fresh objects are zero-filled: counters start at 0 whichever rank comes first)
  void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  Comm* c = new Comm();
  c->sh = static_cast<Shared*>(p);
  c->nranks = nranks;
  c->rank = rank;
  c->round = 0;
  std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
  *comm = c;
  if (!barrier(c)) return 2;  // like ncclCommInitRank: returns once every rank has joined
  return 0;
}
int ncclCommDestroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return 0;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return 0;
}
int ncclCommCount(const void* comm, int* count) {
  if (!comm || !count) return 4;
  *count = static_cast<const Comm*>(comm)->nranks;
  return 0;
}
// dtype: 4 = int64, 8 = float64; op: 0 = sum, 2 = max (the values libnidreg.so uses)
int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t bytes = count * 8;
  if (!c || bytes > kSlotBytes || (dtype != 4 && dtype != 8) || (op != 0 && op != 2)) return 4;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;  // ncclUnhandledCudaError
  if (hipMemcpy(c->sh->slot[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (!barrier(c)) return 2;
  static thread_local unsigned char* acc = new unsigned char[kSlotBytes];
  std::memcpy(acc, c->sh->slot[0], bytes);
  for (int r = 1; r < c->nranks; r++) {
    for (size_t i = 0; i < count; i++) {
      if (dtype == 4) {
        int64_t a, b;
        std::memcpy(&a, acc + 8 * i, 8);
        std::memcpy(&b, c->sh->slot[r] + 8 * i, 8);
        a = op == 0 ? a + b : (a > b ? a : b);
        std::memcpy(acc + 8 * i, &a, 8);
      } else {
        double a, b;
        std::memcpy(&a, acc + 8 * i, 8);
        std::memcpy(&b, c->sh->slot[r] + 8 * i, 8);
        a = op == 0 ? a + b : (a > b ? a : b);
        std::memcpy(acc + 8 * i, &a, 8);
      }
    }
  }
  if (!barrier(c)) return 2;  // every rank has read every slot before anybody overwrites its own
  if (hipMemcpy(recv, acc, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 2 ? "stub: system error / barrier timeout" : (r == 4 ? "stub: invalid argument" : "stub: HIP error")); }
}
