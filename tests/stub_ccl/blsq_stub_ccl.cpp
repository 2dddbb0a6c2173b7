// Test-only stand-in for librccl (tests/, never the product): the six entry points libblsq_hip.so
// binds (ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / AllGather / GetErrorString, plus
// GetVersion), implemented over TCP sockets on 127.0.0.1 with host staging, so that TWO PROCESSES
// SHARING ONE GPU can run the library's multi-rank code for real — RCCL itself refuses two ranks on one
// device.  Selected with BLSQ_RCCL_PATH (include/blsq.h).  Semantics kept from the real thing:
//   * rank order of the all-gather, sum / max of the all-reduce (the root adds in rank order and
//     broadcasts the result: every rank receives the same bits, as a ring/tree all-reduce delivers);
//   * stream order: the collective takes effect after everything enqueued before it on `stream`
//     and before everything enqueued after it (done by synchronising the stream: a stand-in, not fast);
//   * mismatched collectives (op, count, type) across ranks are an ERROR here (ncclInvalidUsage on every
//     rank) where real RCCL would hang — which is what the tests of the rank-agreement logic need.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct StubId {                      // lives inside the 128 bytes of ncclUniqueId
  uint32_t magic, port;
  uint64_t nonce;
};
constexpr uint32_t MAGIC = 0x42535143u;   // "BSQC"

struct Comm {
  int nranks = 1, rank = 0;
  std::vector<int> fds;              // root: fd of rank r at [r] (own slot -1); others: fds[0] = root
  long ncalls = 0;
};

std::mutex g_mu;
std::map<uint32_t, int> g_listen;    // port -> listening fd (made by ncclGetUniqueId in this process)

bool send_all(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) {
    ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n) {
    ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k; n -= (size_t)k;
  }
  return true;
}
void tune(int fd) {
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  timeval tv{120, 0};                // a peer that never shows up must not hang the test for ever
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclFloat64: case ncclInt64: case ncclUint64: return 8;
    case ncclFloat32: case ncclInt32: case ncclUint32: return 4;
    case ncclInt8: case ncclUint8: return 1;
    default: return 0;
  }
}

// BLSQ_STUB_HOST=1: the buffers are HOST memory and no HIP call is made (the CPU test of this file)
bool host_mode() { const char* e = getenv("BLSQ_STUB_HOST"); return e && e[0] == '1'; }
bool stage_in(void* host, const void* dev, size_t bytes, hipStream_t stream) {
  if (host_mode()) { memcpy(host, dev, bytes); return true; }
  return hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) == hipSuccess;
}
bool stage_out(void* dev, const void* host, size_t bytes) {
  if (host_mode()) { memcpy(dev, host, bytes); return true; }
  return hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess;
}

struct Header { uint32_t kind, dtype, op, pad; uint64_t count; };   // what a rank is about to do

// gather every rank's `bytes` at the root, let `combine` turn the stack (rank order) into the
// `out_bytes` every rank receives
template <class F>
ncclResult_t exchange(Comm* c, const Header& h, const void* send_host, size_t bytes, void* recv_host,
                      size_t out_bytes, F combine) {
  c->ncalls++;
  if (c->nranks == 1) {
    std::vector<char> stack((const char*)send_host, (const char*)send_host + bytes);
    combine(stack.data(), (char*)recv_host);
    return ncclSuccess;
  }
  if (c->rank != 0) {
    const int fd = c->fds[0];
    uint32_t verdict = 0;
    if (!send_all(fd, &h, sizeof(h)) || !send_all(fd, send_host, bytes)) return ncclSystemError;
    if (!recv_all(fd, &verdict, sizeof(verdict))) return ncclSystemError;
    if (verdict != 0) return ncclInvalidUsage;
    if (!recv_all(fd, recv_host, out_bytes)) return ncclSystemError;
    return ncclSuccess;
  }
  std::vector<char> stack(bytes * (size_t)c->nranks);
  memcpy(stack.data(), send_host, bytes);
  bool ok = true, io = true;
  for (int r = 1; r < c->nranks; ++r) {
    Header hr{};
    if (!recv_all(c->fds[r], &hr, sizeof(hr))) { io = false; break; }
    const bool same = hr.kind == h.kind && hr.dtype == h.dtype && hr.op == h.op && hr.count == h.count;
    if (!same) {                      // drain what the rank sends for ITS collective, then refuse everybody
      ok = false;
      const size_t theirs = (size_t)hr.count * type_size((ncclDataType_t)hr.dtype);
      std::vector<char> sink(theirs);
      if (!recv_all(c->fds[r], sink.data(), theirs)) { io = false; break; }
    } else if (!recv_all(c->fds[r], stack.data() + bytes * (size_t)r, bytes)) { io = false; break; }
  }
  if (!io) return ncclSystemError;
  const uint32_t verdict = ok ? 0u : 1u;
  for (int r = 1; r < c->nranks; ++r)
    if (!send_all(c->fds[r], &verdict, sizeof(verdict))) return ncclSystemError;
  if (!ok) return ncclInvalidUsage;
  combine(stack.data(), (char*)recv_host);
  for (int r = 1; r < c->nranks; ++r)
    if (!send_all(c->fds[r], recv_host, out_bytes)) return ncclSystemError;
  return ncclSuccess;
}

template <class T>
void reduce_typed(const char* stack, char* out, size_t count, int nranks, ncclRedOp_t op) {
  const T* s = reinterpret_cast<const T*>(stack);
  T* o = reinterpret_cast<T*>(out);
  for (size_t i = 0; i < count; ++i) {
    T v = s[i];
    for (int r = 1; r < nranks; ++r) {
      const T w = s[(size_t)r * count + i];
      switch (op) {
        case ncclSum: v = v + w; break;
        case ncclProd: v = v * w; break;
        case ncclMax: v = w > v ? w : v; break;
        case ncclMin: v = w < v ? w : v; break;
        default: break;
      }
    }
    o[i] = v;
  }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = 1;                       // (no real RCCL has this code: the stand-in is recognisable)
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error (stub ccl)";
    case ncclSystemError: return "socket error or time-out between the ranks (stub ccl)";
    case ncclInvalidUsage: return "the ranks entered different collectives (stub ccl)";
    case ncclInvalidArgument: return "invalid argument (stub ccl)";
    case ncclUnhandledCudaError: return "HIP error while staging (stub ccl)";
    default: return "error (stub ccl)";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* out) {
  if (!out) return ncclInvalidArgument;
  int fd = ::socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) return ncclSystemError;
  int one = 1;
  setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in a{};
  a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
  socklen_t al = sizeof(a);
  if (::bind(fd, (sockaddr*)&a, sizeof(a)) != 0 || ::listen(fd, 64) != 0 ||
      ::getsockname(fd, (sockaddr*)&a, &al) != 0) { ::close(fd); return ncclSystemError; }
  StubId id{MAGIC, (uint32_t)ntohs(a.sin_port),
            (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32)};
  memset(out, 0, sizeof(*out));
  memcpy(out, &id, sizeof(id));
  std::lock_guard<std::mutex> lk(g_mu);
  g_listen[id.port] = fd;
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId uid, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  StubId id{};
  memcpy(&id, &uid, sizeof(id));
  if (id.magic != MAGIC) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->nranks = nranks; c->rank = rank;
  if (rank == 0) {
    int lfd = -1;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      auto it = g_listen.find(id.port);
      if (it != g_listen.end()) { lfd = it->second; g_listen.erase(it); }
    }
    if (lfd < 0) { delete c; return ncclInvalidUsage; }       // rank 0 must be the process that made the id
    c->fds.assign(nranks, -1);
    timeval tv{120, 0};
    setsockopt(lfd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    for (int got = 1; got < nranks;) {
      int fd = ::accept(lfd, nullptr, nullptr);
      if (fd < 0) { ::close(lfd); delete c; return ncclSystemError; }
      tune(fd);
      uint64_t hello[2] = {0, 0};                             // nonce, rank
      if (!recv_all(fd, hello, sizeof(hello)) || hello[0] != id.nonce || hello[1] == 0 ||
          hello[1] >= (uint64_t)nranks || c->fds[hello[1]] != -1) { ::close(fd); continue; }
      c->fds[hello[1]] = fd;
      ++got;
    }
    ::close(lfd);
    const uint32_t go = 1;                                    // everybody is here
    for (int r = 1; r < nranks; ++r) send_all(c->fds[r], &go, sizeof(go));
  } else {
    int fd = -1;
    for (int attempt = 0; attempt < 2400; ++attempt) {        // the root may not listen yet: <= 120 s
      fd = ::socket(AF_INET, SOCK_STREAM, 0);
      sockaddr_in a{};
      a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = htons((uint16_t)id.port);
      if (::connect(fd, (sockaddr*)&a, sizeof(a)) == 0) break;
      ::close(fd); fd = -1;
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    if (fd < 0) { delete c; return ncclSystemError; }
    tune(fd);
    uint64_t hello[2] = {id.nonce, (uint64_t)rank};
    uint32_t go = 0;
    if (!send_all(fd, hello, sizeof(hello)) || !recv_all(fd, &go, sizeof(go)) || go != 1) {
      ::close(fd); delete c; return ncclSystemError;
    }
    c->fds.assign(1, fd);
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclInvalidArgument;
  for (int fd : c->fds) if (fd >= 0) ::close(fd);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                           ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t ts = type_size(datatype);
  if (!c || !sendbuff || !recvbuff || ts == 0) return ncclInvalidArgument;
  if (datatype != ncclFloat64 && datatype != ncclInt32 && datatype != ncclInt64 && datatype != ncclFloat32)
    return ncclInvalidArgument;
  const size_t bytes = count * ts;
  std::vector<char> hs(bytes), hr(bytes);
  if (!stage_in(hs.data(), sendbuff, bytes, stream)) return ncclUnhandledCudaError;
  const Header h{1u, (uint32_t)datatype, (uint32_t)op, 0u, (uint64_t)count};
  const int nr = c->nranks;
  ncclResult_t r = exchange(c, h, hs.data(), bytes, hr.data(), bytes, [&](const char* stack, char* out) {
    switch (datatype) {
      case ncclFloat64: reduce_typed<double>(stack, out, count, nr, op); break;
      case ncclFloat32: reduce_typed<float>(stack, out, count, nr, op); break;
      case ncclInt32: reduce_typed<int32_t>(stack, out, count, nr, op); break;
      default: reduce_typed<int64_t>(stack, out, count, nr, op); break;
    }
  });
  if (r != ncclSuccess) return r;
  if (!stage_out(recvbuff, hr.data(), bytes)) return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t ts = type_size(datatype);
  if (!c || !sendbuff || !recvbuff || ts == 0) return ncclInvalidArgument;
  const size_t bytes = sendcount * ts, all = bytes * (size_t)c->nranks;
  std::vector<char> hs(bytes), hr(all);
  if (!stage_in(hs.data(), sendbuff, bytes, stream)) return ncclUnhandledCudaError;
  const Header h{2u, (uint32_t)datatype, 0u, 0u, (uint64_t)sendcount};
  ncclResult_t r = exchange(c, h, hs.data(), bytes, hr.data(), all,
                            [&](const char* stack, char* out) { memcpy(out, stack, all); });   // slot r = rank r
  if (r != ncclSuccess) return r;
  if (!stage_out(recvbuff, hr.data(), all)) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // extern "C"
