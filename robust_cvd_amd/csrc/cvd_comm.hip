// cvd_comm.hip -- the exchange steps of the pair-sharded multi-GPU mode: sum all-reduce, sum
// reduce-scatter and all-gather over the ranks of one solve.
//
// Production backend: RCCL over xGMI, one process per GPU, one communicator per handle (cvd_comm_init), collectives
// enqueued on the solver's stream.
//
// Test backend ("local group", cvd_comm_init_local_group): the ranks are handles of ONE process on ONE device, each
// driven by its own host thread.  RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the GPU boxes this
// repository is developed on have one GPU -- the local group is how the multi-rank code paths (owner chunks, padding,
// reduce-scatter offsets, the fused product exchange) execute with world > 1 there.  It is host-synchronous (every
// collective drains the caller's stream and meets the other ranks at a host barrier) and is never used by bench.py or by
// the product path of a multi-GPU run.  Reductions add the ranks' contributions in rank order on every rank, so all
// ranks hold bit-identical results -- the property of RCCL's collectives the solver's host decisions rely on.
#include "cvd_host.h"

namespace cvd {

struct LocalGroup {
  int world = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  bool broken = false;  // a rank failed or timed out: every later barrier throws
  std::vector<const unsigned char*> stage;  // per rank: device staging copy of its contribution
  int members = 0;
};

static std::mutex g_groupsMutex;
static std::map<unsigned long long, std::shared_ptr<LocalGroup>> g_groups;

std::shared_ptr<LocalGroup> joinLocalGroup(unsigned long long key, int world) {
  std::lock_guard<std::mutex> lock(g_groupsMutex);
  auto& g = g_groups[key];
  if (!g) {
    g = std::make_shared<LocalGroup>();
    g->world = world;
    g->stage.assign(world, nullptr);
  }
  if (g->world != world) throw std::runtime_error("local group: world size differs from the group's");
  if (g->members >= world) throw std::runtime_error("local group: more members than ranks");
  ++g->members;
  std::shared_ptr<LocalGroup> out = g;
  if (g->members == world) g_groups.erase(key);  // complete: the key may be reused by a later group
  return out;
}

static void groupBarrier(LocalGroup& g) {
  std::unique_lock<std::mutex> lk(g.m);
  if (g.broken) throw std::runtime_error("local group: another rank failed");
  const unsigned long long gen = g.generation;
  if (++g.arrived == g.world) {
    g.arrived = 0;
    ++g.generation;
    g.cv.notify_all();
    return;
  }
  if (!g.cv.wait_for(lk, std::chrono::seconds(120), [&]() { return g.generation != gen || g.broken; })) {
    g.broken = true;
    g.cv.notify_all();
    throw std::runtime_error("local group: barrier timed out (a rank left the collective sequence)");
  }
  // (a barrier that completed stays completed even if a member has left since: it may destroy its handle right after)
  if (g.generation == gen) throw std::runtime_error("local group: another rank failed");
}

void leaveLocalGroup(LocalGroup& g) {
  std::lock_guard<std::mutex> lk(g.m);
  g.broken = true;  // (a destroyed member can never arrive again)
  g.cv.notify_all();
}

static size_t typeSize(CommType t) { return t == CT_F32 || t == CT_I32 ? 4 : 8; }
static ncclDataType_t ncclType(CommType t) {
  switch (t) {
    case CT_F64: return ncclDouble;
    case CT_F32: return ncclFloat;
    case CT_I32: return ncclInt;
    default: return ncclUint64;
  }
}

// out[i] = sum over ranks r (ascending) of stage_r[offset + i]
template <typename T>
__global__ void k_local_sum(int world, const unsigned char* const* __restrict__ stage, size_t offset, size_t count,
                            T* __restrict__ out) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= count) return;
  T acc = reinterpret_cast<const T*>(stage[0])[offset + i];
  for (int r = 1; r < world; ++r) acc += reinterpret_cast<const T*>(stage[r])[offset + i];
  out[i] = acc;
}

// The local group's collectives.  kind 0: all-reduce (send = recv = buf, count elements); 1: reduce-scatter (send holds
// world x count elements, recv receives chunk `rank`); 2: all-gather (send holds count elements, recv world x count).
static void localCollective(cvd_handle* h, int kind, const void* send, void* recv, size_t count, CommType t, hipStream_t s) {
  LocalGroup& g = *h->localGroup;
  const size_t es = typeSize(t);
  const size_t sendCount = kind == 1 ? count * g.world : count;
  h->dCommStage.ensure(std::max<size_t>(sendCount * es, 8));
  HIP_CHECK(hipMemcpyAsync(h->dCommStage.p, send, sendCount * es, hipMemcpyDeviceToDevice, s));
  HIP_CHECK(hipStreamSynchronize(s));
  {
    std::lock_guard<std::mutex> lk(g.m);
    g.stage[h->rank] = h->dCommStage.p;
  }
  groupBarrier(g);  // every rank's contribution is staged and published
  std::vector<const unsigned char*> st;
  {
    std::lock_guard<std::mutex> lk(g.m);
    st = g.stage;
  }
  if (kind == 2) {
    for (int r = 0; r < g.world; ++r)
      HIP_CHECK(hipMemcpyAsync(static_cast<unsigned char*>(recv) + static_cast<size_t>(r) * count * es, st[r], count * es,
                               hipMemcpyDeviceToDevice, s));
  } else {
    h->dCommPtrs.upload(st.data(), st.size(), s);
    const size_t offset = kind == 1 ? static_cast<size_t>(h->rank) * count : 0;
    const unsigned grid = static_cast<unsigned>((count + 255) / 256);
    if (count > 0) {
      switch (t) {
        case CT_F64: hipLaunchKernelGGL(k_local_sum<double>, dim3(grid), dim3(256), 0, s, g.world, h->dCommPtrs.p, offset, count, static_cast<double*>(recv)); break;
        case CT_F32: hipLaunchKernelGGL(k_local_sum<float>, dim3(grid), dim3(256), 0, s, g.world, h->dCommPtrs.p, offset, count, static_cast<float*>(recv)); break;
        case CT_I32: hipLaunchKernelGGL(k_local_sum<int>, dim3(grid), dim3(256), 0, s, g.world, h->dCommPtrs.p, offset, count, static_cast<int*>(recv)); break;
        default: hipLaunchKernelGGL(k_local_sum<unsigned long long>, dim3(grid), dim3(256), 0, s, g.world, h->dCommPtrs.p, offset, count, static_cast<unsigned long long*>(recv)); break;
      }
      HIP_CHECK(hipGetLastError());
    }
  }
  HIP_CHECK(hipStreamSynchronize(s));
  groupBarrier(g);  // nobody restages before every rank has read
}

// Phantom rank (cvd_comm_init_phantom): the in-place forms leave the own contribution where it is; the out-of-place ones copy it.
static void phantomCopy(const void* src, void* dst, size_t bytes, hipStream_t s) {
  if (src != dst && bytes) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
}
void commAllReduce(cvd_handle* h, void* buf, size_t count, CommType t, hipStream_t s) {
  if (h->phantom) return;
  if (h->localGroup) { localCollective(h, 0, buf, buf, count, t, s); return; }
  NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclType(t), ncclSum, h->comm, s));
}
void commReduceScatter(cvd_handle* h, const void* send, void* recv, size_t recvCount, CommType t, hipStream_t s) {
  if (h->phantom) {
    phantomCopy(static_cast<const unsigned char*>(send) + static_cast<size_t>(h->rank) * recvCount * typeSize(t), recv, recvCount * typeSize(t), s);
    return;
  }
  if (h->localGroup) { localCollective(h, 1, send, recv, recvCount, t, s); return; }
  NCCL_CHECK(ncclReduceScatter(send, recv, recvCount, ncclType(t), ncclSum, h->comm, s));
}
void commAllGather(cvd_handle* h, const void* send, void* recv, size_t sendCount, CommType t, hipStream_t s) {
  if (h->phantom) {
    phantomCopy(send, static_cast<unsigned char*>(recv) + static_cast<size_t>(h->rank) * sendCount * typeSize(t), sendCount * typeSize(t), s);
    return;
  }
  if (h->localGroup) { localCollective(h, 2, send, recv, sendCount, t, s); return; }
  NCCL_CHECK(ncclAllGather(send, recv, sendCount, ncclType(t), h->comm, s));
}
void commGroupStart(cvd_handle* h) {
  if (!h->localGroup && !h->phantom) NCCL_CHECK(ncclGroupStart());
}
void commGroupEnd(cvd_handle* h) {
  if (!h->localGroup && !h->phantom) NCCL_CHECK(ncclGroupEnd());
}

}  // namespace cvd
