// Collectives of the corpus-sharded VCMR pass behind the C ABI: RCCL over xGMI, one process per GPU.
//
// The reference has no distributed path; what must be preserved is the driver's
// semantics: moments are taken only from the GLOBAL top-k videos of each query (xml/inference.py:347-348,365-367).
// With the corpus sharded by video range every rank holds a local top-c list for every query; the exact global top-k
// of a query is the top-k of the union of those lists.  The merge is partitioned by QUERY OWNER (rank r owns the
// contiguous query slice r), so a rank receives and merges only the lists it needs:
//
//   xml_rccl_topk_by_owner   grouped ncclSend / ncclRecv (rows of slice r go to rank r: an all-to-all with ragged
//                            counts, no packing pass)  ->  one un-permute kernel (source-rank-major -> query-major
//                            candidate rows)  ->  the K8 top-k kernel with the payload ids (same tie rule as the
//                            single-GPU pass: score desc, video id asc).  Three launches per call.
//   xml_rccl_allgather       the modular query vectors of every owner's slice (one ncclAllGather).
//
// RCCL is resolved at run time (dlopen of the copy already mapped by PyTorch when there is one -- two RCCL copies in
// one process would each want the GPU's IPC handles), so libxmlhip.so itself loads on a machine without RCCL.
#include <dlfcn.h>
#include <limits.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>

#include "common.h"

namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // the copy PyTorch already mapped, if any
      if (h) break;
    }
    if (!h)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    if (!h) return;
    bool all = true;
    auto sym = [&](const char* n) { void* p = dlsym(h, n); all = all && p != nullptr; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.ok = all;
  });
  return api;
}

// recv buffers hold, for every source rank p, the rows of MY query slice: [p][row][c].  The merge wants one candidate
// row per query: [row][p * c + j].  Scores and ids in one launch.
__global__ __launch_bounds__(256) void unpermute_by_owner_kernel(const float* __restrict__ rs, const int32_t* __restrict__ ri,
                                                                 float* __restrict__ cs, int32_t* __restrict__ ci, int world,
                                                                 int n_own, int c) {
  const int64_t total = (int64_t)world * n_own * c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % c);
    const int64_t t = i / c;
    const int p = (int)(t % world);
    const int64_t row = t / world;
    const int64_t src = ((int64_t)p * n_own + row) * c + j;
    cs[i] = rs[src];
    ci[i] = ri[src];
  }
}

inline int slice_rows(int nq, int per, int r) {
  const int64_t lo = (int64_t)r * per;
  if (lo >= nq) return 0;
  const int64_t hi = lo + per < nq ? lo + per : nq;
  return (int)(hi - lo);
}

// the owner's side behind the wire: receive layout [p][row][c] -> candidate rows [row][p * c + j] -> top-k
int merge_received(const float* rs, const int32_t* ri, float* cs, int32_t* ci, int world, int n_rows, int c, int k,
                   float alpha, float* out_val, int32_t* out_id, hipStream_t st) {
  const int64_t total = (int64_t)world * n_rows * c;
  const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(unpermute_by_owner_kernel, dim3(grid), dim3(256), 0, st, rs, ri, cs, ci, world, n_rows, c);
  XML_CHECK_LAUNCH();
  return xml_topk_rows(cs, (int64_t)world * c, ci, out_val, out_id, n_rows, world * c, k, alpha, nullptr, 0, (xml_stream_t)st);
}

}  // namespace

extern "C" size_t xml_merge_shard_topk_workspace_bytes(int world, int n_rows, int c) {
  if (world <= 0 || n_rows <= 0 || c <= 0) return 0;
  return 2 * align_up((size_t)world * n_rows * c * 4, 256);      // candidate scores / ids
}

extern "C" int xml_merge_shard_topk(const float* recv_score, const int32_t* recv_id, int world, int n_rows, int c, int k,
                                    float alpha, float* out_val, int32_t* out_id, void* ws, size_t ws_bytes,
                                    xml_stream_t stream) {
  XML_ENTER();
  if (!recv_score || !recv_id || !out_val || !out_id || !ws) return XML_ERR_BAD_ARG;
  if (world <= 0 || n_rows <= 0 || c <= 0 || k <= 0) return XML_ERR_BAD_ARG;
  if (k > 256 || k > world * c) return XML_ERR_UNSUPPORTED;
  if (ws_bytes < xml_merge_shard_topk_workspace_bytes(world, n_rows, c)) return XML_ERR_WORKSPACE;
  const size_t seg = align_up((size_t)world * n_rows * c * 4, 256);
  return merge_received(recv_score, recv_id, (float*)ws, (int32_t*)((char*)ws + seg), world, n_rows, c, k, alpha, out_val,
                        out_id, (hipStream_t)stream);
}

extern "C" int xml_rccl_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int xml_rccl_unique_id(void* id128) {
  if (!id128) return XML_ERR_BAD_ARG;
  if (!rccl().ok) return XML_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  return rccl().GetUniqueId((ncclUniqueId*)id128) == ncclSuccess ? XML_OK : XML_ERR_LAUNCH;
}

extern "C" int xml_rccl_comm_init(xml_comm_t* comm, int nranks, int rank, const void* id128) {
  if (!comm || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return XML_ERR_BAD_ARG;
  if (!rccl().ok) return XML_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy((void*)&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  if (rccl().CommInitRank(&c, nranks, id, rank) != ncclSuccess) return XML_ERR_LAUNCH;
  *comm = (xml_comm_t)c;
  return XML_OK;
}

extern "C" int xml_rccl_comm_destroy(xml_comm_t comm) {
  if (!comm) return XML_ERR_BAD_ARG;
  if (!rccl().ok) return XML_ERR_UNSUPPORTED;
  return rccl().CommDestroy((ncclComm_t)comm) == ncclSuccess ? XML_OK : XML_ERR_LAUNCH;
}

extern "C" int xml_rccl_allgather(xml_comm_t comm, const void* send, void* recv, int64_t bytes_per_rank,
                                  xml_stream_t stream) {
  if (!comm || !send || !recv || bytes_per_rank <= 0) return XML_ERR_BAD_ARG;
  if (!rccl().ok) return XML_ERR_UNSUPPORTED;
  return rccl().AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess
             ? XML_OK : XML_ERR_LAUNCH;
}

extern "C" int xml_rccl_allreduce_avg_f32(xml_comm_t comm, float* buf, int64_t n, xml_stream_t stream) {
  if (!comm || !buf || n <= 0) return XML_ERR_BAD_ARG;
  if (!rccl().ok) return XML_ERR_UNSUPPORTED;
  return rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclAvg, (ncclComm_t)comm, (hipStream_t)stream) == ncclSuccess
             ? XML_OK : XML_ERR_LAUNCH;
}

// The plain scheme of BASELINE.json's north_star: all-gather the per-shard top-c lists, every rank merges ALL queries.
// (The engine's default is the by-owner exchange below -- 1/P of the receive volume and of the merge work per rank; this
// entry serves the paths that need the global top-k everywhere: the video-owner rerank, API parity with vcmr_search.)
extern "C" size_t xml_rccl_allgather_topk_workspace_bytes(int world, int nq, int c) {
  if (world <= 0 || nq <= 0 || c <= 0) return 0;
  return 4 * align_up((size_t)world * nq * c * 4, 256);
}

extern "C" int xml_rccl_allgather_topk(xml_comm_t comm, int world, const float* loc_score, const int32_t* loc_id, int nq,
                                       int c, int k, float alpha, float* out_val, int32_t* out_id, void* ws,
                                       size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!comm || !loc_score || !loc_id || !out_val || !out_id || !ws) return XML_ERR_BAD_ARG;
  if (world <= 0 || nq <= 0 || c <= 0 || k <= 0) return XML_ERR_BAD_ARG;
  if (k > 256 || k > world * c) return XML_ERR_UNSUPPORTED;
  if (!rccl().ok) return XML_ERR_UNSUPPORTED;
  if (ws_bytes < xml_rccl_allgather_topk_workspace_bytes(world, nq, c)) return XML_ERR_WORKSPACE;
  const size_t seg = align_up((size_t)world * nq * c * 4, 256);
  float* rs = (float*)ws;
  int32_t* ri = (int32_t*)((char*)ws + seg);
  float* cs = (float*)((char*)ws + 2 * seg);
  int32_t* ci = (int32_t*)((char*)ws + 3 * seg);
  hipStream_t st = (hipStream_t)stream;
  ncclComm_t nc = (ncclComm_t)comm;
  RcclApi& r = rccl();
  bool ok = r.GroupStart() == ncclSuccess;
  ok = ok && r.AllGather(loc_score, rs, (size_t)nq * c, ncclFloat32, nc, st) == ncclSuccess;
  ok = ok && r.AllGather(loc_id, ri, (size_t)nq * c, ncclInt32, nc, st) == ncclSuccess;
  ok = (r.GroupEnd() == ncclSuccess) && ok;
  if (!ok) return XML_ERR_LAUNCH;
  return merge_received(rs, ri, cs, ci, world, nq, c, k, alpha, out_val, out_id, st);
}

extern "C" size_t xml_rccl_topk_by_owner_workspace_bytes(int world, int per, int c) {
  if (world <= 0 || per <= 0 || c <= 0) return 0;
  return 4 * align_up((size_t)world * per * c * 4, 256);      // recv scores / ids, candidate scores / ids
}

extern "C" int xml_rccl_topk_by_owner(xml_comm_t comm, int world, int rank, const float* loc_score,
                                      const int32_t* loc_id, int nq, int c, int k, float alpha, float* own_val,
                                      int32_t* own_id, void* ws, size_t ws_bytes, xml_stream_t stream) {
  XML_ENTER();
  if (!comm || !loc_score || !loc_id || !own_val || !own_id || !ws) return XML_ERR_BAD_ARG;
  if (world <= 0 || rank < 0 || rank >= world || nq <= 0 || c <= 0 || k <= 0) return XML_ERR_BAD_ARG;
  if (k > 256 || k > world * c) return XML_ERR_UNSUPPORTED;
  if (!rccl().ok) return XML_ERR_UNSUPPORTED;
  const int per = (nq + world - 1) / world;
  if (ws_bytes < xml_rccl_topk_by_owner_workspace_bytes(world, per, c)) return XML_ERR_WORKSPACE;
  const int n_own = slice_rows(nq, per, rank);
  const size_t seg = align_up((size_t)world * per * c * 4, 256);
  float* rs = (float*)ws;
  int32_t* ri = (int32_t*)((char*)ws + seg);
  float* cs = (float*)((char*)ws + 2 * seg);
  int32_t* ci = (int32_t*)((char*)ws + 3 * seg);
  hipStream_t st = (hipStream_t)stream;
  ncclComm_t nc = (ncclComm_t)comm;
  RcclApi& r = rccl();
  bool ok = r.GroupStart() == ncclSuccess;
  for (int p = 0; p < world && ok; ++p) {
    const int rows_p = slice_rows(nq, per, p);        // what rank p owns: I send those rows of my local lists
    if (rows_p > 0) {
      const int64_t off = (int64_t)p * per * c;
      ok = ok && r.Send(loc_score + off, (size_t)rows_p * c, ncclFloat32, p, nc, st) == ncclSuccess;
      ok = ok && r.Send(loc_id + off, (size_t)rows_p * c, ncclInt32, p, nc, st) == ncclSuccess;
    }
    if (n_own > 0) {                                  // and rank p sends me its lists for my rows
      const int64_t off = (int64_t)p * n_own * c;
      ok = ok && r.Recv(rs + off, (size_t)n_own * c, ncclFloat32, p, nc, st) == ncclSuccess;
      ok = ok && r.Recv(ri + off, (size_t)n_own * c, ncclInt32, p, nc, st) == ncclSuccess;
    }
  }
  ok = (r.GroupEnd() == ncclSuccess) && ok;
  if (!ok) return XML_ERR_LAUNCH;
  if (n_own == 0) return XML_OK;
  return merge_received(rs, ri, cs, ci, world, n_own, c, k, alpha, own_val, own_id, st);
}
