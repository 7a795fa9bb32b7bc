"""RCCL communicator for the C-ABI collectives (include/xmlhip.h: xml_rccl_*).

One process per GPU.  The communicator is created from a 128-byte ncclUniqueId made on rank 0 and handed to the other
ranks through the already-initialised torch.distributed group (its store is the side channel -- plumbing); every
collective of the sharded pass then runs inside libxmlhip.so on the stream the caller names.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check


class RcclComm(object):
    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise _lib.XmlHipError("RcclComm needs an initialised torch.distributed group to exchange the unique id")
        lib = _lib.load()
        if not lib.xml_rccl_available():
            raise _lib.XmlHipError("no RCCL library could be resolved (librccl.so.1)")
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            check(lib.xml_rccl_unique_id(uid), "xml_rccl_unique_id")
        box = [uid.raw if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._uid = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        check(lib.xml_rccl_comm_init(ctypes.byref(handle), self.world, self.rank, self._uid), "xml_rccl_comm_init")
        self.handle = handle
        self._lib = lib

    def close(self):
        if self.handle is not None and self.handle.value:
            self._lib.xml_rccl_comm_destroy(self.handle)
        self.handle = None

    # ---- collectives (device tensors, current stream) ---------------------------------------------------------------
    def allgather(self, send):
        """(n, ...) contiguous -> (world * n, ...): rank-major concatenation."""
        assert send.is_cuda and send.is_contiguous()
        out = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        check(self._lib.xml_rccl_allgather(self.handle, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                           send.numel() * send.element_size(),
                                           ctypes.c_void_p(torch.cuda.current_stream(send.device).cuda_stream)),
              "xml_rccl_allgather")
        return out

    def allreduce_avg_(self, buf):
        assert buf.is_cuda and buf.is_contiguous() and buf.dtype == torch.float32
        check(self._lib.xml_rccl_allreduce_avg_f32(self.handle, ctypes.c_void_p(buf.data_ptr()), buf.numel(),
                                                   ctypes.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)),
              "xml_rccl_allreduce_avg_f32")
        return buf

    def allgather_topk(self, loc_score, loc_id, k, alpha):
        """All ranks' local top-c lists of all queries -> the merged global top-k of ALL queries on every rank."""
        from . import ops
        assert loc_score.is_cuda and loc_score.dtype == torch.float32 and loc_id.dtype == torch.int32
        assert loc_score.is_contiguous() and loc_id.is_contiguous() and loc_score.shape == loc_id.shape
        nq, c = loc_score.shape
        out_val = torch.empty((nq, k), dtype=torch.float32, device=loc_score.device)
        out_id = torch.empty((nq, k), dtype=torch.int32, device=loc_score.device)
        ws = ops._workspace(self._lib.xml_rccl_allgather_topk_workspace_bytes(self.world, nq, c), loc_score.device)
        check(self._lib.xml_rccl_allgather_topk(self.handle, self.world, ctypes.c_void_p(loc_score.data_ptr()),
                                                ctypes.c_void_p(loc_id.data_ptr()), nq, c, k, float(alpha),
                                                ctypes.c_void_p(out_val.data_ptr()), ctypes.c_void_p(out_id.data_ptr()),
                                                ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                                ctypes.c_void_p(torch.cuda.current_stream(loc_score.device).cuda_stream)),
              "xml_rccl_allgather_topk")
        return out_val, out_id

    def topk_by_owner(self, loc_score, loc_id, k, alpha):
        """loc_score (nq, c) f32 / loc_id (nq, c) int32: local top-c of all queries -> merged global top-k of the query
        slice this rank owns: (rows_owned, k) f32 [exp(alpha s) if alpha], (rows_owned, k) int32."""
        from . import ops
        assert loc_score.is_cuda and loc_score.dtype == torch.float32 and loc_id.dtype == torch.int32
        assert loc_score.is_contiguous() and loc_id.is_contiguous() and loc_score.shape == loc_id.shape
        nq, c = loc_score.shape
        per = (nq + self.world - 1) // self.world
        lo = min(self.rank * per, nq)
        n_own = min(lo + per, nq) - lo
        own_val = torch.empty((max(n_own, 1), k), dtype=torch.float32, device=loc_score.device)[:n_own]
        own_id = torch.empty((max(n_own, 1), k), dtype=torch.int32, device=loc_score.device)[:n_own]
        ws = ops._workspace(self._lib.xml_rccl_topk_by_owner_workspace_bytes(self.world, per, c), loc_score.device)
        check(self._lib.xml_rccl_topk_by_owner(self.handle, self.world, self.rank, ctypes.c_void_p(loc_score.data_ptr()),
                                               ctypes.c_void_p(loc_id.data_ptr()), nq, c, k, float(alpha),
                                               ctypes.c_void_p(own_val.data_ptr()), ctypes.c_void_p(own_id.data_ptr()),
                                               ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                               ctypes.c_void_p(torch.cuda.current_stream(loc_score.device).cuda_stream)),
              "xml_rccl_topk_by_owner")
        return own_val, own_id
