"""ctypes face of the RCCL entry points of libgator_hip (include/gator_hip.h: gator_comm_*, gator_allgather_verts): the path's one
collective without torch.distributed.  `NativeComm.create` needs SOME host-side channel to hand rank 0's 128-byte id to the other
ranks; `exchange` is a callable (rank-0 bytes or None) -> bytes on every rank (e.g. an MPI bcast, a file, or -- in the tests --
torch.distributed's object broadcast)."""
import ctypes

import torch

from . import _lib

ID_BYTES = 128


class NativeComm:
    def __init__(self, handle, rank, world):
        self._h, self.rank, self.world = handle, rank, world

    @classmethod
    def create(cls, rank, world, exchange):
        lib = _lib.load()
        buf = (ctypes.c_uint8 * ID_BYTES)()
        if rank == 0:
            _lib.check(lib.gator_comm_unique_id(buf), 'gator_comm_unique_id')
        uid = exchange(bytes(buf) if rank == 0 else None)
        buf = (ctypes.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        h = ctypes.c_void_p()
        _lib.check(lib.gator_comm_create(buf, rank, world, ctypes.byref(h)), 'gator_comm_create')
        return cls(h, rank, world)

    def allgather(self, verts, pose3d=None):
        """verts [B,6890,3] (+ pose3d [B,J,3]) of this rank -> the rank-major concatenation over all ranks, on the current stream."""
        verts = verts.contiguous()
        B = verts.shape[0]
        out_v = torch.empty((self.world * B, 6890, 3), device=verts.device, dtype=torch.float32)
        out_p, J = None, 0
        if pose3d is not None:
            pose3d = pose3d.contiguous()
            J = pose3d.shape[1]
            out_p = torch.empty((self.world * B, J, 3), device=verts.device, dtype=torch.float32)
        st = ctypes.c_void_p(torch.cuda.current_stream(verts.device).cuda_stream)
        _lib.check(_lib.load().gator_allgather_verts(self._h, verts.data_ptr(), pose3d.data_ptr() if pose3d is not None else None, B, J,
                                                     out_v.data_ptr(), out_p.data_ptr() if out_p is not None else None, st),
                   'gator_allgather_verts')
        return (out_v, out_p) if pose3d is not None else out_v

    def close(self):
        if self._h is not None:
            _lib.load().gator_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
