"""Helper of test_reduce_comm_between_two_processes (run as `python gpu_two_rank_rccl.py RANK DIR`, not collected by
pytest): one of two processes that share this box's one GPU, join a two-rank RCCL communicator made with
ncclGetUniqueId / ncclCommInitRank through ctypes, render their halves of a frame schedule and call
mi355pt_reduce_comm.  Prints one line: "init_error <code> <text>" when RCCL refuses the communicator (it does not
accept two ranks on one device), else "ok" after rank 0 has saved the reduced accumulator to DIR/acc.npy."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def main():
    rank, d = int(sys.argv[1]), sys.argv[2]
    import pathtracer_amd as pa
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1")
    rccl.ncclGetErrorString.restype = C.c_char_p
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    uid = UniqueId()
    id_path = os.path.join(d, "id.bin")
    if rank == 0:
        rc = rccl.ncclGetUniqueId(C.byref(uid))
        assert rc == 0, rc
        with open(id_path + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.rename(id_path + ".tmp", id_path)
    else:
        for _ in range(600):
            if os.path.exists(id_path):
                break
            time.sleep(0.05)
        C.memmove(C.byref(uid), open(id_path, "rb").read(), 128)
    W, H, F = 128, 72, 10
    sc = pa.load_ron_scene(os.path.join(ROOT, "fixtures", "scene1.ron"))
    with pa.Renderer(sc, pa.Settings(1, 50), W, H, device=0) as r:      # HIP is up before the communicator is made
        first, count = pa.partition_frames(F, 2, rank)
        r.render(seeds=[pa.frame_seed(0x00C0FFEE, first + k) for k in range(count)])
        comm = C.c_void_p()
        rc = rccl.ncclCommInitRank(C.byref(comm), 2, uid, rank)
        if rc != 0:
            print("init_error", rc, rccl.ncclGetErrorString(rc).decode(), flush=True)
            return
        rc = pa.lib().mi355pt_reduce_comm(r._ctx, comm, 0, rank, F)
        if rc != 0:
            print("reduce_error", rc, pa.lib().mi355pt_last_error(r._ctx).decode(), flush=True)
        elif rank == 0:
            np.save(os.path.join(d, "acc.npy"), r.read_accum())
            assert r.samples() == F
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
        if rc == 0:
            print("ok", flush=True)


if __name__ == "__main__":
    main()
