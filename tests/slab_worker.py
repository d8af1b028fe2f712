"""Worker of tests/test_slab_gloo.py: one rank of a world_size-2 gloo job on the CPU.  The kernels are the
product's .hip sources executed by the SIMT emulator (tests/emu); device memory is host memory, so the
halo planes travel through gloo exactly where RCCL moves them on the GPU box."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift3d_amd import synth                 # noqa: E402
from sift3d_amd.slab import Comm, SlabSift3D  # noqa: E402


def main():
    out_path, nx, ny, nz, nblobs, seed = sys.argv[1], *(int(v) for v in sys.argv[2:7])
    params = json.loads(sys.argv[7])
    units = tuple(json.loads(sys.argv[8]))
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libsift3d_emu.so"))
    comm = Comm(dist)
    sl = SlabSift3D(L, "cpu", comm, nx, ny, nz, units=units, params=params)
    z0, z1 = sl.part[0]
    vol = synth.blobs(nx, ny, nz, nblobs, seed, z0=z0, z1=z1)
    k = sl.detect(torch.from_numpy(vol))
    desc = sl.describe()
    xyzos, R, d = sl.gather_keypoints(desc)
    if rank == 0:
        np.savez(out_path, xyzos=xyzos, R=R, desc=d, o_shard=sl.o_shard, H=sl.H, local_k=k,
                 bytes_exchanged=comm.bytes_exchanged, ncand=sl.num_candidates)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
