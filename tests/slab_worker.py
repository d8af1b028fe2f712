"""Worker of tests/test_slab_gloo.py: one rank of a world_size-N gloo job on the CPU, driving the C Z-slab driver
(csrc/host/s3d_host_slab.c) through a callback transport over torch.distributed.  The kernels are the product's
.hip sources executed by the SIMT emulator (tests/emu); device memory is host memory, so the halo planes travel
through gloo exactly where RCCL moves them on the GPU box."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sift3d_amd import abi, synth            # noqa: E402
from sift3d_amd.slab import DistTransport, Slab  # noqa: E402


def main():
    out_path, nx, ny, nz, nblobs, seed = sys.argv[1], *(int(v) for v in sys.argv[2:7])
    params = json.loads(sys.argv[7])
    units = tuple(json.loads(sys.argv[8]))
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "libsift3d_emu.so"))
    tr = DistTransport(L, dist)
    sl = Slab(L, tr.struct, nx, ny, nz, units=units, params=params)
    inf = sl.info()
    vol = synth.blobs(nx, ny, nz, nblobs, seed, z0=inf.z0, z1=inf.z1)
    k = sl.detect(vol, on_device=False)
    sl.describe()
    kp_all, d_all = sl.gather()
    inf = sl.info()
    if rank == 0:
        xyzos, sd, R = abi.Sift3dLib.keypoints_to_numpy(kp_all)
        bins, xyzs = abi.Sift3dLib.descriptors_to_numpy(d_all)
        np.savez(out_path, xyzos=xyzos, sd=sd, R=R, desc=bins, dxyzs=xyzs, o_shard=inf.o_shard, H=inf.halo, local_k=k,
                 bytes_exchanged=tr.bytes_sent, halo_bytes=inf.halo_bytes, ncand=inf.num_candidates)
    dist.barrier()
    sl.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
