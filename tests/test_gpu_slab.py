"""GPU (-m gpu): the Z-slab path on real hardware.  The box has ONE GPU, so the two ranks share it and the
collectives are staged through gloo (Comm(stage_via_host=True)); kernels, views, halos and the global
ordering are exactly those of the multi-GPU run, only the transport differs (RCCL there)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sift3d_amd import abi, synth
from tests import parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes as C, json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import sift3d_amd
from sift3d_amd import synth
from sift3d_amd.slab import Comm, SlabSift3D
out, nx, ny, nz, nblobs, seed = sys.argv[2], *(int(v) for v in sys.argv[3:8])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
comm = Comm(dist, stage_via_host=True)
sl = SlabSift3D(sift3d_amd.cdll(), "cuda:0", comm, nx, ny, nz)
z0, z1 = sl.part[0]
vol = torch.from_numpy(synth.blobs(nx, ny, nz, nblobs, seed, z0=z0, z1=z1)).cuda()
k = sl.detect(vol)
desc = sl.describe()
xyzos, R, d = sl.gather_keypoints(desc)
if dist.get_rank() == 0:
    np.savez(out, xyzos=xyzos, R=R, desc=d, o_shard=sl.o_shard, local_k=k)
dist.barrier()
dist.destroy_process_group()
'''


def test_slab_two_ranks_share_one_gpu(hip, tmp_path):
    nx, ny, nz, nblobs, seed = 96, 80, 192, 1400, 5          # slab 96 slices: octaves 0 and 1 are sharded (H = 40)
    out = str(tmp_path / "slab.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 2000), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, out, str(nx), str(ny), str(nz), str(nblobs), str(seed)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    got = np.load(out)
    assert int(got["o_shard"]) == 1
    vol = synth.blobs(nx, ny, nz, nblobs, seed)
    s, im, kp = parity.run_detect(hip, vol, (1, 1, 1))
    xyzos, sd, R = hip.keypoints_to_numpy(kp)
    d = abi.SIFT3D_Descriptor_store()
    hip.sift.init_SIFT3D_Descriptor_store(C.byref(d))
    assert hip.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
    bins, _ = hip.descriptors_to_numpy(d)
    assert len(xyzos) > 100 and 0 < int(got["local_k"]) < len(xyzos)
    assert np.array_equal(got["xyzos"], xyzos)              # same keypoints, same (o, s, z, y, x) order
    assert np.array_equal(got["R"], R)
    assert np.array_equal(got["desc"], bins)                # integer histogram: bitwise reproducible
