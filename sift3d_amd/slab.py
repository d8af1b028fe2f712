"""ctypes bindings of the multi-GPU Z-slab API (include/sift3d_amd_slab.h).  Plumbing only.

The driver itself -- partitioning, halo exchange schedule, kernel sequence -- is host C
(``csrc/host/s3d_host_slab.c``); its production transport is RCCL (``csrc/s3d_rccl.hip``).  What lives here:

* ``Slab``                 one rank of a job, over any transport;
* ``loopback_transports``  the library's in-process transport (ranks = host threads, possibly sharing one GPU);
* ``rccl_transport``       RCCL for one-process-per-GPU jobs: rank 0's 128-byte id is shipped by
                           ``torch.distributed`` (that is all PyTorch does here);
* ``rccl_all_transports``  RCCL for one process driving N GPUs (ncclCommInitAll), one host thread per rank;
* ``DistTransport``        a callback transport over ``torch.distributed`` -- gloo in the world_size-2/3 CPU tests
                           (kernels under the SIMT emulator, "device" memory = host memory), and the fall-back of
                           ``bench.py`` should RCCL refuse to initialise;
* ``run_ranks``            run one function per rank on host threads (loop-back tests).
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import abi

_vp = C.c_void_p
RCCL_ID_BYTES = 128
SLAB_LOOPBACK = 1

ALLREDUCE_T = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_int, _vp)
EXCHANGE_T = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int, _vp)
ALLGATHER_T = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_size_t, _vp)
ALLGATHER_HOST_T = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_size_t)
DESTROY_T = C.CFUNCTYPE(None, _vp)
ABORT_T = C.CFUNCTYPE(None, _vp)


class Transport(C.Structure):              # sift3d_amd_transport
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("self", _vp), ("allreduce_max", ALLREDUCE_T),
                ("exchange", EXCHANGE_T), ("allgather", ALLGATHER_T), ("allgather_host", ALLGATHER_HOST_T),
                ("destroy", DESTROY_T), ("abort", ABORT_T)]


class SlabInfo(C.Structure):               # sift3d_amd_slab_info
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("z0", C.c_int), ("z1", C.c_int), ("o_shard", C.c_int),
                ("halo", C.c_int), ("num_octaves", C.c_int), ("num_levels", C.c_int), ("num_candidates", C.c_long),
                ("num_keypoints", C.c_long), ("halo_bytes", C.c_double), ("device_bytes", C.c_double), ("detect_ms", C.c_double),
                ("describe_ms", C.c_double), ("comm_ms", C.c_double), ("halo_wait_ms", C.c_double),
                ("num_described", C.c_long), ("plan_send_lo_bytes", C.c_double), ("plan_send_hi_bytes", C.c_double),
                ("plan_seed_gather_bytes", C.c_double), ("plan_halo_planes", C.c_int * 8)]


def _last_errors(L) -> str:
    """What the C driver and the device layer (incl. RCCL: "<call>: <ncclGetErrorString> -- <ncclGetLastError>") last said on
    the calling thread -- a first-contact problem with a fabric has to be readable from a failed run's stderr."""
    out = []
    for fn in ("sift3d_amd_slab_last_error", "s3d_rt_last_error"):
        try:
            f = getattr(L, fn)
            f.restype = C.c_char_p
            m = f()
            if m:
                out.append(m.decode("utf-8", "replace"))
        except Exception:                                   # noqa: BLE001
            pass
    return " | ".join(out) if out else "(no message)"


def bind(L: C.CDLL) -> C.CDLL:
    P = C.POINTER
    if getattr(L, "_slab_bound", False):
        return L
    L.sift3d_amd_loopback_create.argtypes = [C.c_int, P(Transport)]
    L.sift3d_amd_slab_create.argtypes = [P(_vp), P(abi.SIFT3D), P(Transport), C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_double, C.c_double, _vp]
    L.sift3d_amd_slab_destroy.argtypes = [_vp]
    L.sift3d_amd_slab_destroy.restype = None
    L.sift3d_amd_slab_get_info.argtypes = [_vp, P(SlabInfo)]
    L.sift3d_amd_slab_detect.argtypes = [_vp, _vp, C.c_int, P(abi.Keypoint_store)]
    L.sift3d_amd_slab_describe.argtypes = [_vp, P(abi.Keypoint_store), P(abi.SIFT3D_Descriptor_store), P(_vp)]
    L.sift3d_amd_slab_gather.argtypes = [_vp, P(abi.Keypoint_store), P(abi.SIFT3D_Descriptor_store),
                                         P(abi.Keypoint_store), P(abi.SIFT3D_Descriptor_store)]
    L.sift3d_amd_set_num_gpus.argtypes = [P(abi.SIFT3D), C.c_int, C.c_int]
    L.sift3d_amd_get_slab_info.argtypes = [P(abi.SIFT3D), C.c_int, P(SlabInfo)]
    L.init_SIFT3D.argtypes = [P(abi.SIFT3D)]
    L.cleanup_SIFT3D.argtypes = [P(abi.SIFT3D)]
    L.cleanup_SIFT3D.restype = None
    for f in ("set_peak_thresh_SIFT3D", "set_corner_thresh_SIFT3D", "set_sigma_n_SIFT3D", "set_sigma0_SIFT3D"):
        getattr(L, f).argtypes = [P(abi.SIFT3D), C.c_double]
    L.set_num_kp_levels_SIFT3D.argtypes = [P(abi.SIFT3D), C.c_uint]
    L.init_Keypoint_store.argtypes = [P(abi.Keypoint_store)]
    L.init_Keypoint_store.restype = None
    L.cleanup_Keypoint_store.argtypes = [P(abi.Keypoint_store)]
    L.cleanup_Keypoint_store.restype = None
    L.init_SIFT3D_Descriptor_store.argtypes = [P(abi.SIFT3D_Descriptor_store)]
    L.init_SIFT3D_Descriptor_store.restype = None
    L.cleanup_SIFT3D_Descriptor_store.argtypes = [P(abi.SIFT3D_Descriptor_store)]
    L.cleanup_SIFT3D_Descriptor_store.restype = None
    L.s3d_rt_d2d.argtypes = [_vp, _vp, C.c_size_t, _vp]
    L.s3d_rt_sync.argtypes = [_vp]
    L.s3d_rt_last_error.restype = C.c_char_p
    L.sift3d_amd_last_error.restype = C.c_char_p
    if hasattr(L, "sift3d_amd_slab_test_inject"):          # the TESTING build of the library only
        L.sift3d_amd_slab_test_inject.argtypes = [C.c_int, C.c_int]
        L.sift3d_amd_slab_test_inject.restype = None
    if hasattr(L, "sift3d_amd_rccl_unique_id"):          # absent from builds without the RCCL transport
        L.sift3d_amd_rccl_unique_id.argtypes = [C.c_char_p]
        L.sift3d_amd_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int, P(Transport)]
        L.sift3d_amd_rccl_create_all.argtypes = [C.c_int, P(C.c_int), P(Transport)]
        L.sift3d_amd_rccl_info.argtypes = [P(Transport), P(C.c_int), P(C.c_int)]
    L._slab_bound = True
    return L


def make_params(L: C.CDLL, params: dict | None = None) -> abi.SIFT3D:
    """A SIFT3D struct carrying the parameters (peak_thresh, corner_thresh, num_kp_levels, sigma_n, sigma0)."""
    bind(L)
    s = abi.SIFT3D()
    if L.init_SIFT3D(C.byref(s)) != 0:
        raise RuntimeError("init_SIFT3D failed")
    for k, v in (params or {}).items():
        if getattr(L, f"set_{k}_SIFT3D")(C.byref(s), v) != 0:
            raise ValueError(f"bad parameter {k}={v}")
    return s


# ---- transports --------------------------------------------------------------------------------------------------
def loopback_transports(L: C.CDLL, world: int):
    bind(L)
    arr = (Transport * world)()
    if L.sift3d_amd_loopback_create(world, arr) != 0:
        raise RuntimeError("sift3d_amd_loopback_create failed")
    return arr


def rccl_all_transports(L: C.CDLL, world: int, devices=None):
    """RCCL communicators for `world` GPUs of THIS process (ncclCommInitAll): transport r drives devices[r]; the caller
    runs one host thread per rank with that device current.  No launcher, no torch.distributed."""
    bind(L)
    arr = (Transport * world)()
    devs = (C.c_int * world)(*(devices if devices is not None else range(world)))
    if L.sift3d_amd_rccl_create_all(world, devs, arr) != 0:
        raise RuntimeError("sift3d_amd_rccl_create_all: " + (L.s3d_rt_last_error() or b"").decode())
    return arr


def rccl_info(L: C.CDLL, t: Transport):
    """(ranks of the communicator, RCCL version code) as the library reports them."""
    bind(L)
    n, v = C.c_int(), C.c_int()
    if L.sift3d_amd_rccl_info(C.byref(t), C.byref(n), C.byref(v)) != 0:
        raise RuntimeError("sift3d_amd_rccl_info: " + (L.s3d_rt_last_error() or b"").decode())
    return n.value, v.value


def rccl_transport(L: C.CDLL, dist, rank: int, world: int) -> Transport:
    """RCCL communicators for this process's current HIP device; `dist` (torch.distributed, any backend -- gloo will
    do) only carries rank 0's unique id to the other ranks."""
    import torch
    bind(L)
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    if rank == 0 and L.sift3d_amd_rccl_unique_id(buf) != 0:
        raise RuntimeError("sift3d_amd_rccl_unique_id: " + (L.s3d_rt_last_error() or b"").decode())
    box = [bytes(buf.raw)]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    t = Transport()
    if L.sift3d_amd_rccl_create(box[0], rank, world, C.byref(t)) != 0:
        raise RuntimeError("sift3d_amd_rccl_create: " + (L.s3d_rt_last_error() or b"").decode())
    del torch
    return t


class DistTransport:
    """sift3d_amd_transport whose operations are torch.distributed calls.  device=None: "device" pointers are host
    addresses (CPU emulator + gloo).  device="cuda:N": payloads are staged through torch tensors of that device
    (backend nccl = RCCL) or, with stage_via_host, through host memory (gloo between processes that share a GPU)."""

    def __init__(self, L: C.CDLL, dist, device=None, stage_via_host: bool = False):
        import torch
        self.torch, self.L, self.dist = torch, bind(L), dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device, self.stage = device, stage_via_host
        self.bytes_sent = 0
        self._keep = (ALLREDUCE_T(self._allreduce_max), EXCHANGE_T(self._exchange), ALLGATHER_T(self._allgather),
                      ALLGATHER_HOST_T(self._allgather_host), DESTROY_T(lambda _s: None))
        self.struct = Transport(self.rank, self.world, None, *self._keep)

    # -- raw pointer <-> tensor ------------------------------------------------------------------------------------
    def _host_view(self, ptr, nbytes):
        return self.torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))

    def _fetch(self, ptr, nbytes, stream):
        """A tensor holding the nbytes at device address ptr, on the device the collectives run on."""
        if self.device is None:
            return self._host_view(ptr, nbytes)
        t = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)
        assert self.L.s3d_rt_d2d(t.data_ptr(), ptr, nbytes, stream) == 0 and self.L.s3d_rt_sync(stream) == 0
        return t.cpu() if self.stage else t

    def _store(self, ptr, t, stream):
        if self.device is None:
            return                                              # received in place
        t = t.to(self.device)
        assert self.L.s3d_rt_d2d(ptr, t.data_ptr(), t.numel(), stream) == 0 and self.L.s3d_rt_sync(stream) == 0

    def _recv_buf(self, ptr, nbytes):
        if self.device is None:
            return self._host_view(ptr, nbytes)
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device="cpu" if self.stage else self.device)

    # -- the five operations -----------------------------------------------------------------------------------------
    def _allreduce_max(self, _self, d_buf, n, stream):
        try:
            self.L.s3d_rt_sync(stream)
            t = self._fetch(d_buf, 4 * n, stream).view(self.torch.float32)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            self._store(d_buf, t.view(self.torch.uint8), stream)
            return 0
        except Exception as e:                                  # noqa: BLE001 -- must not unwind through C
            print("DistTransport.allreduce_max:", e, flush=True)
            return -1

    def _exchange(self, _self, send_lo, recv_lo, send_hi, recv_hi, nbytes, _lane, stream):
        try:
            d, r, w = self.dist, self.rank, self.world
            self.L.s3d_rt_sync(stream)
            ops, back = [], []
            if r > 0:
                ops.append(d.P2POp(d.isend, self._fetch(send_lo, nbytes, stream), r - 1))
                b = self._recv_buf(recv_lo, nbytes)
                back.append((recv_lo, b))
                ops.append(d.P2POp(d.irecv, b, r - 1))
                self.bytes_sent += nbytes
            if r < w - 1:
                ops.append(d.P2POp(d.isend, self._fetch(send_hi, nbytes, stream), r + 1))
                b = self._recv_buf(recv_hi, nbytes)
                back.append((recv_hi, b))
                ops.append(d.P2POp(d.irecv, b, r + 1))
                self.bytes_sent += nbytes
            for q in (d.batch_isend_irecv(ops) if ops else []):
                q.wait()
            for ptr, b in back:
                self._store(ptr, b, stream)
            return 0
        except Exception as e:                                  # noqa: BLE001
            print("DistTransport.exchange:", e, flush=True)
            return -1

    def _allgather(self, _self, d_send, d_recv, nbytes, stream):
        try:
            self.L.s3d_rt_sync(stream)
            src = self._fetch(d_send, nbytes, stream).clone()
            out = [self.torch.empty_like(src) for _ in range(self.world)]
            self.dist.all_gather(out, src)
            cat = self.torch.cat(out)
            if self.device is None:
                self._host_view(d_recv, nbytes * self.world).copy_(cat)
            else:
                self._store(d_recv, cat, stream)
            self.bytes_sent += nbytes * (self.world - 1)
            return 0
        except Exception as e:                                  # noqa: BLE001
            print("DistTransport.allgather:", e, flush=True)
            return -1

    def _allgather_host(self, _self, send, recv, nbytes):
        try:
            src = self._host_view(send, nbytes).clone()
            if self.device is not None and not self.stage:
                src = src.to(self.device)
            out = [self.torch.empty_like(src) for _ in range(self.world)]
            self.dist.all_gather(out, src)
            self._host_view(recv, nbytes * self.world).copy_(self.torch.cat(out).cpu())
            return 0
        except Exception as e:                                  # noqa: BLE001
            print("DistTransport.allgather_host:", e, flush=True)
            return -1


# ---- one rank ------------------------------------------------------------------------------------------------------
class Slab:
    """sift3d_amd_slab: one rank's share of a Z-slab job (the calling thread's current HIP device is its GPU)."""

    def __init__(self, L: C.CDLL, transport: Transport, nx: int, ny: int, nz: int, units=(1.0, 1.0, 1.0),
                 params: dict | None = None, stream=None):
        self.L = bind(L)
        self.transport = transport                      # must outlive the slab
        self._p = make_params(L, params)
        self.h = _vp()
        rc = L.sift3d_amd_slab_create(C.byref(self.h), C.byref(self._p), C.byref(transport), nx, ny, nz,
                                      float(units[0]), float(units[1]), float(units[2]), stream)
        if rc != 0:
            raise ValueError("sift3d_amd_slab_create failed (slabs thinner than the descriptor halo?)")
        self.kp = abi.Keypoint_store()
        L.init_Keypoint_store(C.byref(self.kp))
        self.desc = abi.SIFT3D_Descriptor_store()
        L.init_SIFT3D_Descriptor_store(C.byref(self.desc))
        self.d_desc = _vp()

    def info(self) -> SlabInfo:
        i = SlabInfo()
        self.L.sift3d_amd_slab_get_info(self.h, C.byref(i))
        return i

    def detect(self, vol, on_device: bool) -> int:
        """vol: device address (on_device) or a float32 numpy array of this rank's base slices [z0, z1)."""
        ptr = vol if on_device else np.ascontiguousarray(vol, np.float32).ctypes.data
        if self.L.sift3d_amd_slab_detect(self.h, ptr, 1 if on_device else 0, C.byref(self.kp)) != 0:
            raise RuntimeError("sift3d_amd_slab_detect failed: " + _last_errors(self.L))
        return int(self.kp.slab.num)

    def describe(self, to_host: bool = True) -> int:
        """Descriptors of self.kp; returns the device address of the 776-float records."""
        rc = self.L.sift3d_amd_slab_describe(self.h, C.byref(self.kp), C.byref(self.desc) if to_host else None,
                                             C.byref(self.d_desc))
        if rc != 0:
            raise RuntimeError("sift3d_amd_slab_describe failed: " + _last_errors(self.L))
        return self.d_desc.value or 0

    def gather(self, with_desc: bool = True):
        """(Keypoint_store, SIFT3D_Descriptor_store | None) of the whole volume in the reference order; collective."""
        kp_all = abi.Keypoint_store()
        self.L.init_Keypoint_store(C.byref(kp_all))
        d_all = abi.SIFT3D_Descriptor_store()
        self.L.init_SIFT3D_Descriptor_store(C.byref(d_all))
        rc = self.L.sift3d_amd_slab_gather(self.h, C.byref(self.kp), C.byref(self.desc) if with_desc else None,
                                           C.byref(kp_all), C.byref(d_all) if with_desc else None)
        if rc != 0:
            raise RuntimeError("sift3d_amd_slab_gather failed")
        return kp_all, (d_all if with_desc else None)

    def close(self):
        if self.h:
            self.L.sift3d_amd_slab_destroy(self.h)
            self.h = _vp()
            self.L.cleanup_Keypoint_store(C.byref(self.kp))
            self.L.cleanup_SIFT3D_Descriptor_store(C.byref(self.desc))
            self.L.cleanup_SIFT3D(C.byref(self._p))


def run_ranks(world: int, fn):
    """fn(rank) on `world` host threads (the loop-back transport's ranks); returns the list of results, re-raising
    the first failure."""
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:                              # noqa: BLE001
            err[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out
