"""Z-slab sharded detect + describe across the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests where the same kernels run under the SIMT emulator).  PyTorch is used for device
buffers and the collectives only; all arithmetic is the library's HIP kernels, called through the
flat C-ABI of ``include/s3d_device.h`` with plain device pointers.

Decomposition.  Rank r of G owns the base slices [r*NZ/G, (r+1)*NZ/G); at octave o its slab is those
indices >> o.  Octaves whose slab is still thicker than the descriptor window (H planes) are SHARDED:
every GSS level is stored as slab + 2*H halo planes and addressed through a *view* pointer indexed by
global z, so the single-GPU kernels work unchanged on it (they only need the global depth for the
reference's mirror rule at the two global ends).  Coarser octaves are REPLICATED on every rank
(<= 1/64 of the data) and only the work (extrema / keypoints) is partitioned by z.

Exchanges (all with the two Z-neighbours only, point to point -- one xGMI link per pair):
  * after a sharded level is produced: its boundary planes -> the neighbours' halos
    (H planes for the levels descriptors are taken from, the next filter's reach otherwise),
  * all_reduce(MAX) of one float for im_scale and per DoG level for the peak threshold,
  * all_gather of the decimated slab that seeds the first replicated octave.
X and Y passes, decimation, DoG and scaling are slab-local; halo planes of a level are also run through
the fused X+Y pass locally (<= 2*hw/slab extra work) so each Gaussian needs exactly one exchange.

Results are bit-identical to the single-GPU path (same kernels, same global indices); keypoints come
out ordered (o, s, z, y, x) within a rank and ranks are ordered by z, so concatenating the ranks'
lists per (o, s) reproduces the reference order (``gather_keypoints``).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import abi

S3D_MAX_OCTAVES, S3D_MAX_LEVELS = 16, 16
DESC_REC_FLOATS = 776
_vp = C.c_void_p
_f32p = C.POINTER(C.c_float)


class PyramidDesc(C.Structure):            # s3d_pyramid_desc, include/s3d_device.h
    _fields_ = [("d_level", C.c_uint64 * (S3D_MAX_OCTAVES * S3D_MAX_LEVELS)),
                ("dims", (C.c_int * 3) * S3D_MAX_OCTAVES), ("unitsf", (C.c_float * 3) * S3D_MAX_OCTAVES),
                ("num_octaves", C.c_int), ("num_levels", C.c_int), ("first_level", C.c_int)]


class DescKey(C.Structure):                # s3d_desc_key
    _fields_ = [("cx", C.c_float), ("cy", C.c_float), ("cz", C.c_float), ("sigma", C.c_float),
                ("rad", C.c_float), ("half", C.c_float), ("binf", C.c_float), ("level", C.c_int),
                ("octave", C.c_int), ("R", C.c_float * 9)]


class Comm:
    """The three collectives the path needs, over torch.distributed (or nothing for one rank)."""

    def __init__(self, dist=None, stage_via_host: bool = False):
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.stage = stage_via_host      # gloo with GPU tensors (single-GPU debugging of the 2-rank path)
        self.bytes_exchanged = 0

    def allreduce_max_(self, t: torch.Tensor) -> None:
        if self.world == 1:
            return
        if self.stage:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)

    def exchange_async(self, send_lo, send_hi, recv_lo, recv_hi):
        """send_lo -> rank-1 (lands in its recv_hi), send_hi -> rank+1 (its recv_lo).  Views of 1-D tensors.
        Posts the transfers and returns a handle for finish(): the receive buffers must not be read (and the send
        buffers not be overwritten) before that.  With RCCL the transfers run on the communicator's stream next to
        whatever is launched afterwards."""
        if self.world == 1:
            return None
        d, r, w = self.dist, self.rank, self.world
        ops, stage_back = [], []

        def snd(t, peer):
            if self.stage:
                t = t.cpu()
            self.bytes_exchanged += t.numel() * 4
            ops.append(d.P2POp(d.isend, t, peer))

        def rcv(t, peer):
            if self.stage:
                h = torch.empty(t.shape, dtype=t.dtype)
                stage_back.append((t, h))
                t = h
            ops.append(d.P2POp(d.irecv, t, peer))

        if r > 0 and send_lo is not None:
            snd(send_lo, r - 1)
        if r < w - 1 and send_hi is not None:
            snd(send_hi, r + 1)
        if r > 0 and recv_lo is not None:
            rcv(recv_lo, r - 1)
        if r < w - 1 and recv_hi is not None:
            rcv(recv_hi, r + 1)
        works = d.batch_isend_irecv(ops) if ops else []
        return works, stage_back, ops          # ops: keeps the (possibly staged) send tensors alive until finish()

    def finish(self, handle) -> None:
        if handle is None:
            return
        works, stage_back, _ = handle
        for q in works:
            q.wait()
        for t, h in stage_back:
            t.copy_(h)

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi) -> None:
        self.finish(self.exchange_async(send_lo, send_hi, recv_lo, recv_hi))

    def allgather_cat(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenate equally sized 1-D tensors of all ranks in rank order."""
        if self.world == 1:
            return t
        src = t.cpu() if self.stage else t
        out = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(out, src)
        self.bytes_exchanged += t.numel() * 4 * (self.world - 1)
        return torch.cat(out).to(t.device)

    def allgather_object(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


class _Level:
    """A GSS level: backing tensor + the global-z view pointer the kernels take."""

    def __init__(self, planes_lo: int, nplanes: int, plane_elems: int, device):
        self.zlo = planes_lo                  # global z of the first backed plane
        self.n = nplanes
        self.pe = plane_elems
        self.t = torch.zeros(nplanes * plane_elems + 16, dtype=torch.float32, device=device)   # +16: k_describe's wide loads
        self.view = self.t.data_ptr() - planes_lo * plane_elems * 4

    def planes(self, za: int, zb: int) -> torch.Tensor:   # backed planes [za, zb), global z
        return self.t[(za - self.zlo) * self.pe:(zb - self.zlo) * self.pe]

    def ptr(self, z: int) -> int:
        return self.view + z * self.pe * 4


class SlabSift3D:
    def __init__(self, cdll: C.CDLL, device, comm: Comm, nx: int, ny: int, nz: int, units=(1.0, 1.0, 1.0),
                 params: dict | None = None):
        self.L = cdll
        self.dev = torch.device(device)
        self.comm = comm
        self.nx, self.ny, self.NZ = nx, ny, nz
        self.units = tuple(float(u) for u in units)
        self._bind()
        G, r = comm.world, comm.rank
        if nz % G:
            raise ValueError("nz must be divisible by the number of ranks")
        # ---- plan on the host with the library's own SIFT3D object ---------------------------------
        s = self.s = abi.SIFT3D()
        assert cdll.init_SIFT3D(C.byref(s)) == 0
        for k, v in (params or {}).items():
            assert getattr(cdll, f"set_{k}_SIFT3D")(C.byref(s), v) == 0
        if cdll.sift3d_amd_plan(C.byref(s), nx, ny, nz, *self.units) != 0:
            raise RuntimeError("sift3d_amd_plan failed")
        g = s.gpyr
        self.no, self.nl, self.nkp, self.first_level = g.num_octaves, g.num_levels, g.num_kp_levels, g.first_level
        self.dims = [(g.levels[o * self.nl].nx, g.levels[o * self.nl].ny, g.levels[o * self.nl].nz) for o in range(self.no)]
        self.lunits = [(g.levels[o * self.nl].ux, g.levels[o * self.nl].uy, g.levels[o * self.nl].uz) for o in range(self.no)]
        self.scale = [[g.levels[o * self.nl + k].s for k in range(self.nl)] for o in range(self.no)]

        def taps_of(f):
            return np.ctypeslib.as_array(f.kernel, shape=(f.width,)).copy()
        self.taps_first = taps_of(s.gss.first_gauss.f)
        self.taps = [taps_of(s.gss.gauss_octave[k].f) for k in range(self.nl - 1)]
        # ---- decomposition ----------------------------------------------------------------------------
        # H: planes a descriptor window (+1 for the gradient) can reach beyond its centre, in octave voxels
        sd_max = g.sigma0 * 2.0 ** ((self.nkp - 1) / self.nkp)
        self.H = int(math.ceil(2.0 * 7.071067812 * sd_max / self.units[2])) + 3
        slab0 = nz // G
        self.o_shard = -1
        if G == 1:
            self.o_shard = self.no - 1
        else:
            for o in range(self.no):
                if slab0 % (1 << o) == 0 and (slab0 >> o) >= self.H and self.dims[o][2] == nz >> o:
                    self.o_shard = o
                else:
                    break
            if self.o_shard < 0:
                raise ValueError(f"slab of {slab0} slices is thinner than the descriptor halo ({self.H}); "
                                 "use fewer ranks or a deeper volume")
        self.part = []                               # work partition [z0, z1) per octave
        for o in range(self.no):
            nzo = self.dims[o][2]
            if o <= self.o_shard:
                self.part.append(((r * slab0) >> o, ((r + 1) * slab0) >> o))
            else:
                self.part.append((r * nzo // G, (r + 1) * nzo // G))
        # ---- buffers -------------------------------------------------------------------------------------
        self.lev = []
        for o in range(self.no):
            nxo, nyo, nzo = self.dims[o]
            row = []
            for k in range(self.nl):
                if o <= self.o_shard and G > 1:
                    z0, z1 = self.part[o]
                    row.append(_Level(z0 - self.H, (z1 - z0) + 2 * self.H, nxo * nyo, self.dev))
                else:
                    row.append(_Level(0, nzo, nxo * nyo, self.dev))
            self.lev.append(row)
        z0, z1 = self.part[0]
        hal = self.H if G > 1 else 0
        self.im = _Level(z0 - hal, (z1 - z0) + 2 * hal, nx * ny, self.dev)
        self.tmp = _Level(z0 - hal, (z1 - z0) + 2 * hal, nx * ny, self.dev)      # big enough for every octave
        nmax = (z1 - z0) * nx * ny
        self.bits_words = nmax // 64 + 2
        self.bits = torch.zeros(3 * self.bits_words, dtype=torch.int64, device=self.dev)   # one bitmap per keypoint level
        self.scratch = torch.zeros(nmax // 64 // 256 + 4096, dtype=torch.int32, device=self.dev)
        self.orient_scr = None
        self._pending = []                       # deferred halo transfers of the current detect()
        self.red = torch.zeros(8, dtype=torch.float32, device=self.dev)
        self.count = torch.zeros(8, dtype=torch.int32, device=self.dev)
        self.cap = 0
        mesh = np.zeros(20 * 16 + 32, np.float32)
        cdll.s3d_mesh_table(mesh.ctypes.data_as(_f32p))
        self.mesh = torch.from_numpy(mesh).to(self.dev)
        self.pd = PyramidDesc()
        self.pd.num_octaves, self.pd.num_levels, self.pd.first_level = self.no, self.nl, self.first_level
        for o in range(self.no):
            for a in range(3):
                self.pd.dims[o][a] = self.dims[o][a]
                self.pd.unitsf[o][a] = np.float32(self.lunits[o][a])
            for k in range(self.nl):
                self.pd.d_level[o * self.nl + k] = self.lev[o][k].view
        self.sigma_tab = torch.tensor([1.5 * self.scale[o][k] for o in range(self.no) for k in range(self.nl)],
                                      dtype=torch.float64, device=self.dev)
        self.xyzos = np.zeros((0, 5), np.int32)
        self.R = np.zeros((0, 3, 3), np.float32)
        self.num_candidates = 0

    # ---- plumbing -------------------------------------------------------------------------------------------
    def _bind(self):
        L = self.L
        P = C.POINTER
        L.sift3d_amd_plan.argtypes = [P(abi.SIFT3D), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.init_SIFT3D.argtypes = [P(abi.SIFT3D)]
        for f in ("set_peak_thresh_SIFT3D", "set_corner_thresh_SIFT3D", "set_sigma_n_SIFT3D", "set_sigma0_SIFT3D"):
            getattr(L, f).argtypes = [P(abi.SIFT3D), C.c_double]
        L.set_num_kp_levels_SIFT3D.argtypes = [P(abi.SIFT3D), C.c_uint]
        L.s3d_rt_last_error.restype = C.c_char_p
        L.s3d_mesh_table.argtypes = [_f32p]
        L.s3d_k_absmax.argtypes = [_vp, C.c_size_t, _vp, _vp]
        L.s3d_k_dogmax.argtypes = [_vp, _vp, C.c_size_t, _vp, _vp]
        L.s3d_k_dogmax3.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, _vp, _vp]
        L.s3d_k_scale_div.argtypes = [_vp, C.c_size_t, _vp, _vp]
        L.s3d_k_decimate2.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]
        L.s3d_k_sep_fir.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_int, _vp]
        L.s3d_k_sep_fir_slab.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p,
                                         C.c_int, _vp]
        L.s3d_k_extrema_fused.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, _vp, C.POINTER(C.c_void_p), _vp]
        L.s3d_k_extrema_slab.argtypes = [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                         _vp, _vp, _vp]
        L.s3d_k_compact_bits_base.argtypes = [_vp, C.c_size_t, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp,
                                              _vp]
        L.s3d_k_orient.argtypes = [P(PyramidDesc), _vp, _vp, _vp, C.c_uint32, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]
        L.s3d_k_orient_scratch_bytes.argtypes = [C.c_uint32]
        L.s3d_k_orient_scratch_bytes.restype = C.c_size_t
        L.s3d_k_compact_keys.argtypes = [P(PyramidDesc), _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]
        L.s3d_k_describe.argtypes = [P(PyramidDesc), _vp, C.c_uint32, _vp, _vp, C.c_size_t, _vp]
        L.s3d_rt_sync.argtypes = [_vp]

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {(self.L.s3d_rt_last_error() or b'').decode()}")

    def _uf(self, o):
        u = self.lunits[o]
        return np.array([np.float32(1.0 / u[0]), np.float32(1.0 / u[1]), np.float32(1.0 / u[2])], np.float32)

    def _reach(self, taps, o):                    # planes of z halo one application needs (s3d_k_sep_fir_slab)
        hw, uf = len(taps) // 2, self._uf(o)
        if uf[0] == 1.0 and uf[1] == 1.0 and uf[2] == 1.0:
            return hw                             # fused unit-spacing path: exact
        return int(math.ceil(np.float32(hw) * uf[2])) + 1      # + 1: the reference's drifting tap coordinate

    def _exchange(self, lv: _Level, o: int, h: int, now: int | None = None):
        """Fill h halo planes on each interior side of a sharded level from the Z-neighbours.  now < h: only the
        `now` planes next to the slab are waited for (what the next Gaussian and the extrema read); the outer
        h - now planes -- the orientation / descriptor windows, read only in _keypoints() -- travel while the rest of
        the pyramid is computed and are collected by _finish_halos()."""
        if self.comm.world == 1 or h <= 0:
            return
        z0, z1 = self.part[o]
        n = h if now is None or now >= h else max(now, 1)
        self.comm.exchange(lv.planes(z0, z0 + n), lv.planes(z1 - n, z1), lv.planes(z0 - n, z0), lv.planes(z1, z1 + n))
        if n < h:
            self._pending.append(self.comm.exchange_async(lv.planes(z0 + n, z0 + h), lv.planes(z1 - h, z1 - n),
                                                          lv.planes(z0 - h, z0 - n), lv.planes(z1 + n, z1 + h)))

    def _finish_halos(self):
        for hnd in self._pending:
            self.comm.finish(hnd)
        self._pending = []

    def _gauss(self, src: _Level, dst: _Level, o: int, taps):
        nxo, nyo, nzo = self.dims[o]
        uf = self._uf(o)
        t = np.ascontiguousarray(taps, np.float32)
        if o <= self.o_shard and self.comm.world > 1:
            z0, z1 = self.part[o]
            tmpv = self.tmp.t.data_ptr() - (z0 - self.H) * nxo * nyo * 4
            self._ck(self.L.s3d_k_sep_fir_slab(src.view, dst.view, tmpv, nxo, nyo, nzo, z0, z1,
                                               uf.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), t.size, None),
                     "s3d_k_sep_fir_slab")
        else:
            self._ck(self.L.s3d_k_sep_fir(src.view, dst.view, self.tmp.t.data_ptr(), nxo, nyo, nzo, 1,
                                          uf.ctypes.data_as(_f32p), t.ctypes.data_as(_f32p), t.size, None),
                     "s3d_k_sep_fir")

    # ---- the path ---------------------------------------------------------------------------------------------
    def detect(self, vol_slab: torch.Tensor) -> int:
        """vol_slab: this rank's base slices [z0, z1) as a float32 tensor [z1-z0, ny, nx] on self.dev.
        Returns the number of keypoints this rank owns."""
        L, comm = self.L, self.comm
        sharded = comm.world > 1
        z0, z1 = self.part[0]
        n_local = (z1 - z0) * self.nx * self.ny
        assert vol_slab.numel() == n_local and vol_slab.dtype == torch.float32
        own = self.im.planes(z0, z1)
        own.copy_(vol_slab.reshape(-1))
        # im_scale with the global maximum (sift.c:903, imutil.c:1977-1991)
        self._ck(L.s3d_k_absmax(own.data_ptr(), n_local, self.red.data_ptr(), None), "absmax")
        comm.allreduce_max_(self.red[0:1])
        self._ck(L.s3d_k_scale_div(own.data_ptr(), n_local, self.red.data_ptr(), None), "scale_div")
        # build_gpyr (sift.c:989-1050)
        lev = self.lev
        self._exchange(self.im, 0, self._reach(self.taps_first, 0))
        self._gauss(self.im, lev[0][0], 0, self.taps_first)
        for o in range(self.no):
            shard_o = sharded and o <= self.o_shard
            for k in range(1, self.nl):
                if shard_o:     # the next Gaussian reads `reach` planes, the extrema one; the rest may arrive later
                    self._exchange(lev[o][k - 1], o, self._halo_of_level(o, k - 1),
                                   now=max(1, self._reach(self.taps[k - 1], o)))
                self._gauss(lev[o][k - 1], lev[o][k], o, self.taps[k - 1])
            if shard_o:
                self._exchange(lev[o][self.nl - 1], o, self._halo_of_level(o, self.nl - 1))
            if o + 1 < self.no:
                ds = max(self.nl - 3, 0)                       # level index of s_end - 2
                nxo, nyo, nzo = self.dims[o]
                nxn, nyn, nzn = self.dims[o + 1]
                if sharded and o + 1 <= self.o_shard:          # slab-local decimation
                    a, b = self.part[o + 1]
                    self._ck(L.s3d_k_decimate2(lev[o][ds].ptr(2 * a), nxo, nyo, 2 * (b - a), lev[o + 1][0].ptr(a), None),
                             "decimate2")
                elif sharded and o == self.o_shard:            # seed the first replicated octave
                    zs0, zs1 = self.part[o]
                    a, b = zs0 // 2, zs1 // 2
                    part = torch.empty((b - a) * nxn * nyn, dtype=torch.float32, device=self.dev)
                    self._ck(L.s3d_k_decimate2(lev[o][ds].ptr(2 * a), nxo, nyo, 2 * (b - a), part.data_ptr(), None),
                             "decimate2")
                    full = comm.allgather_cat(part)
                    lev[o + 1][0].t[:full.numel()].copy_(full)
                else:
                    self._ck(L.s3d_k_decimate2(lev[o][ds].view, nxo, nyo, nzo, lev[o + 1][0].view, None), "decimate2")
        self._finish_halos()
        return self._keypoints()

    def _halo_of_level(self, o: int, k: int) -> int:
        """Halo planes level k of a sharded octave needs from each neighbour once it is complete."""
        h = 1                                                   # extrema look at z +- 1
        if k + 1 < self.nl:
            h = max(h, self._reach(self.taps[k], o))            # next Gaussian's reach
        if 1 <= k <= self.nkp:                                  # levels s = 0..nkp-1: orientation + descriptor windows
            h = max(h, self.H)
        return h

    def _keypoints(self) -> int:
        L, comm = self.L, self.comm
        lev = self.lev
        # generous first guess, grown on overflow
        nloc = sum((self.part[o][1] - self.part[o][0]) * self.dims[o][0] * self.dims[o][1] for o in range(self.no))
        cap = self.cap or (nloc // 128 + 4096)
        while True:
            if cap != self.cap:
                self.cand_idx = torch.zeros(cap, dtype=torch.int32, device=self.dev)
                self.cand_tag = torch.zeros(cap, dtype=torch.int32, device=self.dev)
                self.keep = torch.zeros(cap, dtype=torch.int32, device=self.dev)
                self.Rc = torch.zeros(cap * 9, dtype=torch.float32, device=self.dev)
                self.Rk = torch.zeros(cap * 9, dtype=torch.float32, device=self.dev)
                self.kxyzos = torch.zeros(cap * 5, dtype=torch.int32, device=self.dev)
                self.cap = cap
            self.count.zero_()
            for o in range(self.no):
                nxo, nyo, nzo = self.dims[o]
                pe = nxo * nyo
                za, zb = self.part[o]
                shard_o = comm.world > 1 and o <= self.o_shard
                if zb <= za:
                    continue
                nwords = ((zb - za) * pe + 63) // 64
                fused = self.nkp == 3 and nxo % 4 == 0          # all keypoint levels in one pass (s3d_k_extrema_fused)
                if fused:           # the three dogmax values in one pass over GSS levels 1..4 (s3d_k_dogmax3)
                    l4 = (C.c_void_p * 4)(*[(lev[o][k].ptr(za) if shard_o else lev[o][k].view) for k in range(1, 5)])
                    self._ck(L.s3d_k_dogmax3(l4, ((zb - za) if shard_o else nzo) * pe, self.red[1:].data_ptr(), None),
                             "dogmax3")
                for ks in range(1, self.nkp + 1):
                    if fused:
                        break
                    red = self.red[1:]
                    if shard_o:     # max |DoG| over my planes, then over the ranks (sift.c:1161-1169)
                        self._ck(L.s3d_k_dogmax(lev[o][ks].ptr(za), lev[o][ks + 1].ptr(za), (zb - za) * pe,
                                                red.data_ptr(), None), "dogmax")
                        comm.allreduce_max_(self.red[1:2])
                    else:           # replicated octave: every rank sees the whole level
                        self._ck(L.s3d_k_dogmax(lev[o][ks].view, lev[o][ks + 1].view, nzo * pe, red.data_ptr(), None),
                                 "dogmax")
                    self._ck(L.s3d_k_extrema_slab(lev[o][ks - 1].view, lev[o][ks].view, lev[o][ks + 1].view,
                                                  lev[o][ks + 2].view, nxo, nyo, nzo, za, zb, float(self.s.peak_thresh),
                                                  self.red[1:].data_ptr(), self.bits.data_ptr(), None), "extrema")
                    self._ck(L.s3d_k_compact_bits_base(self.bits.data_ptr(), nwords, za * pe, self.cand_idx.data_ptr(),
                                                       self.cand_tag.data_ptr(), (o << 8) | ks, self.cap,
                                                       self.count.data_ptr(), self.scratch.data_ptr(), None), "compact")
                if fused:
                    if shard_o:
                        comm.allreduce_max_(self.red[1:4])         # the three maxima in one collective
                    levels = (C.c_void_p * 6)(*[lev[o][k].view for k in range(6)])
                    bits = (C.c_void_p * 3)(*[self.bits.data_ptr() + 8 * k * self.bits_words for k in range(3)])
                    rc = L.s3d_k_extrema_fused(levels, 3, nxo, nyo, nzo, za, zb, float(self.s.peak_thresh),
                                               self.red[1:].data_ptr(), bits, None)
                    self._ck(rc, "extrema_fused")
                    for ks in range(1, 4):
                        self._ck(L.s3d_k_compact_bits_base(bits[ks - 1], nwords, za * pe, self.cand_idx.data_ptr(),
                                                           self.cand_tag.data_ptr(), (o << 8) | ks, self.cap,
                                                           self.count.data_ptr(), self.scratch.data_ptr(), None),
                                 "compact")
            ncand = int(self.count[0].item())
            # the redo decision must be collective: a rank that looped alone would re-enter the all-reduces
            over = torch.tensor([1.0 if ncand > self.cap else 0.0], dtype=torch.float32, device=self.dev)
            comm.allreduce_max_(over)
            if float(over.item()) == 0.0:
                break
            cap = max(ncand + 1024, self.cap)
        self.num_candidates = ncand
        if ncand == 0:
            self.xyzos = np.zeros((0, 5), np.int32)
            self.R = np.zeros((0, 3, 3), np.float32)
            return 0
        need = int(L.s3d_k_orient_scratch_bytes(ncand))
        if self.orient_scr is None or self.orient_scr.numel() < need:
            self.orient_scr = torch.empty(need, dtype=torch.uint8, device=self.dev)
        self._ck(L.s3d_k_orient(C.byref(self.pd), self.cand_idx.data_ptr(), self.cand_tag.data_ptr(), None, ncand,
                                self.sigma_tab.data_ptr(), float(self.s.corner_thresh), self.Rc.data_ptr(),
                                self.keep.data_ptr(), None, self.orient_scr.data_ptr(), None), "orient")
        self._ck(L.s3d_k_compact_keys(C.byref(self.pd), self.cand_idx.data_ptr(), self.cand_tag.data_ptr(),
                                      self.Rc.data_ptr(), self.keep.data_ptr(), ncand, self.kxyzos.data_ptr(),
                                      self.Rk.data_ptr(), self.count[1:].data_ptr(), self.scratch.data_ptr(), None),
                 "compact_keys")
        K = int(self.count[1].item())
        self.xyzos = self.kxyzos[:5 * K].cpu().numpy().reshape(K, 5).copy()
        self.R = self.Rk[:9 * K].cpu().numpy().reshape(K, 3, 3).copy()
        return K

    def keypoint_scales(self) -> np.ndarray:
        tab = np.array(self.scale, np.float64)                    # [octave][level index]
        return tab[self.xyzos[:, 3], self.xyzos[:, 4] - self.first_level]

    def describe(self) -> torch.Tensor:
        """Descriptors of this rank's keypoints: tensor [K, 776] on the device (768 bins + coordinate slots,
        the layout of SIFT3D_Descriptor)."""
        K = len(self.xyzos)
        out = torch.zeros(max(K, 1) * DESC_REC_FLOATS, dtype=torch.float32, device=self.dev)
        if K == 0:
            return out[:0].reshape(0, DESC_REC_FLOATS)
        # scalar set-up of extract_descrip (sift.c:1845-1851) in the reference's f32/f64 steps, vectorised
        sd = self.keypoint_scales()
        sigma = (sd * 7.071067812).astype(np.float32)
        rad = (2.0 * sigma.astype(np.float64)).astype(np.float32)
        half = (rad.astype(np.float64) / math.sqrt(2.0)).astype(np.float32)
        cell = (np.float32(2.0) * half) / np.float32(4)
        keys = np.zeros(K, dtype=np.dtype([("c", np.float32, 3), ("sigma", np.float32), ("rad", np.float32),
                                           ("half", np.float32), ("binf", np.float32), ("level", np.int32),
                                           ("octave", np.int32), ("R", np.float32, 9)]))
        assert keys.dtype.itemsize == C.sizeof(DescKey)
        keys["c"] = self.xyzos[:, :3].astype(np.float32)
        keys["sigma"], keys["rad"], keys["half"], keys["binf"] = sigma, rad, half, np.float32(1.0) / cell
        keys["level"] = self.xyzos[:, 3] * self.nl + (self.xyzos[:, 4] - self.first_level)
        keys["octave"] = self.xyzos[:, 3]
        keys["R"] = self.R.reshape(K, 9)
        kb = torch.from_numpy(keys.view(np.uint8).reshape(-1)).to(self.dev)
        self._ck(self.L.s3d_k_describe(C.byref(self.pd), kb.data_ptr(), K, self.mesh.data_ptr(), out.data_ptr(),
                                       DESC_REC_FLOATS, None), "describe")
        self._ck(self.L.s3d_rt_sync(None), "sync")
        return out.reshape(K, DESC_REC_FLOATS)

    # ---- assembling the global result (tests, API completeness) ---------------------------------------------------
    def gather_keypoints(self, desc: torch.Tensor | None = None):
        """All ranks receive the global keypoint list in the reference order (o, s, z, y, x)."""
        mine = (self.xyzos, self.R, None if desc is None else desc[:, :768].cpu().numpy())
        parts = self.comm.allgather_object(mine)
        xyzos = np.concatenate([p[0] for p in parts])
        R = np.concatenate([p[1] for p in parts])
        rank_of = np.concatenate([np.full(len(p[0]), i) for i, p in enumerate(parts)])
        local = np.concatenate([np.arange(len(p[0])) for p in parts])
        order = np.lexsort((local, rank_of, xyzos[:, 4], xyzos[:, 3]))
        d = None if desc is None else np.concatenate([p[2] for p in parts])[order]
        return xyzos[order], R[order], d
