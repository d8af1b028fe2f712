#!/usr/bin/env python3
"""bench.py -- detect+describe throughput of the HIP path and the roofline of its dominant kernel.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (one rank per GPU)

N > 1 runs N GPUs either way: launched plainly, ONE process drives the N GPUs with one host thread per rank and RCCL
communicators from ncclCommInitAll (no launcher, no torch); under torchrun every process is one rank, torch.distributed
over GLOO ships rank 0's 128-byte ncclUniqueId and carries the barriers -- the only RCCL in the process is the one
csrc/s3d_rccl.hip opens.  Fewer than N visible GPUs, or a WORLD_SIZE that is not N: exit status 2, no number.

A step = one pass of the hot path over one synthetic volume that is ALREADY RESIDENT IN HBM:
SIFT3D_detect_keypoints (copy + scale + 36 Gaussian applications + extrema + orientation; the small
keypoint list is downloaded because the API returns it) followed by SIFT3D_extract_descriptors for
every keypoint with the descriptors left in HBM.  At N = 1 the workload is BASELINE.json configs[1]:
512^3 float32, "blobs+noise" generator, 128 000 blobs (31 207 keypoints).  At N > 1 the workload is BASELINE
configs[3], the one north_star quotes its multi-GPU figure on: ONE 1024^3 volume sharded by Z-slab over the N GPUs
("scaling": "strong": the work is fixed, nz/N slices per GPU) by the host-C driver of include/sift3d_amd_slab.h
(csrc/host/s3d_host_slab.c): halo planes exchanged with ncclSend/ncclRecv between Z-neighbours, ncclAllReduce(max)
for the scale and peak thresholds, ncclAllGather for the seed of the replicated coarse octaves (SURVEY.md section
8e).  The weak-scaling volume (512 x 512 x 512*N, 512 slices per GPU) rides along untimed as config.weak_512xN, or is
the timed workload with --weak.  `--dry` prints the plan of such a job (partition, halo, sharded octaves, HBM per rank) for
N = 2, 4, 8 without touching a device.  `--replicas` runs one independent volume per rank instead; `--loopback R` runs
R slab ranks on ONE GPU (diagnostic).

One JSON line on stdout (rank 0).  Besides the driver's contract it carries
  roofline      the fused X+Y Gaussian kernel at 512^3 (dominant kernel of the north-star Gaussian),
                timed with HIP events on the stream it is launched on; per-width application
                numbers ride along in config.gauss_apps.
  cpu_baseline  the unmodified reference (oracle/_ref, kind "reference") or, if that is not
                available, the oracle port -- timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sift3d_amd                      # noqa: E402
from sift3d_amd import abi, synth      # noqa: E402

HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GAUSS_APP_BYTES_PER_VOXEL = 24.0       # 3 passes x (4 B read + 4 B write), SURVEY.md 8(d)
GAUSS_XY_BYTES_PER_VOXEL = 16.0        # the fused kernel performs 2 of the 3 algorithmic passes


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def gauss_roofline(dev, n, widths_sigmas, reps=5):
    """Time one Gaussian application per filter of the default bank on an n^3 volume in HBM."""
    from sift3d_amd.device import _vp
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((n, n, n)).astype(np.float32)
    d_src = dev.upload(vol)
    d_dst = dev.malloc(vol.nbytes)
    d_tmp = dev.malloc(vol.nbytes)
    ev = [_vp() for _ in range(3)]
    for e in ev:
        dev.check(dev.L.s3d_rt_event_create(C.byref(e)))
    lib = sift3d_amd.load()
    out = []
    nvox = float(n) ** 3
    for sigma in widths_sigmas:
        g = abi.Gauss_filter()
        assert lib.imutil.init_Gauss_filter(C.byref(g), sigma, 3) == 0
        taps = np.ctypeslib.as_array(g.f.kernel, shape=(g.f.width,)).copy()
        dev.sep_fir(d_src, d_dst, d_tmp, n, n, n, 1, (1, 1, 1), taps, path=2)      # warm-up
        dev.sync()
        t_xy = t_z = 0.0
        for _ in range(reps):
            dev.L.s3d_k_gauss_set_events(ev[0], ev[1], ev[2])
            dev.sep_fir(d_src, d_dst, d_tmp, n, n, n, 1, (1, 1, 1), taps, path=2)
            dev.L.s3d_k_gauss_set_events(None, None, None)
            ms = C.c_float()
            dev.check(dev.L.s3d_rt_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
            t_xy += ms.value
            dev.check(dev.L.s3d_rt_event_elapsed_ms(ev[1], ev[2], C.byref(ms)))
            t_z += ms.value
        t_xy /= reps
        t_z /= reps
        out.append({"width": int(g.f.width), "xy_ms": round(t_xy, 4), "z_ms": round(t_z, 4),
                    "app_GBs": round(GAUSS_APP_BYTES_PER_VOXEL * nvox / ((t_xy + t_z) * 1e-3) / 1e9, 1),
                    "app_frac_of_8TBs": round(GAUSS_APP_BYTES_PER_VOXEL * nvox / ((t_xy + t_z) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        lib.imutil.cleanup_Gauss_filter(C.byref(g))
    for e in ev:
        dev.L.s3d_rt_event_destroy(e)
    for p in (d_src, d_dst, d_tmp):
        dev.free(p)
    return out


def _cpu_baseline_worker(sample_n):
    """Runs in a child process (see cpu_baseline): reference (or port) detect+describe on the host."""
    from oracle import oracle as orc
    vol = synth.blobs(sample_n, sample_n, sample_n, synth.default_nblobs(sample_n, sample_n, sample_n), 0)
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    if orc.have_ref():
        ref = orc.load_ref()
        s = abi.SIFT3D()
        assert ref.sift.init_SIFT3D(C.byref(s)) == 0
        im = ref.image_from_numpy(vol)
        kp = abi.Keypoint_store()
        ref.sift.init_Keypoint_store(C.byref(kp))
        d = abi.SIFT3D_Descriptor_store()
        ref.sift.init_SIFT3D_Descriptor_store(C.byref(d))
        t0 = time.perf_counter()
        assert ref.sift.SIFT3D_detect_keypoints(C.byref(s), C.byref(im), C.byref(kp)) == 0
        if kp.slab.num:
            assert ref.sift.SIFT3D_extract_descriptors(C.byref(s), C.byref(kp), C.byref(d)) == 0
        dt = time.perf_counter() - t0
        k = int(kp.slab.num)
        kind = "reference"
    else:
        O = orc.Oracle()
        t0 = time.perf_counter()
        xyzos, sd, R = O.detect(vol)
        if len(xyzos):
            O.describe(xyzos[:, :3].astype(np.float64), xyzos[:, 3:5], sd, R)
        dt = time.perf_counter() - t0
        k = len(xyzos)
        kind = "port"
    print(json.dumps({"value": round(sample_n ** 3 / dt / 1e6, 4), "unit": "Mvox/s", "cores": threads, "kind": kind,
                      "sample": f"detect+describe on one {sample_n}^3 volume of the same generator ({k} keypoints, "
                                f"{dt:.1f} s, {threads} OpenMP threads)"}), flush=True)


def cpu_baseline(sample_n=160, timeout=600):
    """The CPU leg runs in a child process with a bounded OpenMP team: the reference calls LAPACK from
    inside its OpenMP loops and the OpenBLAS bundled with scipy aborts beyond 128 caller threads (this
    box has 256 hardware threads); OPENBLAS_NUM_THREADS=1 as in BASELINE.md."""
    import subprocess
    threads = min(os.cpu_count() or 1, 64)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(sample_n)], env=env,
                       capture_output=True, text=True, timeout=timeout)
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise RuntimeError(f"cpu baseline worker failed (rc {r.returncode}): {r.stderr[-300:]}")


METRIC = "Mvoxels/s detect+describe on 512^3 float32; 3D Gaussian achieved HBM GB/s vs roofline"


class RankFailure(Exception):
    """Some rank's job failed (its own error or a peer's); .errors = [(rank, text)]."""

    def __init__(self, errors):
        super().__init__("; ".join(f"rank {r}: {t}" for r, t in errors))
        self.errors = errors


PARITY_GOLDEN = os.path.join(ROOT, "tests", "golden", "bench_parity.json")
PREFLIGHT_DIMS = (64, 64, 1024)                     # the geometry of tests/test_gpu_slab.py::test_loopback_ranks_equal_single_gpu[8-...]
LIVE_CHECK_MAX_VOXELS = 1 << 25                     # volumes up to this size are also checked against a single-GPU run made on the spot


def kp_digest(kp):
    """(count, SHA-256) of a Keypoint_store: coordinates, scale, octave, level and the rotation of every keypoint, in list
    order -- the bytes tests/test_gpu_slab.py::test_config3_1024_cubed hashes."""
    import hashlib
    import numpy as np
    K = int(kp.slab.num)
    if K == 0:
        return 0, hashlib.sha256(b"").hexdigest()
    raw = np.ctypeslib.as_array(C.cast(kp.buf, C.POINTER(C.c_uint8)), shape=(K, C.sizeof(abi.Keypoint)))
    return K, hashlib.sha256(np.ascontiguousarray(raw[:, 72:112]).tobytes() + np.ascontiguousarray(raw[:, 0:36]).tobytes()).hexdigest()


def single_gpu_digest(L, dev, dims, nblobs):
    """kp_digest of the same volume detected whole on the calling thread's GPU (the product's single-GPU path)."""
    from sift3d_amd import slab as S
    nx, ny, nz = dims
    d_vol = dev.upload(synth.blobs(nx, ny, nz, nblobs, seed=0))
    s = S.make_params(L, bench_params())
    kp = abi.Keypoint_store()
    L.init_Keypoint_store(C.byref(kp))
    try:
        if L.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_double(1.0),
                                             C.c_double(1.0), C.c_double(1.0), C.byref(kp)) != 0:
            raise RuntimeError("single-GPU detect of the parity check failed")
        return kp_digest(kp)
    finally:
        L.cleanup_Keypoint_store(C.byref(kp))
        L.cleanup_SIFT3D(C.byref(s))
        dev.free(d_vol)


def expected_parity(dims):
    """The committed single-GPU result of this volume (default parameters, default blob count, seed 0), or None.  The file is
    written by tests/golden/make_bench_parity.py on a GPU box; S3D_BENCH_PARITY_GOLDEN points the tests at another one."""
    if bench_params() and not os.environ.get("S3D_BENCH_PARITY_GOLDEN"):
        return None                                  # (the committed results are those of the default parameters)
    try:
        g = json.load(open(os.environ.get("S3D_BENCH_PARITY_GOLDEN", PARITY_GOLDEN)))
    except (OSError, ValueError):
        return None
    return g.get("volumes", {}).get("x".join(str(d) for d in dims))


def parity_block(L, dev, dims, per_rank):
    """config.parity of an N > 1 line: what the ranks found against what ONE GPU finds in the same volume.  Every rank holds the
    gathered list (sift3d_amd_slab_gather, reference order): they must agree with each other, with the committed single-GPU
    result where there is one, and -- for volumes a single GPU detects in a moment -- with a single-GPU run made here."""
    K, sha = per_rank[0]["kp_total"], per_rank[0]["kp_sha256"]
    out = {"keypoints": K, "kp_sha256": sha, "expected": None, "kp_sha256_equals_single_gpu": None, "checked_against": []}
    ok = all(r["kp_total"] == K and r["kp_sha256"] == sha for r in per_rank)
    if not ok:
        out["ranks_disagree"] = [[r["kp_total"], r["kp_sha256"][:16]] for r in per_rank]
    exp = expected_parity(dims)
    if exp is not None:
        out["expected"] = int(exp["keypoints"])
        same = K == int(exp["keypoints"]) and sha == exp["kp_sha256"]
        out["kp_sha256_equals_single_gpu"] = bool(same)
        out["checked_against"].append(f"tests/golden/bench_parity.json ({exp.get('source', 'single-GPU run of this library')})")
        ok = ok and same
    if float(dims[0]) * dims[1] * dims[2] <= LIVE_CHECK_MAX_VOXELS:
        k1, sha1 = single_gpu_digest(L, dev, dims, per_rank[0]["nblobs"])
        same = K == k1 and sha == sha1
        out["expected"] = k1 if out["expected"] is None else out["expected"]
        out["kp_sha256_equals_single_gpu"] = bool(same and out["kp_sha256_equals_single_gpu"] is not False)
        out["checked_against"].append("a single-GPU detect of the same volume in this run (rank 0's GPU)")
        ok = ok and same
    out["ok"] = bool(ok)
    return out


def slab_job(L, dev, transport, dims, steps, warmup, sync_all, tag):
    """One rank's share of a Z-slab job on an nx x ny x nz volume through the C driver (include/sift3d_amd_slab.h):
    synthesise + upload my slab, `warmup` untimed steps, one split step, `steps` timed steps, then -- untimed -- the gathered
    keypoint list's digest (kp_digest; collective).  sync_all(): barrier over the ranks with this rank's stream drained.
    Returns this rank's measurements."""
    from sift3d_amd import slab as S
    nx, ny, nz = dims
    sl = S.Slab(L, transport, nx, ny, nz, params=bench_params())
    inf = sl.info()
    nblobs = synth.default_nblobs(nx, ny, nz)
    t0 = time.perf_counter()
    vol = synth.blobs(nx, ny, nz, nblobs, seed=0, z0=inf.z0, z1=inf.z1)
    d_vol = dev.upload(vol)
    del vol
    log(f"[{tag} rank {inf.rank}] slab z=[{inf.z0},{inf.z1}) of {nx}x{ny}x{nz} synthesised + uploaded in "
        f"{time.perf_counter() - t0:.1f} s; {inf.device_bytes / 2**30:.1f} GiB of HBM")

    def step():
        sl.detect(d_vol, on_device=True)
        sl.describe(to_host=False)

    try:
        for _ in range(warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        sl.detect(d_vol, on_device=True)
        t_detect = time.perf_counter() - t0
        t0 = time.perf_counter()
        sl.describe(to_host=False)
        t_describe = time.perf_counter() - t0
        split = sl.info()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync_all()
        elapsed = time.perf_counter() - t0
        inf = sl.info()
        kp_all, _ = sl.gather(with_desc=False)              # untimed: the whole volume's list, reference order, on every rank
        kp_total, kp_sha = kp_digest(kp_all)
        L.cleanup_Keypoint_store(C.byref(kp_all))
        out = {"elapsed": elapsed, "detect_ms": t_detect * 1e3, "describe_ms": t_describe * 1e3, "keypoints": int(inf.num_keypoints),
               "candidates": int(inf.num_candidates), "halo_bytes": float(inf.halo_bytes), "o_shard": int(inf.o_shard),
               "halo": int(inf.halo), "slices": int(inf.z1 - inf.z0), "device_GiB": inf.device_bytes / 2**30, "nblobs": nblobs,
               "comm_ms": float(split.comm_ms), "halo_wait_ms": float(split.halo_wait_ms), "kp_total": kp_total, "kp_sha256": kp_sha}
    finally:
        dev.free(d_vol)
        sl.close()
    return out


def slab_summary(per_rank, dims, steps, world, transport_name):
    nx, ny, nz = dims
    elapsed = max(r["elapsed"] for r in per_rank)
    r0 = per_rank[0]
    return elapsed, {
        "keypoints": sum(r["keypoints"] for r in per_rank), "extrema_candidates": sum(r["candidates"] for r in per_rank),
        "detect_ms": round(max(r["detect_ms"] for r in per_rank), 3), "describe_ms": round(max(r["describe_ms"] for r in per_rank), 3),
        "keypoints_per_rank": [r["keypoints"] for r in per_rank], "slices_per_rank": [r["slices"] for r in per_rank],
        # GPU time (HIP events) of the split step's detect, per rank: comm_ms = the compute stream inside halo exchanges /
        # all-reduces / the seed all-gather (transfer + waiting for the peers); halo_wait_ms = how long it then stood
        # waiting for the deferred halo planes (0 = they overlapped the pyramid kernels completely)
        "comm_ms_per_rank": [round(r["comm_ms"], 3) for r in per_rank],
        "halo_wait_ms_per_rank": [round(r["halo_wait_ms"], 3) for r in per_rank],
        "detect_ms_per_rank": [round(r["detect_ms"], 3) for r in per_rank],
        "describe_ms_per_rank": [round(r["describe_ms"], 3) for r in per_rank],
        "sharded_octaves": r0["o_shard"] + 1, "halo_planes": r0["halo"],
        "halo_MB_per_step_all_ranks": round(sum(r["halo_bytes"] for r in per_rank) / 1e6, 1),
        "HBM_GiB_per_rank": round(max(r["device_GiB"] for r in per_rank), 2),
        "Mvox_s": round(float(nx) * ny * nz * steps / elapsed / 1e6, 1), "ms_per_step": round(elapsed / steps * 1e3, 3),
        "parallelism": f"Z-slab x{world}, host C driver (csrc/host/s3d_host_slab.c), transport: {transport_name}; halos between "
                       f"Z-neighbours, max all-reduce for the scale / peak thresholds, all-gather of the coarse-octave seed"}


def bench_params():
    """Test aid: S3D_BENCH_PARAMS="sigma_n=0.8,sigma0=1.2" shrinks the descriptor windows so that the multi-rank path can
    be driven on volumes of a few dozen slices (the CPU tests); unset = default SIFT3D parameters."""
    e = os.environ.get("S3D_BENCH_PARAMS")
    if not e:
        return None
    return {k: float(v) for k, v in (kv.split("=") for kv in e.split(","))}


def slab_result(args, per_rank, extra, dims, world, tname, rccl, parity=None, preflight=None, fallback=None):
    elapsed, cfg = slab_summary(per_rank, dims, args.steps, world, tname)
    if parity is not None:
        cfg["parity"] = parity                   # keypoints of the N-rank run against one GPU's (parity_block)
    if preflight is not None:
        cfg["preflight"] = preflight             # the small job run over the same transport before the timed one
    if fallback is not None:
        cfg["transport_fallback"] = fallback     # the intended transport failed: why, and what this line was measured on
    nvox = float(dims[0]) * dims[1] * dims[2]
    pdesc = "default SIFT3D parameters" if not bench_params() else f"TEST parameters {bench_params()}"
    cfg["workload"] = (f"one {dims[0]}x{dims[1]}x{dims[2]} float32 blobs+noise volume ({per_rank[0]['nblobs']} blobs), unit voxels, "
                       f"{pdesc}, Z-slab sharded: {per_rank[0]['slices']} slices per GPU; detect + describe "
                       f"all keypoints, slabs and descriptors resident in HBM")
    if rccl is not None:
        cfg["rccl_ranks"], cfg["rccl_version"] = rccl       # what the communicator itself reports (ncclCommCount, ncclGetVersion)
    result = {"metric": METRIC, "value": round(nvox * args.steps / elapsed / 1e6, 2), "unit": "Mvox/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
              "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
              "dtype": "f32", "data": "synthetic", "config": cfg}
    if extra is not None:
        edims, ekey = extra_job(args, world)
        _, c3 = slab_summary(extra, edims, 2, world, tname)
        c3["workload"] = (f"one {edims[0]}x{edims[1]}x{edims[2]} volume, Z-slabs of {extra[0]['slices']} slices on {world} GPUs "
                          f"({'BASELINE configs[3], strong scaling' if args.weak else 'weak scaling: 512 slices per GPU'}); 2 steps, untimed extra")
        result["config"][ekey] = c3
    return result


def timed_dims(args, world):
    """The timed N > 1 workload: BASELINE configs[3] (one --strong-size^3 volume, strong scaling) unless --weak."""
    n = args.size
    return (n, n, n * world) if args.weak else (args.strong_size,) * 3


def extra_job(args, world):
    """The other decomposition, run for two untimed steps after the timed region."""
    n = args.size
    return ((args.strong_size,) * 3, "strong_1024") if args.weak else ((n, n, n * world), "weak_512xN")


def dry_plan(L, dims, world, params=None):
    """Plans of all ranks of a `world`-way Z-slab job on a dims volume (sift3d_amd_slab_plan: no device).  Raises
    RuntimeError with the library's message where the real call would refuse the decomposition."""
    from sift3d_amd import slab as S
    s = S.make_params(L, params)
    L.sift3d_amd_slab_plan.argtypes = [C.POINTER(abi.SIFT3D)] + [C.c_int] * 5 + [C.c_double] * 3 + [C.POINTER(S.SlabInfo)]
    L.sift3d_amd_slab_last_error.restype = C.c_char_p
    out = []
    try:
        for r in range(world):
            inf = S.SlabInfo()
            if L.sift3d_amd_slab_plan(C.byref(s), world, r, dims[0], dims[1], dims[2], 1.0, 1.0, 1.0, C.byref(inf)) != 0:
                raise RuntimeError((L.sift3d_amd_slab_last_error() or b"").decode())
            out.append(inf)
    finally:
        L.cleanup_SIFT3D(C.byref(s))
    plane = dims[0] * dims[1] * 4
    # Link traffic of one detect and what it has to hide under.  A rank sends `plan_send_lo/hi_bytes` to its two Z-neighbours
    # and receives as much; the two directions of an xGMI link run concurrently, the two neighbours sit on different links,
    # so a rank's exposed transfer time is max(lo, hi) / link rate -- at the 153 GB/s of MI355X_MICROARCH.md and at a
    # pessimistic 60 GB/s -- plus the seed all-gather of the first replicated octave.  The compute beside it: the single-GPU
    # detect of the same volume (profiles/r04_run5_bench_single_1024.json: 48.7 ms at 1024^3, 46 ps per voxel; 6.5 ms at
    # 512^3) divided by the ranks, i.e. what the halos have to overlap with for linear scaling.  Overlap is by
    # construction partial: a level's Gaussian needs only the next filter's reach (8 planes) at once, the window halo of
    # the keypoint levels (26 / 32 / 39 planes at the default parameters) travels on a second stream and communicator and
    # is needed only by orientation + description.
    ps_per_voxel = 46.0
    detect_ms_single = ps_per_voxel * 1e-12 * dims[0] * dims[1] * dims[2] * 1e3
    links = []
    for i in out:
        worst = max(i.plan_send_lo_bytes, i.plan_send_hi_bytes)
        links.append({"rank": int(i.rank), "send_lo_MB": round(i.plan_send_lo_bytes / 1e6, 1), "send_hi_MB": round(i.plan_send_hi_bytes / 1e6, 1),
                      "halo_bytes_per_link": int(worst), "seed_gather_MB": round(i.plan_seed_gather_bytes / 1e6, 1),
                      "link_ms_at_153GBs": round((worst + i.plan_seed_gather_bytes / max(1, world - 1)) / 153e9 * 1e3, 2),
                      "link_ms_at_60GBs": round((worst + i.plan_seed_gather_bytes / max(1, world - 1)) / 60e9 * 1e3, 2)})
    return {"volume": list(dims), "ranks": world,
            "halo_planes_per_level_octave0": [int(v) for v in out[0].plan_halo_planes[:int(out[0].num_levels)]],
            "per_rank_links": links,
            "halo_MB_per_detect_all_ranks": round(sum(i.plan_send_lo_bytes + i.plan_send_hi_bytes for i in out) / 1e6, 1),
            "time_model": {"single_gpu_detect_ms": round(detect_ms_single, 2), "ideal_detect_ms_per_rank": round(detect_ms_single / world, 2),
                           "worst_link_ms_at_153GBs": max(l["link_ms_at_153GBs"] for l in links),
                           "worst_link_ms_at_60GBs": max(l["link_ms_at_60GBs"] for l in links),
                           "note": "near-linear needs the link time hidden under ideal_detect_ms_per_rank; describe (2x detect, "
                                   "no traffic) dilutes what stays exposed"},
            "slices_per_rank": [int(i.z1 - i.z0) for i in out], "z_bounds": [int(out[0].z0)] + [int(i.z1) for i in out],
            "sharded_octaves": int(out[0].o_shard) + 1, "octaves": int(out[0].num_octaves), "halo_planes": int(out[0].halo),
            # planes a rank holds per sharded octave-0 level: its slices plus the halo on each interior side
            "halo_overhead_octave0": round(2.0 * out[0].halo / max(1, min(int(i.z1 - i.z0) for i in out)), 3),
            "halo_MiB_per_level_and_side_octave0": round(out[0].halo * plane / 2**20, 1),
            "HBM_GiB_per_rank": [round(i.device_bytes / 2**30, 2) for i in out],
            "fits_288_GB": all(i.device_bytes < 0.9 * 288e9 for i in out)}


def dry_run(args):
    """--dry: validate the decomposition of the N > 1 workloads without a device.  One JSON line."""
    from sift3d_amd import slab as S
    L = S.bind(sift3d_amd.cdll())
    worlds = [args.gpus] if args.gpus > 1 else [2, 4, 8]
    out = {"dry": True, "metric": METRIC, "plans": {}}
    for N in worlds:
        entry = {}
        for key, dims in (("strong", (args.strong_size,) * 3), ("weak", (args.size, args.size, args.size * N))):
            try:
                entry[key] = dry_plan(L, dims, N, bench_params())
            except RuntimeError as e:
                entry[key] = {"volume": list(dims), "ranks": N, "refused": str(e)}
        entry["timed"] = "weak" if args.weak else "strong"
        entry["command"] = (f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {N} --master-addr 127.0.0.1 --master-port P "
                            f"bench.py --gpus {N} --steps K --warmup W" + (" --weak" if args.weak else ""))
        out["plans"][str(N)] = entry
    # what a decomposition that cannot work says (slabs thinner than a descriptor window)
    try:
        dry_plan(L, (64, 64, 64), 8, bench_params())
        out["refusal_example"] = None
    except RuntimeError as e:
        out["refusal_example"] = {"volume": [64, 64, 64], "ranks": 8, "refused": str(e)}
    print(json.dumps(out), flush=True)


def arm_inject(L, stage):
    """Test aid (TESTING / emulator builds of the library only): S3D_BENCH_TEST_INJECT="stage:rank:where" arms the driver's
    one-shot failure hook (sift3d_amd_slab_test_inject) just before `stage` (preflight | timed)."""
    e = os.environ.get("S3D_BENCH_TEST_INJECT")
    if not e or not hasattr(L, "sift3d_amd_slab_test_inject"):
        return
    st, r, w = e.split(":")
    if st == stage:
        L.sift3d_amd_slab_test_inject.argtypes = [C.c_int, C.c_int]
        L.sift3d_amd_slab_test_inject.restype = None
        L.sift3d_amd_slab_test_inject(int(r), int(w))


def preflight_dims(world):
    """The pre-flight volume: PREFLIGHT_DIMS unless S3D_BENCH_PREFLIGHT says "nx,ny,nz" (the CPU tests) or "0" (none)."""
    e = os.environ.get("S3D_BENCH_PREFLIGHT")
    if e is None:
        return PREFLIGHT_DIMS
    if e.strip() in ("", "0"):
        return None
    return tuple(int(v) for v in e.split(","))


def run_inprocess(args, dev):
    """N > 1 launched plainly: this ONE process drives the N GPUs -- one host thread per rank (thread r: GPU r), RCCL
    communicators from ncclCommInitAll (sift3d_amd_rccl_create_all), the C Z-slab driver on device-resident slabs.  The
    same transport and driver a relinked caller gets from sift3d_amd_set_num_gpus(&sift3d, N, 0); no launcher, no
    torch.distributed, one RCCL instance.

    Fail-soft: a pre-flight job on a small volume runs over the RCCL transport first and is compared with one GPU's result;
    if that, or later the timed job, fails on RCCL, the ranks run again over the library's loop-back transport (peer copies
    behind a host barrier) and the line says so, with the error text -- a labelled number instead of exit status 3."""
    import threading
    from sift3d_amd import slab as S
    L = sift3d_amd.cdll()
    N = args.gpus
    ndev = dev.device_count()
    if ndev < N:
        log(f"bench.py: --gpus {N} requested, {ndev} GPU(s) visible: refusing to benchmark fewer GPUs than asked for")
        raise SystemExit(2)
    n = args.size
    dims = timed_dims(args, N)
    RCCL_NAME = "RCCL, one process (ncclCommInitAll; ncclSend/ncclRecv, ncclAllReduce, ncclAllGather; csrc/s3d_rccl.hip)"
    LOOP_NAME = "in-process loop-back over hipMemcpy between the GPUs -- RCCL did NOT initialise"
    fallback = None
    try:
        tr = S.rccl_all_transports(L, N)
        rccl = S.rccl_info(L, tr[0])
        if rccl[0] != N:
            log(f"bench.py: the RCCL communicator has {rccl[0]} ranks, not {N}")
            raise SystemExit(2)
        tname = RCCL_NAME
    except RuntimeError as e:
        # N GPUs are there but RCCL is not: the N ranks still run, halos as peer copies behind a host barrier (the
        # library's loop-back transport).  The line says so -- it is NOT the RCCL number.
        log(f"bench.py: RCCL unavailable ({e}); falling back to the in-process loop-back transport (hipMemcpy between the GPUs)")
        tr, rccl, tname = S.loopback_transports(L, N), None, LOOP_NAME

    def run(tr, what, steps, warmup, tag):
        bar = threading.Barrier(N)
        out, err = [None] * N, [None] * N

        def body(r):
            try:
                dev.check(dev.L.s3d_rt_set_device(r), "set_device")      # per thread

                def sync_all():
                    dev.sync()
                    bar.wait()
                out[r] = slab_job(L, dev, tr[r], what, steps, warmup, sync_all, tag)
            except BaseException as e:                                 # noqa: BLE001
                err[r] = e
                bar.abort()                                            # host barrier ...
                for q in range(N):                                     # ... and the ranks' transports: nobody waits for ever
                    if tr[q].abort:
                        tr[q].abort(tr[q].self)

        th = [threading.Thread(target=body, args=(r,)) for r in range(N)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        bad = [(r, f"{type(e_).__name__}: {e_}") for r, e_ in enumerate(err)
               if e_ is not None and not isinstance(e_, threading.BrokenBarrierError)]
        for r, text in bad:
            log(f"bench.py: rank {r} failed: {text}")
        if any(e_ is not None for e_ in err):
            raise RankFailure(bad or [(-1, "a rank left the barrier")])
        return out

    def to_loopback(why):
        """the RCCL transports are given up (aborted by the failure, destroyed here); the ranks go on over peer copies"""
        nonlocal tr, rccl, tname, fallback
        log(f"bench.py: {why}; running the job again over the in-process loop-back transport (hipMemcpy between the GPUs)")
        for r in range(N):
            tr[r].destroy(tr[r].self)
        fallback = {"intended": RCCL_NAME, "error": why, "measured_on": "in-process loop-back over hipMemcpy between the GPUs"}
        tr, rccl, tname = S.loopback_transports(L, N), None, "in-process loop-back over hipMemcpy between the GPUs -- RCCL FAILED, see transport_fallback"

    preflight = None
    pdims = preflight_dims(N)
    if pdims is not None:
        dev.check(dev.L.s3d_rt_set_device(0), "set_device")
        t0 = time.perf_counter()
        arm_inject(L, "preflight")
        try:
            pr = run(tr, pdims, 1, 0, "preflight")
            pb = parity_block(L, dev, pdims, pr)
            preflight = {"volume": list(pdims), "ok": pb["ok"], "keypoints": pb["keypoints"], "expected": pb["expected"],
                         "seconds": round(time.perf_counter() - t0, 2), "transport": "RCCL" if rccl is not None else "loop-back"}
            if not pb["ok"]:
                if rccl is None:
                    log(f"bench.py: the pre-flight job disagrees with the single-GPU result on the loop-back transport: {pb}")
                    raise SystemExit(4)
                to_loopback(f"pre-flight on {pdims[0]}x{pdims[1]}x{pdims[2]} over RCCL: {pb['keypoints']} keypoints, single GPU {pb['expected']}")
        except RankFailure as e:
            if rccl is None:
                log(f"bench.py: the pre-flight job failed on the loop-back transport: {e}")
                raise SystemExit(3)
            preflight = {"volume": list(pdims), "ok": False, "error": str(e), "transport": "RCCL"}
            to_loopback(f"pre-flight over RCCL failed: {e}")

    def timed():
        per_rank = run(tr, dims, args.steps, args.warmup, "timed")
        extra = None
        if not args.no_match and N in (2, 4, 8, 16) and n >= 512:
            extra = run(tr, extra_job(args, N)[0], 2, 1, extra_job(args, N)[1])
        return per_rank, extra

    arm_inject(L, "timed")
    try:
        per_rank, extra = timed()
    except RankFailure as e:
        if rccl is None:
            log(f"bench.py: the job failed on the loop-back transport: {e}")
            raise SystemExit(3)
        to_loopback(f"the timed job failed over RCCL: {e}")
        try:
            per_rank, extra = timed()
        except RankFailure as e2:
            log(f"bench.py: the job failed on the loop-back transport as well: {e2}")
            raise SystemExit(3)
    dev.check(dev.L.s3d_rt_set_device(0), "set_device")
    parity = parity_block(L, dev, dims, per_rank)
    result = slab_result(args, per_rank, extra, dims, N, tname, rccl, parity, preflight, fallback)
    if not args.no_roofline:
        add_roofline(result, dev, n)
    print(json.dumps(result), flush=True)
    for r in range(N):
        tr[r].destroy(tr[r].self)
    if not parity["ok"]:
        log(f"bench.py: PARITY FAILURE: the {N}-rank keypoint list differs from the single-GPU one: {parity}")
        raise SystemExit(4)


class StoreSync:
    """Barriers of the torchrun ranks that a failed rank can break.  A gloo barrier cannot be left: a rank whose job has failed
    would have to keep calling the same collectives as its peers, or everybody hangs.  These barriers are counters in the
    rendezvous store (c10d TCPStore): arrive = add 1, wait = poll the counter AND the failure key of the current attempt; a
    rank that fails sets the key, so that its peers leave their barrier with PeerFailure instead of waiting for it."""

    class PeerFailure(Exception):
        pass

    def __init__(self, dist, rank, world):
        self.store = dist.distributed_c10d._get_default_store()
        self.rank, self.world, self.attempt, self.n = rank, world, 0, 0

    def next_attempt(self):
        self.attempt += 1
        self.n = 0

    def fail(self, text):
        self.store.set(f"s3d/fail/{self.attempt}", f"rank {self.rank}: {text}")

    def failure(self):
        k = f"s3d/fail/{self.attempt}"
        return self.store.get(k).decode() if self.store.check([k]) else None

    def barrier(self, breakable=True, timeout_s=1800.0):
        """breakable: the n-th barrier of the attempt (the ranks count alike while all goes well); not breakable: THE meeting point
        of an attempt's survivors -- ranks get there from different barriers, so its key carries no count"""
        if breakable:
            self.n += 1
        key = f"s3d/bar/{self.attempt}/{self.n}" if breakable else f"s3d/rdv/{self.attempt}"
        self.store.add(key, 1)
        t0 = time.perf_counter()
        while self.store.add(key, 0) < self.world:
            if breakable:
                f = self.failure()
                if f is not None:
                    raise StoreSync.PeerFailure(f)
            waited = time.perf_counter() - t0
            if waited > timeout_s:
                raise StoreSync.PeerFailure(f"barrier {key}: a rank did not arrive within {timeout_s:.0f} s")
            if waited > 0.005:
                time.sleep(0.0002)                   # (a long wait -- peers still synthesising -- need not hammer the store)


def run_slab(args, dist, dev, rank, local_rank, world, full_sync):
    """N > 1 under torchrun: one rank per GPU, the C Z-slab driver over RCCL (ncclSend/ncclRecv halos).  Timed
    workload: BASELINE configs[3] -- ONE --strong-size^3 volume (1024^3), nz/N slices per GPU, strong scaling (--weak: one
    n x n x (n*N) volume, n slices per GPU).  After the timed region, unless --no-match: the other of the two for two
    untimed steps (config.weak_512xN / config.strong_1024).

    Fail-soft (as run_inprocess): pre-flight job + comparison with one GPU before the timed job; a failure on RCCL at any
    point after initialisation -- a rank's own error, or a peer's, noticed through the breakable barriers of StoreSync or
    through SIFT3D_SLAB_TIMEOUT_S in the C driver -- makes every rank give up its RCCL transport and run the job again over
    torch.distributed callbacks (gloo, staged through the host); the line is labelled and carries the error text."""
    import torch
    from sift3d_amd import slab as S
    L = sift3d_amd.cdll()
    os.environ.setdefault("SIFT3D_SLAB_TIMEOUT_S", "60")       # a rank waits this long for a dead peer's halo (and for RCCL's first connection set-up) before it fails too
    same_gpu = bool(os.environ.get("S3D_BENCH_SAME_GPU"))
    on_gpu = torch.cuda.is_available()                          # False: the CPU tests (emulator build of the library, gloo only)
    gloo_dev = f"cuda:{local_rank}" if on_gpu else None
    RCCL_NAME = "RCCL, one process per GPU (ncclSend/ncclRecv, ncclAllReduce, ncclAllGather; csrc/s3d_rccl.hip; id shipped over gloo)"
    GLOO_NAME = "torch.distributed callbacks (gloo, staged through the host)"
    tname, tr, keep = RCCL_NAME, None, None
    ok = 0.0
    # S3D_BENCH_FAKE_RCCL (CPU tests only): the first transport is a gloo one that the protocol below treats as "the RCCL
    # transport" -- there is no multi-process stand-in for librccl, and the give-up-and-go-on-over-gloo path wants a test
    fake_rccl = bool(os.environ.get("S3D_BENCH_FAKE_RCCL")) and not on_gpu
    if not same_gpu and not fake_rccl:
        try:
            tr = S.rccl_transport(L, dist, rank, world)
            ok = 1.0
        except Exception as e:              # the decision to fall back must be collective
            log(f"[rank {rank}] RCCL transport unavailable: {e}")
    flag = torch.tensor([ok])                                # the process group is gloo: host tensors
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    rccl, fallback = None, None
    if fake_rccl:
        keep = S.DistTransport(L, dist, device=gloo_dev, stage_via_host=on_gpu)
        tr, rccl = keep.struct, (world, 0)
    elif float(flag.item()) == 0.0:
        if tr is not None and tr.destroy:
            tr.destroy(tr.self)
        keep = S.DistTransport(L, dist, device=gloo_dev, stage_via_host=on_gpu)
        tr = keep.struct
        tname = GLOO_NAME + " -- RCCL did NOT initialise"
    else:
        rccl = S.rccl_info(L, tr)
        if rccl[0] != world:
            log(f"[rank {rank}] the RCCL communicator has {rccl[0]} ranks, not {world}")
            raise SystemExit(2)
    sync = StoreSync(dist, rank, world)

    def sync_all():
        if on_gpu:
            torch.cuda.synchronize()                           # this rank's GPU has drained (every stream) ...
        dev.check(dev.L.s3d_rt_sync(None))
        sync.barrier()                                         # ... and so have the others'

    def job(what, steps, warmup, tag):
        """This rank's share; RankFailure if it or any peer fails.  The results of all ranks (rank order) otherwise."""
        try:
            mine = slab_job(L, dev, tr, what, steps, warmup, sync_all, tag)
            sync.barrier()                                      # everybody got through: the gloo gather below is matched
        except StoreSync.PeerFailure as e:
            if tr is not None and tr.abort:
                tr.abort(tr.self)
            raise RankFailure([(-1, str(e))])
        except BaseException as e:                              # noqa: BLE001
            text = f"{type(e).__name__}: {e}"
            log(f"[rank {rank}] {tag} job failed: {text}")
            sync.fail(f"{tag}: {text}")
            if tr is not None and tr.abort:
                tr.abort(tr.self)
            raise RankFailure([(rank, text)])
        box = [None] * world
        dist.all_gather_object(box, mine)
        return box

    def to_gloo(why):
        """collective: every rank has left its job with RankFailure; they meet here, drop RCCL and go on over gloo"""
        nonlocal tr, keep, rccl, tname, fallback
        sync.barrier(breakable=False, timeout_s=300.0)
        why_all = sync.failure() or why
        sync.next_attempt()
        log(f"[rank {rank}] {why_all}; running the job again over torch.distributed callbacks (gloo)")
        if keep is None and tr is not None and tr.destroy:
            tr.destroy(tr.self)
        fallback = {"intended": RCCL_NAME, "error": why_all, "measured_on": GLOO_NAME}
        keep = S.DistTransport(L, dist, device=gloo_dev, stage_via_host=on_gpu)
        tr, rccl, tname = keep.struct, None, GLOO_NAME + " -- RCCL FAILED, see transport_fallback"

    def give_up(text):
        log(f"[rank {rank}] {text}")
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)

    n = args.size
    dims = timed_dims(args, world)
    preflight = None
    pdims = preflight_dims(world)
    if pdims is not None:
        t0 = time.perf_counter()
        arm_inject(L, "preflight")
        try:
            pr = job(pdims, 1, 0, "preflight")
            box = [parity_block(L, dev, pdims, pr) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            pb = box[0]
            preflight = {"volume": list(pdims), "ok": pb["ok"], "keypoints": pb["keypoints"], "expected": pb["expected"],
                         "seconds": round(time.perf_counter() - t0, 2), "transport": "RCCL" if rccl is not None else "gloo"}
            if not pb["ok"]:
                if rccl is None:
                    give_up(f"the pre-flight job disagrees with the single-GPU result on the gloo transport: {pb}")
                sync.fail(f"pre-flight on {pdims[0]}x{pdims[1]}x{pdims[2]} over RCCL: {pb['keypoints']} keypoints, single GPU {pb['expected']}")
                to_gloo("pre-flight mismatch over RCCL")
        except RankFailure as e:
            if rccl is None:
                give_up(f"the pre-flight job failed on the gloo transport: {e}")
            preflight = {"volume": list(pdims), "ok": False, "error": str(e), "transport": "RCCL"}
            to_gloo(f"pre-flight over RCCL failed: {e}")

    def timed():
        per_rank = job(dims, args.steps, args.warmup, "timed")
        extra = None
        if not args.no_match and world in (2, 4, 8, 16) and n >= 512:
            extra = job(extra_job(args, world)[0], 2, 1, extra_job(args, world)[1])
        return per_rank, extra

    arm_inject(L, "timed")
    try:
        per_rank, extra = timed()
    except RankFailure as e:
        if rccl is None:
            give_up(f"the job failed on the gloo transport: {e}")
        to_gloo(f"the timed job failed over RCCL: {e}")
        try:
            per_rank, extra = timed()
        except RankFailure as e2:
            give_up(f"the job failed on the gloo transport as well: {e2}")
    status = 0
    if rank == 0:
        parity = parity_block(L, dev, dims, per_rank)
        result = slab_result(args, per_rank, extra, dims, world, tname, rccl, parity, preflight, fallback)
        if not args.no_roofline:
            add_roofline(result, dev, n)
        print(json.dumps(result), flush=True)
        if not parity["ok"]:
            log(f"bench.py: PARITY FAILURE: the {world}-rank keypoint list differs from the single-GPU one: {parity}")
            status = 4
    box = [status]
    dist.broadcast_object_list(box, src=0)
    if keep is None and tr is not None and tr.destroy:
        tr.destroy(tr.self)
    dist.destroy_process_group()
    if box[0]:
        raise SystemExit(box[0])


def run_loopback(args, dev):
    """Diagnostic on ONE GPU: --loopback R runs R ranks as host threads that share the device (the library's
    loop-back transport) -- the multi-rank code path with its slab geometry, halos and orderings, minus xGMI.  What it
    measures is the decomposition's overhead (halo recomputation, replicated octaves, synchronous hand-offs), not
    scaling."""
    import threading
    from sift3d_amd import slab as S
    L = sift3d_amd.cdll()
    R = args.loopback
    n = args.size
    dims = timed_dims(args, R)
    tr = S.loopback_transports(L, R)
    bar = threading.Barrier(R)

    def body(r):
        def sync_all():
            dev.sync()
            bar.wait()
        return slab_job(L, dev, tr[r], dims, args.steps, args.warmup, sync_all, "loopback")

    per_rank = S.run_ranks(R, body)
    for r in range(R):
        tr[r].destroy(tr[r].self)
    elapsed, cfg = slab_summary(per_rank, dims, args.steps, R, "in-process loop-back, all ranks on one GPU")
    cfg["workload"] = f"one {dims[0]}x{dims[1]}x{dims[2]} volume in {R} Z-slabs on ONE GPU (diagnostic, not a scaling number)"
    cfg["parity"] = parity_block(L, dev, dims, per_rank)          # the R ranks' gathered list against one GPU's, as for --gpus N
    nvox = float(dims[0]) * dims[1] * dims[2]
    print(json.dumps({"metric": METRIC, "value": round(nvox * args.steps / elapsed / 1e6, 2), "unit": "Mvox/s", "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                      "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
                      "dtype": "f32", "data": "synthetic", "config": cfg}), flush=True)
    if not cfg["parity"]["ok"]:
        log(f"bench.py: PARITY FAILURE: the {R}-rank keypoint list differs from the single-GPU one: {cfg['parity']}")
        raise SystemExit(4)


def add_roofline(result, dev, n):
    sig = [0.538701, 0.973294, 1.22627, 1.54501, 1.94659, 2.45255]      # default bank: widths 5..17
    apps = gauss_roofline(dev, 512 if n >= 512 else n, sig)
    worst = max(apps, key=lambda a: a["xy_ms"])                       # widest filter = slowest fused kernel
    nv = float(512 if n >= 512 else n) ** 3
    ach = GAUSS_XY_BYTES_PER_VOXEL * nv / (worst["xy_ms"] * 1e-3) / 1e9
    # HBM bytes per launch of that kernel from PMC passes (scripts/pmc_hbm.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs,
    # FETCH_SIZE doubled per the gfx950 correction).  A rocprofv3 counter pass cannot run inside this process, so the figure
    # is read from profiles/pmc_gauss.json -- and used ONLY if that file was measured on these very kernels: the SHA-256 over
    # the MACHINE CODE of the k_gauss_xy / k_gauss_z functions inside the library this process has loaded
    # (sift3d_amd/codeobj.py; a comment edit or a change elsewhere in the file does not move it) at 512^3; otherwise traffic is
    # null.  traffic_source says which run and commit measured it.
    traffic, traffic_source = None, None
    try:
        from sift3d_amd import codeobj
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_gauss.json")))
        isa = codeobj.kernel_isa_sha256(os.path.join(ROOT, "sift3d_amd", "lib", "libsift3d_amd.so"), codeobj.GAUSS_KERNELS)
        if int(nv) == int(pmc["voxels"]) and isa and pmc.get("gauss_kernels_isa_sha256") == isa:
            hw_ = worst['width'] // 2                       # the aligned instantiation: "k_gauss_xy<8>" or "k_gauss_xy<8, false>"
            key = [k for k in pmc["kernels"] if k.replace(" ", "") in (f"k_gauss_xy<{hw_}>", f"k_gauss_xy<{hw_},false>")][0]
            traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
            traffic_source = (f"profiles/pmc_gauss.json: run {pmc.get('run')}, commit {pmc.get('commit')}, the same machine code "
                              f"of k_gauss_xy / k_gauss_z (sha256 {isa[:16]}...)")
        else:
            traffic_source = (f"profiles/pmc_gauss.json was measured on other machine code of the fused Gaussian kernels "
                              f"({str(pmc.get('gauss_kernels_isa_sha256'))[:16]}... vs loaded {isa[:16]}...): not used")
    except Exception as e:                                # noqa: BLE001
        traffic, traffic_source = None, f"profiles/pmc_gauss.json unusable: {e}"
    # What the kernel physically does, next to the algorithmic figure: the PMC bytes of one launch over the same
    # HIP-event time.  `frac` credits the fusion (two algorithmic passes for one read and one write of the volume);
    # `physical_frac` is the DRAM-side rate against the same 8 TB/s -- the guide's float4-copy ceiling is 6.29 TB/s.
    phys = traffic / (worst["xy_ms"] * 1e-3) / 1e9 if traffic else None
    # `frac`: one whole filter application (x, y and z passes: SURVEY 8(d)'s 24 algorithmic B/voxel) of the slowest width over
    # the HIP-event time of its two kernels -- the figure north_star's ">= 70 % of the HBM roofline" is about.  Beside it:
    # kernel_frac (the fused X+Y kernel alone on its 16 algorithmic B/voxel: two passes for one read and one write, so it can
    # exceed what a copy could) and physical_frac (the PMC bytes of that kernel over the same time: what the DRAM side
    # actually moved, against 8 TB/s; the guide's float4-copy ceiling is 6.29 TB/s).
    slow = min(apps, key=lambda a: a["app_GBs"])
    result["roofline"] = {"bound": "hbm", "achieved": slow["app_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(slow["app_GBs"] / HBM_PEAK_GBS, 4),
                          "basis": f"one Gaussian application, width {slow['width']} (the slowest of the default bank): 24 algorithmic "
                                   f"B/voxel x {int(nv)} voxels over xy {slow['xy_ms']} ms + z {slow['z_ms']} ms",
                          "kernel_GBs": round(ach, 1), "kernel_frac": round(ach / HBM_PEAK_GBS, 4),
                          "traffic": traffic, "traffic_source": traffic_source,
                          "physical_GBs": round(phys, 1) if phys else None,
                          "physical_frac": round(phys / HBM_PEAK_GBS, 4) if phys else None,
                          "physical_frac_of_copy_ceiling": round(phys / 6290.0, 4) if phys else None,
                          "time_source": f"HIP events on the launch stream inside this run (s3d_k_gauss_set_events), mean of 5 launches: "
                                         f"k_gauss_xy<{worst['width'] // 2}> {worst['xy_ms']} ms; the rocprofv3 --kernel-trace --stats average "
                                         f"of the same kernel is kept in profiles/ (per round: rNN_*_kernel_stats.md) and was 8-9 % shorter in "
                                         f"round 3 (281.8 vs 307.3 us: the events bracket launch gaps as well)",
                          "kernel": f"k_gauss_xy<{worst['width'] // 2}> (fused X+Y pass, width {worst['width']}): "
                                    f"16 algorithmic B/voxel x {int(nv)} voxels per launch"}
    result["config"]["gauss_apps"] = apps


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-baseline-worker":
        _cpu_baseline_worker(int(sys.argv[2]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512, help="volume edge (default 512 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="skip the untimed extras after the timed steps (matcher, descriptor-kernel statistics, host-buffer API, dense 256^3, anisotropic and two-volume configurations)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1 (or --loopback): time the weak-scaling volume (n x n x n*N, n slices per GPU) instead of BASELINE "
                         "configs[3] (one --strong-size^3 volume in nz/N-slice slabs, the default)")
    ap.add_argument("--strong", action="store_true", help="accepted for compatibility: strong scaling is the default")
    ap.add_argument("--strong-size", type=int, default=1024, help="edge of the strong-scaling volume (1024 = BASELINE configs[3])")
    ap.add_argument("--dry", action="store_true",
                    help="no device: print the Z-slab plan (partition, halo, sharded octaves, HBM per rank) of the N > 1 jobs for "
                         "N = 2, 4, 8 (or --gpus N), and what a refused decomposition says")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time the reference on BASELINE configs[1] itself (512^3: ~6.5 minutes on 64 cores) instead of quoting "
                         "profiles/r03_cpu_baseline_512.json")
    ap.add_argument("--loopback", type=int, default=0,
                    help="diagnostic on one GPU: R Z-slab ranks as host threads sharing the device (loop-back transport)")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: one independent volume per rank instead of the Z-slab decomposition")
    args = ap.parse_args()

    if args.dry:
        dry_run(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "WORLD_SIZE" in os.environ                       # under torchrun: one process per rank
    if launched and world != args.gpus:
        log(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a number for a different GPU count")
        raise SystemExit(2)
    dist = None
    if world > 1 or args.gpus > 1:
        # RCCL's own warnings on stderr: a first contact with a fabric that fails must be diagnosable from the tail of the
        # run's output (the library adds ncclGetLastError to its error texts: csrc/s3d_rccl.hip nccl_failed)
        os.environ.setdefault("NCCL_DEBUG", "WARN")
    if world > 1 or os.environ.get("S3D_BENCH_FORCE_SLAB"):     # FORCE_SLAB: the N > 1 code path with a world of one (1-GPU boxes)
        import torch
        import torch.distributed as dist
        # gloo for the bootstrap (a 128-byte id, a flag, the barriers): the process then holds ONE RCCL, the one the
        # transport opens -- not PyTorch's bundled copy beside it
        if os.environ.get("S3D_BENCH_SAME_GPU"):
            local_rank = 0                                       # debugging aid for a 1-GPU box: all ranks share GPU 0
        if torch.cuda.is_available():                            # (not in the CPU tests: emulator build of the library)
            torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo")

    lib = sift3d_amd.load()
    dev = sift3d_amd.load_device()
    if dev.device_count() < 1:
        raise SystemExit("bench.py: no HIP device -- the HIP path has no CPU fallback")
    dev.check(dev.L.s3d_rt_set_device(local_rank), "set_device")

    def full_sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        dev.check(dev.L.s3d_rt_sync(None))

    n = args.size
    if world == 1 and args.loopback > 1:
        run_loopback(args, dev)
        return
    if not launched and args.gpus > 1 and not os.environ.get("S3D_BENCH_FORCE_SLAB"):
        if args.replicas:
            log("bench.py: --replicas needs one process per GPU (torchrun)")
            raise SystemExit(2)
        run_inprocess(args, dev)
        return
    if (world > 1 or (dist is not None and os.environ.get("S3D_BENCH_FORCE_SLAB"))) and not args.replicas:
        run_slab(args, dist, dev, rank, local_rank, world, full_sync)
        return
    nblobs = synth.default_nblobs(n, n, n)            # 128 000 at 512^3
    t0 = time.perf_counter()
    vol = synth.blobs(n, n, n, nblobs, seed=rank)
    log(f"[rank {rank}] synthesised {n}^3 ({nblobs} blobs) in {time.perf_counter() - t0:.1f} s")
    d_vol = dev.upload(vol)

    s = abi.SIFT3D()
    assert lib.sift.init_SIFT3D(C.byref(s)) == 0
    kp = abi.Keypoint_store()
    lib.sift.init_Keypoint_store(C.byref(kp))
    d_desc = C.c_void_p()

    def step():
        rc = lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
        if rc != 0:
            raise SystemExit("detect failed: " + lib.sift.sift3d_amd_last_error().decode())
        if kp.slab.num:
            rc = lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d_desc))
            if rc != 0:
                raise SystemExit("describe failed: " + lib.sift.sift3d_amd_last_error().decode())

    for _ in range(args.warmup):
        step()
    # split timing of three extra untimed steps (detect vs describe; the smaller of each: a single shot picks up host jitter), for the record
    t_detect = t_describe = float("inf")
    for _ in range(3):
        full_sync()
        t0 = time.perf_counter()
        lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.0, C.byref(kp))
        dev.sync()
        t_detect = min(t_detect, time.perf_counter() - t0)
        t0 = time.perf_counter()
        if kp.slab.num:
            lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(s), C.byref(kp), C.byref(d_desc))
        dev.sync()
        t_describe = min(t_describe, time.perf_counter() - t0)

    full_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    full_sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    K = int(kp.slab.num)
    ncand = int(lib.sift.sift3d_amd_last_num_candidates(C.byref(s)))

    result = None
    if rank == 0:
        nvox = float(n) ** 3
        value = world * nvox * args.steps / elapsed / 1e6
        result = {
            "metric": METRIC,
            "value": round(value, 2), "unit": "Mvox/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n}^3 float32 blobs+noise ({nblobs} blobs, seed=rank), unit voxels, default SIFT3D "
                                   f"parameters; detect + describe all keypoints, input and descriptors resident in HBM",
                       "keypoints": K, "extrema_candidates": ncand,
                       "detect_ms": round(t_detect * 1e3, 3), "describe_ms": round(t_describe * 1e3, 3),
                       "parallelism": "1 volume per GPU (weak), no data-path collective" if world > 1 else "1 GPU"},
        }
    if rank == 0 and K and not args.no_match:
        # row f1, outside the timed region: SIFT3D_nn_match of this volume's K descriptors against themselves
        # (device resident; the cost does not depend on the data: 2 exhaustive K x K f64 SSD passes)
        # The first call of a process allocates the matcher's device scratch (4.3 bytes per pair: 4 GB here; 10-130 ms
        # depending on the box): reported apart, the figure is the better of the next two calls.
        t0 = time.perf_counter()
        m = dev.nn_match(d_desc.value, K, d_desc.value, K, 0.8, stride=776)   # records laid out like SIFT3D_Descriptor
        t_first = time.perf_counter() - t0
        t_match = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            m = dev.nn_match(d_desc.value, K, d_desc.value, K, 0.8, stride=776)
            t_match = min(t_match, time.perf_counter() - t0)
        result["config"]["match"] = {"pairs": K * K, "ms": round(t_match * 1e3, 2), "first_call_ms": round(t_first * 1e3, 2),
                                     "self_matches": int((m == np.arange(K)).sum()),
                                     "Gpairs_per_s": round(2.0 * K * K / t_match / 1e9, 1),
                                     "method": "f32 screening of all pairs (forward and backward pass) + exact f64 verification "
                                               "of the candidates; indices bit-identical to the exhaustive f64 search"}
    if rank == 0 and K and not args.no_match:
        # what the descriptor kernel (65 % of the step) works through, outside the timed region: the kernel's counting
        # instantiation returns the number of accepted window voxels per keypoint
        lib.sift.sift3d_amd_describe_window_stats.argtypes = [C.POINTER(abi.SIFT3D), C.POINTER(abi.Keypoint_store),
                                                              C.POINTER(C.c_uint)]
        st = np.zeros((K, 2), np.uint32)
        assert lib.sift.sift3d_amd_describe_window_stats(C.byref(s), C.byref(kp), st.ctypes.data_as(C.POINTER(C.c_uint))) == 0
        wv = int(st[:, 0].astype(np.int64).sum())
        described, redone = C.c_ulonglong(0), C.c_ulonglong(0)
        try:                                     # how often the histogram grid of a keypoint had to be redone
            L = lib.sift
            L.s3d_k_describe_redo_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]
            if L.s3d_k_describe_redo_stats(C.byref(described), C.byref(redone), 0) != 0:
                described, redone = C.c_ulonglong(0), C.c_ulonglong(0)
        except AttributeError:
            pass
        result["config"]["describe_kernel"] = {
            "window_voxels": wv, "lds_atomics_per_window_voxel": 24,
            "Gvox_window_per_s": round(wv / t_describe / 1e9, 1),
            "G_lds_atomic_lane_ops_per_s": round(24.0 * wv / t_describe / 1e9, 1),
            "lds_data_path_floor_ms": round(24.0 * (wv / 64.0) * 4.2 / 256.0 / 2.4e9 * 1e3, 2),
            "windows_described_since_start": int(described.value), "windows_described_twice": int(redone.value),
            "bound": "VALU issue (about 245 instructions per window voxel, 45 of them f64: 10.7 ms at 512^3); the LDS pipe (24 "
                     "conflict-free ds_add_u32 per voxel into 16 bank-private copies of 32-bit fields, 4.2 clk per wave and CU: "
                     "lds_data_path_floor_ms) and the per-keypoint fixed work run under it (two 512-thread workgroups per CU); not "
                     "HBM, not MFMA -- DESIGN.md section 4, profiles/r04_describe_field32.txt"}
    if rank == 0 and not args.no_match:
        # BASELINE configs[0] flavour, outside the timed region: the kpSift3D program on a 128^3 NIfTI-1 volume
        # (.nii.gz in, keypoint and descriptor CSVs out) -- process start, HIP initialisation, zlib and CSV
        # formatting included: plumbing, not throughput.
        exe = os.path.join(ROOT, "sift3d_amd", "bin", "kpSift3D")
        if os.path.exists(exe):
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                v128 = synth.blobs(128, 128, 128, synth.default_nblobs(128, 128, 128), 1)
                him = lib.image_from_numpy(v128)
                src = os.path.join(td, "vol.nii.gz")
                lib.imutil.im_write.argtypes = [C.c_char_p, C.POINTER(abi.Image)]
                if lib.imutil.im_write(src.encode(), C.byref(him)) == 0:
                    t0 = time.perf_counter()
                    r = subprocess.run([exe, "--keys", os.path.join(td, "k.csv"), "--desc", os.path.join(td, "d.csv"), src],
                                       capture_output=True, text=True)
                    t_cli = time.perf_counter() - t0
                    nk = sum(1 for _ in open(os.path.join(td, "k.csv"))) if r.returncode == 0 else -1
                    result["config"]["kpSift3D_128"] = {"wall_ms": round(t_cli * 1e3, 1), "keypoints": nk,
                                                        "returncode": r.returncode}
                lib.free_image(him)
    if rank == 0 and not args.no_match:
        # SURVEY 8(d) "API-to-API", outside the timed region: the same volume through the reference's own entry points
        # with HOST buffers (pageable Image in, host Keypoint_store / SIFT3D_Descriptor_store out): what a relinked
        # caller sees, PCIe transfers included.  Never `value`.
        him = lib.image_from_numpy(vol)
        hs = abi.SIFT3D()
        assert lib.sift.init_SIFT3D(C.byref(hs)) == 0
        hkp = abi.Keypoint_store()
        lib.sift.init_Keypoint_store(C.byref(hkp))
        hd = abi.SIFT3D_Descriptor_store()
        lib.sift.init_SIFT3D_Descriptor_store(C.byref(hd))
        def host_step():
            assert lib.sift.SIFT3D_detect_keypoints(C.byref(hs), C.byref(him), C.byref(hkp)) == 0
            t1 = time.perf_counter()
            assert lib.sift.SIFT3D_extract_descriptors(C.byref(hs), C.byref(hkp), C.byref(hd)) == 0
            return t1

        host_step()                                           # warm-up: allocations, first-touch of the stores
        dev.sync()
        t_begin = time.perf_counter()
        t_det = t_desc = 0.0
        for _ in range(args.steps):                           # the second TIMED figure (SURVEY 8d Metric 1, API-to-API)
            t0 = time.perf_counter()
            t1 = host_step()
            t2 = time.perf_counter()
            t_det += t1 - t0
            t_desc += t2 - t1
        dev.sync()
        t_host = time.perf_counter() - t_begin
        result["config"]["host_api"] = {"steps": args.steps, "ms_per_step": round(t_host / args.steps * 1e3, 3),
                                        "detect_ms": round(t_det / args.steps * 1e3, 2), "describe_ms": round(t_desc / args.steps * 1e3, 2),
                                        "Mvox_s": round(n ** 3 * args.steps / t_host / 1e6, 1), "keypoints": int(hkp.slab.num),
                                        "note": "SIFT3D_detect_keypoints + SIFT3D_extract_descriptors on a pageable host Image, host "
                                                "stores out: 512 MiB up, 97 MB of descriptor records down over PCIe (streamed "
                                                "batch-wise beside the descriptor kernel)"}
        lib.sift.cleanup_SIFT3D_Descriptor_store(C.byref(hd))
        lib.sift.cleanup_Keypoint_store(C.byref(hkp))
        lib.free_image(him)
        lib.sift.cleanup_SIFT3D(C.byref(hs))
    if rank == 0 and not args.no_match:
        # BASELINE configs[2], outside the timed region: SIFT3D_extract_dense_descriptors on a 256^3 volume,
        # device to device (12-channel output, 805 MB)
        nd = 256
        dvol = synth.blobs(nd, nd, nd, synth.default_nblobs(nd, nd, nd), seed=2)
        d_in = dev.upload(dvol)
        d_out = dev.malloc(dvol.nbytes * 12)
        s2 = abi.SIFT3D()
        assert lib.sift.init_SIFT3D(C.byref(s2)) == 0
        ou = (C.c_double * 3)(1.0, 1.0, 1.0)
        times = []
        for _ in range(3):
            dev.sync()
            t0 = time.perf_counter()
            rc = lib.sift.sift3d_amd_extract_dense_dev(C.byref(s2), C.c_void_p(d_in), nd, nd, nd, 1.0, 1.0, 1.0, ou,
                                                       C.c_void_p(d_out))
            dev.sync()
            times.append(time.perf_counter() - t0)
            if rc != 0:
                raise SystemExit("dense failed: " + lib.sift.sift3d_amd_last_error().decode())
        t_dense = min(times[1:])
        result["config"]["dense_256"] = {"ms": round(t_dense * 1e3, 3), "Mvox_s": round(nd ** 3 / t_dense / 1e6, 1),
                                         "blur_alg_GBs": round(288.0 * nd ** 3 / t_dense / 1e9, 1)}
        dev.free(d_in)
        dev.free(d_out)
        lib.sift.cleanup_SIFT3D(C.byref(s2))
    if rank == 0 and not args.no_match:
        # BASELINE configs[4] flavour, outside the timed region: the same volume read as anisotropic slices
        # (units 1 x 1 x 1.5): in-plane passes on the fused kernel, the z pass on the generic one
        s3 = abi.SIFT3D()
        assert lib.sift.init_SIFT3D(C.byref(s3)) == 0
        kp3 = abi.Keypoint_store()
        lib.sift.init_Keypoint_store(C.byref(kp3))
        d3 = C.c_void_p()
        ta = []
        for _ in range(3):
            dev.sync()
            t0 = time.perf_counter()
            lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s3), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.5, C.byref(kp3))
            dev.sync()
            t1 = time.perf_counter()
            if kp3.slab.num:
                lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(s3), C.byref(kp3), C.byref(d3))
            dev.sync()
            ta.append((t1 - t0, time.perf_counter() - t1))
        result["config"]["aniso_1x1x1.5"] = {"detect_ms": round(min(t[0] for t in ta[1:]) * 1e3, 2),
                                             "describe_ms": round(min(t[1] for t in ta[1:]) * 1e3, 2),
                                             "keypoints": int(kp3.slab.num)}
        # BASELINE configs[4]: two anisotropic 512^3 volumes (the second one the first shifted by (3, -2, 1) voxels),
        # detect + describe each, then SIFT3D_nn_match of the two descriptor sets -- everything resident in HBM
        if kp3.slab.num:
            d_vol2 = dev.upload(np.roll(vol, (1, -2, 3), axis=(0, 1, 2)))
            s4 = abi.SIFT3D()
            assert lib.sift.init_SIFT3D(C.byref(s4)) == 0
            kp4 = abi.Keypoint_store()
            lib.sift.init_Keypoint_store(C.byref(kp4))
            d4 = C.c_void_p()
            best = None
            for _ in range(2):
                dev.sync()
                t0 = time.perf_counter()
                lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s3), C.c_void_p(d_vol), n, n, n, 1.0, 1.0, 1.5, C.byref(kp3))
                lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(s3), C.byref(kp3), C.byref(d3))
                lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(s4), C.c_void_p(d_vol2), n, n, n, 1.0, 1.0, 1.5, C.byref(kp4))
                lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(s4), C.byref(kp4), C.byref(d4))
                dev.sync()
                t1 = time.perf_counter()
                mm = dev.nn_match(d3.value, int(kp3.slab.num), d4.value, int(kp4.slab.num), 0.8, stride=776)
                t2 = time.perf_counter()
                if best is None or t2 - t0 < best[0]:
                    best = (t2 - t0, t1 - t0, t2 - t1, int((mm >= 0).sum()))
            result["config"]["two_volume_match"] = {"total_ms": round(best[0] * 1e3, 1), "features_ms": round(best[1] * 1e3, 1),
                                                    "match_ms": round(best[2] * 1e3, 1), "keypoints": [int(kp3.slab.num), int(kp4.slab.num)],
                                                    "matches": best[3], "Mvox_s": round(2 * n ** 3 / best[0] / 1e6, 1)}
            dev.free(d_vol2)
            lib.sift.cleanup_SIFT3D(C.byref(s4))
        lib.sift.cleanup_SIFT3D(C.byref(s3))
    if rank == 0 and not args.no_match:
        # Off the unit-voxel, multiple-of-four grid (outside the timed region): a 0.7 x 0.7 x 1.5 mm volume (clinical CT
        # spacing: taps 1.43 voxels apart in plane, 2/3 along z -- every pass table-driven, s3d_gauss_tab.hip) and a volume
        # whose rows are 511 voxels long.  ratio_to_unit = detect time per voxel over the unit-voxel 512^3 detect of this run.
        unit_ps = t_detect / float(n) ** 3 * 1e12
        for key, dims, units in (("aniso_0.7x0.7x1.5", (512, 512, 300), (0.7, 0.7, 1.5)), ("odd_511", (511, 509, 303), (1.0, 1.0, 1.0))):
            ex, ey, ez = dims
            evol = synth.blobs(ex, ey, ez, synth.default_nblobs(ex, ey, ez), 0)
            d_e = dev.upload(evol)
            se = abi.SIFT3D()
            assert lib.sift.init_SIFT3D(C.byref(se)) == 0
            kpe = abi.Keypoint_store()
            lib.sift.init_Keypoint_store(C.byref(kpe))
            de = C.c_void_p()
            te = []
            for _ in range(4):
                dev.sync()
                t0 = time.perf_counter()
                rc = lib.sift.sift3d_amd_detect_keypoints_dev(C.byref(se), C.c_void_p(d_e), ex, ey, ez, *units, C.byref(kpe))
                dev.sync()
                t1 = time.perf_counter()
                if rc == 0 and kpe.slab.num:
                    rc = lib.sift.sift3d_amd_extract_descriptors_dev(C.byref(se), C.byref(kpe), C.byref(de))
                dev.sync()
                te.append((t1 - t0, time.perf_counter() - t1))
                if rc != 0:
                    raise SystemExit(f"{key} failed: " + lib.sift.sift3d_amd_last_error().decode())
            det = min(t[0] for t in te[1:])
            ps = det / (float(ex) * ey * ez) * 1e12
            result["config"][key] = {"dims": list(dims), "units": list(units), "detect_ms": round(det * 1e3, 2),
                                     "describe_ms": round(min(t[1] for t in te[1:]) * 1e3, 2), "keypoints": int(kpe.slab.num),
                                     "detect_ps_per_voxel": round(ps, 1), "unit_voxel_detect_ps_per_voxel": round(unit_ps, 1),
                                     "ratio_to_unit": round(ps / unit_ps, 3)}
            dev.free(d_e)
            lib.sift.cleanup_Keypoint_store(C.byref(kpe))
            lib.sift.cleanup_SIFT3D(C.byref(se))
    if rank == 0 and not args.no_roofline:
        add_roofline(result, dev, n)
    if rank == 0 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline()
        except Exception as e:            # the baseline is a report, never a reason to lose the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "Mvox/s", "cores": os.cpu_count(), "kind": "port",
                                      "sample": f"failed: {e}"}
        # The in-run sample above is bounded (160^3: ~12 s).  The same reference build on BASELINE configs[1] ITSELF -- the
        # 512^3 volume this line's `value` is quoted on -- was timed once on an MI355X box's host cores (6.4 minutes; SURVEY
        # 8d): quoted here with its provenance, not re-measured per run.
        if args.cpu_baseline_full:
            try:
                result["cpu_baseline"]["configs1_512"] = dict(cpu_baseline(512, timeout=1500), quoted=False)
            except Exception as e:
                result["cpu_baseline"]["configs1_512"] = {"value": None, "quoted": False, "sample": f"failed: {e}"}
        else:
            try:
                full = json.load(open(os.path.join(ROOT, "profiles", "r03_cpu_baseline_512.json")))
                result["cpu_baseline"]["configs1_512"] = dict({k: full[k] for k in ("value", "unit", "cores", "kind", "sample", "provenance")},
                                                              quoted=True)     # not timed in this run: --cpu-baseline-full does that
            except Exception:
                pass
    if rank == 0:
        print(json.dumps(result), flush=True)
    dev.free(d_vol)
    lib.sift.cleanup_SIFT3D(C.byref(s))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
