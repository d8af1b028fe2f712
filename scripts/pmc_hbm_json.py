#!/usr/bin/env python3
"""Turns the four rocprofv3 --pmc databases written by scripts/pmc_hbm.sh into
  gpurun_out/<tag>_pmc_gauss.json / .md   HBM bytes per launch of the fused Gaussian kernels (bench.py's roofline.traffic),
                                          stamped with the commit, the run tag and the SHA-256 of csrc/s3d_gauss.hip
  gpurun_out/<tag>_pmc_hbm_step.md        the same two counters for every kernel of a detect + describe step.
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB): gfx950 reports half the bytes of wide coalesced streaming reads
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is."""
import collections
import glob
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, commit = sys.argv[1], sys.argv[2]
OUT = os.path.join(ROOT, "gpurun_out")


def per_kernel(workload, counter):
    dbs = glob.glob(os.path.join(OUT, f"{tag}_{workload}_{counter}", "**", "*.db"), recursive=True)
    res = collections.defaultdict(lambda: [0.0, 0])
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if "pmc_event" in t][0]
        info = [t for t in tabs if "info_pmc" in t][0]
        disp = [t for t in tabs if "kernel_dispatch" in t][0]
        sym = [t for t in tabs if "kernel_symbol" in t][0]
        q = (f"select s.kernel_name, sum(e.value), count(distinct e.event_id) from {pmc} e join {info} i on e.pmc_id=i.id "
             f"join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id where i.name='{counter}' group by s.kernel_name")
        for k, v, c in cur.execute(q):
            res[k][0] += v
            res[k][1] += c
    return res


def short(name):
    m = re.match(r"_Z\d+(k_[a-z0-9_]+?)(I.*)?$", name.replace(".kd", ""))
    if name.startswith("_Z"):
        import subprocess
        try:
            d = subprocess.run(["c++filt", name.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
            d = re.sub(r"^void ", "", d)
            return re.sub(r"\(.*$", "", d)
        except Exception:
            pass
    return m.group(1) if m else name[:40]


sha = hashlib.sha256(open(os.path.join(ROOT, "sift3d_amd", "csrc", "s3d_gauss.hip"), "rb").read()).hexdigest()
sys.path.insert(0, ROOT)
from sift3d_amd import codeobj          # noqa: E402
isa = codeobj.kernel_isa_sha256(os.path.join(ROOT, "sift3d_amd", "lib", "libsift3d_amd.so"), codeobj.GAUSS_KERNELS)
# ---- (1) the fused Gaussians ----
f, w = per_kernel("gauss_only", "FETCH_SIZE"), per_kernel("gauss_only", "WRITE_SIZE")
nvox = 512 ** 3
kern = {}
for k in sorted(f):
    if "gauss" not in k:
        continue
    fk, wk = f[k][0] / f[k][1], w[k][0] / max(w[k][1], 1)
    bytes_ = (2.0 * fk + wk) * 1024.0
    alg = 16 if "gauss_xy" in k else 8
    kern[short(k)] = {"fetch_kb_raw": fk, "write_kb": wk, "hbm_bytes_per_launch": bytes_, "bytes_per_voxel": bytes_ / nvox,
                      "algorithmic_bytes_per_voxel": alg, "launches": f[k][1]}
doc = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of scripts/gauss_only.py: 512^3 "
               "f32, 3 launches per kernel, 1x MI355X.  Counter unit KB.  FETCH_SIZE doubled (gfx950 reports half the bytes of wide "
               "coalesced streaming reads, MI355X_MICROARCH.md section HBM); WRITE_SIZE taken as is.",
       "run": tag, "commit": commit, "gauss_source_sha256": sha, "gauss_kernels_isa_sha256": isa, "voxels": nvox, "kernels": kern}
json.dump(doc, open(os.path.join(OUT, f"{tag}_pmc_gauss.json"), "w"), indent=1)
with open(os.path.join(OUT, f"{tag}_pmc_gauss.md"), "w") as o:
    o.write(f"# {tag} -- HBM traffic of the fused Gaussian kernels from PMC counters (commit {commit})\n\n{doc['note']}\n\n")
    o.write("| kernel | FETCH_SIZE KB (raw) | WRITE_SIZE KB | HBM bytes/launch (2*FETCH+WRITE) | B/voxel | algorithmic B/voxel |\n|---|---:|---:|---:|---:|---:|\n")
    for k, v in kern.items():
        o.write(f"| `{k}` | {v['fetch_kb_raw']:.0f} | {v['write_kb']:.0f} | {v['hbm_bytes_per_launch']:.4g} | {v['bytes_per_voxel']:.2f} | {v['algorithmic_bytes_per_voxel']} |\n")
# ---- (2) every kernel of a step ----
f, w = per_kernel("describe_only", "FETCH_SIZE"), per_kernel("describe_only", "WRITE_SIZE")
rows = []
for k in f:
    fk, wk = f[k][0], w.get(k, [0.0, 0])[0]
    rows.append((2.0 * fk + wk, short(k), f[k][1], fk, wk))
rows.sort(reverse=True)
with open(os.path.join(OUT, f"{tag}_pmc_hbm_step.md"), "w") as o:
    o.write(f"# {tag} -- HBM traffic per kernel, one detect + three describes at 512^3 (scripts/describe_only.py; commit {commit})\n\n"
            "Sums over all launches of the run; HBM MB = (2 * FETCH_SIZE + WRITE_SIZE) KB / 1024.\n\n"
            "| kernel | launches | FETCH_SIZE KB (raw) | WRITE_SIZE KB | HBM MB | HBM MB per launch |\n|---|---:|---:|---:|---:|---:|\n")
    for tot, k, n, fk, wk in rows[:40]:
        o.write(f"| `{k}` | {n} | {fk:.0f} | {wk:.0f} | {tot / 1024:.1f} | {tot / 1024 / max(n, 1):.1f} |\n")
print(open(os.path.join(OUT, f"{tag}_pmc_gauss.md")).read())
print(open(os.path.join(OUT, f"{tag}_pmc_hbm_step.md")).read())
