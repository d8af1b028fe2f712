"""Identity of compiled kernels: SHA-256 over the machine code of selected kernels inside a built library / object.

bench.py stamps its roofline line with HBM-traffic figures that a separate rocprofv3 --pmc pass measured (profiles/
pmc_gauss.json).  Such a figure may only be used for the kernels it was measured on.  The text of the source file is the
wrong key -- a comment edit changes it, the machine code not -- so the key is the code itself: the bytes of the kernels'
functions in the gfx950 code object that hipcc embedded (clang offload bundle -> AMDGPU ELF -> .symtab / .text), read with
nothing but the standard library."""
from __future__ import annotations

import hashlib
import struct

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob: bytes, arch: str = "gfx950"):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        pos = i + len(MAGIC)
        (n,) = struct.unpack_from("<Q", blob, i + 24)
        p = i + 32
        if n > 64:
            continue
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + ts].decode("ascii", "replace")
            p += ts
            if arch in triple and size > 0:
                yield blob[i + off:i + off + size]


def _elf_functions(elf: bytes):
    """{name: bytes} of the STT_FUNC symbols of a little-endian ELF64."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return {}
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for k in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize)
        secs.append((name, typ, addr, off, size, link, entsize))
    out = {}
    for (name, typ, addr, off, size, link, entsize) in secs:
        if typ != 2:                                     # SHT_SYMTAB
            continue
        stroff = secs[link][3]
        for k in range(size // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + k * 24)
            if (st_info & 0xF) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= len(secs):
                continue
            end = elf.index(b"\0", stroff + st_name)
            sym = elf[stroff + st_name:end].decode("ascii", "replace")
            s = secs[st_shndx]
            o = s[3] + (st_value - s[2])
            out[sym] = elf[o:o + st_size]
    return out


def kernel_isa(path: str, substrings, arch: str = "gfx950"):
    """{mangled kernel name: machine-code bytes} of the functions whose name contains one of `substrings`."""
    blob = open(path, "rb").read()
    found = {}
    for co in _code_objects(blob, arch):
        for name, code in _elf_functions(co).items():
            if any(s in name for s in substrings):
                found[name] = code
    return found


def kernel_isa_sha256(path: str, substrings, arch: str = "gfx950") -> str:
    """One digest over the selected kernels (sorted by name; names included).  Empty selection -> ''."""
    fns = kernel_isa(path, substrings, arch)
    if not fns:
        return ""
    h = hashlib.sha256()
    for name in sorted(fns):
        h.update(name.encode() + b"\0" + struct.pack("<Q", len(fns[name])) + fns[name])
    return h.hexdigest()


GAUSS_KERNELS = ("k_gauss_xy", "k_gauss_z")               # the fused Gaussian: what roofline.traffic is about
