"""Test helper: read the AMDGPU kernel metadata (register counts, LDS, workgroup size) of every gfx950 kernel in a HIP shared
library — the .hip_fatbin section holds one clang offload bundle per translation unit, each with an ELF code object whose
NT_AMDGPU_METADATA note is a msgpack map."""
from __future__ import annotations

import struct

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_section(data: bytes, name: bytes):
    assert data[:4] == b"\x7fELF" and data[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    secs = []
    for i in range(shnum):
        o = shoff + i * shentsize
        sh_name, sh_type, _flags, _addr, sh_off, sh_size = struct.unpack_from("<IIQQQQ", data, o)
        secs.append((sh_name, sh_type, sh_off, sh_size))
    stro = secs[shstrndx][2]
    for sh_name, sh_type, sh_off, sh_size in secs:
        end = data.index(b"\0", stro + sh_name)
        if data[stro + sh_name:end] == name:
            return data[sh_off:sh_off + sh_size]
    return None


def _code_objects(fatbin: bytes, arch: str):
    pos = 0
    while True:
        pos = fatbin.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", fatbin, pos + len(MAGIC))
        o = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", fatbin, o)
            triple = fatbin[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            if arch in triple and size:
                yield fatbin[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernel_metadata(so_path: str, arch: str = "gfx950") -> dict:
    """{demangled-or-mangled kernel name: metadata dict} for every kernel of `arch` in the library"""
    data = open(so_path, "rb").read()
    fat = _elf_section(data, b".hip_fatbin")
    assert fat is not None, "no .hip_fatbin section"
    out = {}
    for co in _code_objects(fat, arch):
        note = _elf_section(co, b".note")
        o = 0
        while note is not None and o + 12 <= len(note):
            namesz, descsz, ntype = struct.unpack_from("<III", note, o)
            o += 12
            name = note[o:o + namesz]
            o += (namesz + 3) & ~3
            desc = note[o:o + descsz]
            o += (descsz + 3) & ~3
            if name.rstrip(b"\0") == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
    return out
