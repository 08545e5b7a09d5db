"""Registers, scratch and LDS of the gfx950 kernels in a built library, read from the code objects' metadata (no GPU, no recompile):
python tools/kernel_resources.py [qoi_amd/lib/libqoi_mi355x.so] - used by tests/test_kernel_resources.py to hold the occupancy
the kernels were measured at."""
import struct
import sys

import msgpack

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(blob: bytes):
    """the gfx950 ELF images of every offload bundle in a host object / shared library"""
    pos = blob.find(MAGIC)
    while pos >= 0:
        off = pos + len(MAGIC)
        n, = struct.unpack_from('<Q', blob, off); off += 8
        for _ in range(n):
            o, s, tl = struct.unpack_from('<QQQ', blob, off); off += 24
            triple = blob[off:off + tl].decode(); off += tl
            if 'gfx950' in triple and s:
                yield blob[pos + o:pos + o + s]
        pos = blob.find(MAGIC, pos + 1)


def elf_notes(elf: bytes):
    """(name, type, desc) of every note of a 64-bit little-endian ELF"""
    assert elf[:4] == b'\x7fELF' and elf[4] == 2 and elf[5] == 1
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
    for i in range(shnum):
        sh = elf[shoff + i * shentsize: shoff + (i + 1) * shentsize]
        sh_type, = struct.unpack_from('<I', sh, 4)
        if sh_type != 7:            # SHT_NOTE
            continue
        off, size = struct.unpack_from('<QQ', sh, 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from('<III', elf, p); p += 12
            name = elf[p:p + namesz].rstrip(b'\0').decode(); p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]; p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernels(path: str) -> dict:
    """demangled-ish kernel symbol -> {vgpr, agpr, sgpr, scratch, lds, wavefront_size}"""
    out = {}
    blob = open(path, 'rb').read()
    for elf in code_objects(blob):
        for name, ntype, desc in elf_notes(elf):
            if name != 'AMDGPU' or ntype != 32:       # NT_AMDGPU_METADATA (msgpack)
                continue
            meta = msgpack.unpackb(desc, raw=False)
            for k in meta.get('amdhsa.kernels', []):
                out[k['.name']] = {'vgpr': k['.vgpr_count'], 'agpr': k.get('.agpr_count', 0), 'sgpr': k['.sgpr_count'],
                                   'scratch': k['.private_segment_fixed_size'], 'lds': k['.group_segment_fixed_size'],
                                   'vgpr_spills': k.get('.vgpr_spill_count', 0)}
    return out


def disassembly(path: str, objdump: str = '/opt/rocm/lib/llvm/bin/llvm-objdump') -> dict:
    """kernel symbol -> list of its instruction lines (llvm-objdump -d of the library's gfx950 code objects)"""
    import os
    import re
    import subprocess
    import tempfile
    out = {}
    blob = open(path, 'rb').read()
    for elf in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(elf)
        try:
            text = subprocess.run([objdump, '-d', '--no-show-raw-insn', f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        cur = None
        for line in text.split('\n'):
            m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
            if m:
                cur = out.setdefault(m.group(1), [])
            elif cur is not None and line.strip():
                cur.append(line.strip())
    return out


if __name__ == '__main__':
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else 'qoi_amd/lib/libqoi_mi355x.so')
    for n in sorted(ks):
        print(f"{n:100s} {ks[n]}")
