#!/usr/bin/env python3
"""Registers, LDS, scratch and spills of every kernel in the shipped library, read from the code objects inside
lowthrustopt_amd/liblto_hip.so (not from a build log): the figures DESIGN.md quotes per kernel come from here.

  python tools/kernel_resources.py [path/to/liblto_hip.so] > profiles/<tag>_kernel_resources.txt

The shared object carries one clang offload bundle per translation unit (magic __CLANG_OFFLOAD_BUNDLE__; entries: offset, size,
triple); the gfx950 entry of each is an ELF whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count / .agpr_count / .sgpr_count /
.group_segment_fixed_size (LDS bytes) / .private_segment_fixed_size (scratch bytes per lane) / .vgpr_spill_count / .sgpr_spill_count.
Waves per SIMD follow from the unified register file of CDNA3/4: 512 registers per lane and SIMD, allocated in blocks of 8;
.vgpr_count is the unified total (arch VGPRs + AGPRs)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """(triple, bytes) of every device entry of every offload bundle in `blob`."""
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", blob, i + len(MAGIC))
        q = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "amdgcn" in triple and size:
                yield triple, blob[i + off:i + off + size]
        pos = i + len(MAGIC)


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes); f.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
    out = []
    for block in re.split(r"\n\s+- \.agpr_count:", "\n" + txt)[1:]:
        block = ".agpr_count:" + block
        kv = dict(re.findall(r"^\s*(\.[a-z_]+):\s+(\S+)\s*$", block, flags=re.M))
        if ".name" in kv and ".vgpr_count" in kv:
            out.append(kv)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return [re.sub(r"^void ", "", ln).replace("lto::", "").replace("(IndirectArgs)", "").replace("(DirectArgs)", "") for ln in p.stdout.splitlines()]


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "lowthrustopt_amd", "liblto_hip.so")
    blob = open(so, "rb").read()
    rows = []
    for triple, elf in code_objects(blob):
        if "gfx950" not in triple:
            raise SystemExit("code object for %s: this library is gfx950 only" % triple)
        rows += kernels_of(elf)
    names = demangle([k[".name"] for k in rows])
    print("# %s: %d kernels in %d code objects (gfx950)" % (os.path.relpath(so, ROOT), len(rows), sum(1 for _ in code_objects(blob))))
    print("# vgpr = unified register count (arch VGPRs + AGPRs), agpr = the accumulation VGPRs among them (spill space here: no MFMA),")
    print("# lds / scratch in bytes (scratch per lane)")
    print("# waves/SIMD = 512 // roundup(vgpr, 8), at most 8; a workgroup's LDS can lower it further")
    print("%-84s %5s %5s %5s %7s %8s %7s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "vspill", "sspill", "w/SIMD"))
    for name, k in sorted(zip(names, rows)):
        v, a = int(k[".vgpr_count"]), int(k.get(".agpr_count", 0))
        tot = -(-v // 8) * 8                       # .vgpr_count is the unified total: arch VGPRs + AGPRs
        print("%-84s %5d %5d %5d %7d %8d %7d %7d %6d" % (name[:84], v, a, int(k[".sgpr_count"]), int(k.get(".group_segment_fixed_size", 0)),
                                                         int(k.get(".private_segment_fixed_size", 0)), int(k.get(".vgpr_spill_count", 0)),
                                                         int(k.get(".sgpr_spill_count", 0)), min(8, 512 // max(tot, 1))))


if __name__ == "__main__":
    main()
