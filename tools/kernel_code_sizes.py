#!/usr/bin/env python3
"""Code bytes of every kernel in liberl_hip.so (round 5: on some boxes of the pool a workgroup's instruction fetch is ~10x slower once the
code paths sharing an instruction cache exceed its 64 KB -- tools/clock_probe.hip's code walks -- so a kernel's size is a performance figure).
Walks the clang offload bundles in the library's .hip_fatbin, pulls out the gfx950 code objects and reads their symbol tables.
    python tools/kernel_code_sizes.py [lib] [min_bytes]"""
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "elegantrl_amd", "lib", "liberl_hip.so")
min_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
READELF, CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-readelf", "c++filt"
data = open(lib, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
rows = []
pos = data.find(MAGIC)
with tempfile.TemporaryDirectory() as tmp:
    n = 0
    while pos >= 0:
        (cnt,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                f = os.path.join(tmp, f"co{n}.elf")
                n += 1
                open(f, "wb").write(data[pos + off:pos + off + size])
                out = subprocess.run([READELF, "-sW", f], capture_output=True, text=True).stdout
                for ln in out.splitlines():
                    p = ln.split()
                    if len(p) >= 8 and p[3] == "FUNC" and p[2].isdigit() and int(p[2]) >= min_bytes:
                        rows.append((int(p[2]), p[7]))
        pos = data.find(MAGIC, pos + len(MAGIC))
names = subprocess.run([CXXFILT], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (size, _), name in sorted(set(zip(rows, names)), key=lambda x: -x[0][0]):
    name = name.replace("(anonymous namespace)::", "")
    print(f"{size:8d}  {size / 1024:6.1f} KB  {name[:150]}")
