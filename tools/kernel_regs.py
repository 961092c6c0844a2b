#!/usr/bin/env python3
"""Register / scratch figures of the gfx950 kernels in an object file or in libipdamg.so (no GPU needed):
splits the clang offload bundle out of the .hip_fatbin section and reads the kernels' metadata notes.
  python tools/kernel_regs.py [FILE] [NAME-SUBSTRING ...]      (default FILE: csrc/build/ipd_cycle.o)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    pos = data.find(magic)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(magic))[0]
        p = pos + len(magic) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", data, p)
            ident = data[p + 24:p + 24 + idlen].decode()
            p += 24 + idlen
            if "gfx950" in ident and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(magic, pos + 1)
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(
        ROOT, "codes_of_ipd_ssn_amg_method_amd", "csrc", "build", "ipd_cycle.o")
    pats = [a for a in sys.argv[1:] if a != path]
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            dem = dem.split("(")[0]
            if pats and not any(p in dem for p in pats):
                continue
            g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, "?"])[1]
            print("%-60s vgpr %3s spill %3s sgpr_spill %3s scratch %4s B lds %6s" % (
                dem[-60:], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                g("private_segment_fixed_size"), g("group_segment_fixed_size")))


if __name__ == "__main__":
    main()
