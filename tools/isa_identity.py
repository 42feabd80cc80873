#!/usr/bin/env python3
"""Per-kernel machine-code identity of two builds (refactoring check: a source reorganisation must not change the ISA).

    python tools/isa_identity.py dump <out.json> <object.o | code-object> ...     sha256 of every kernel's .text bytes and of its
                                                                                  kernel descriptor (.kd: registers, LDS, scratch)
    python tools/isa_identity.py diff <a.json> <b.json>                            kernels that differ / exist on one side only

Whole-file comparison of the .hip_fatbin section does not work for this: clang names a `__hip_cuid_<hash>` symbol after a hash of
the source file, so any edit (a comment) changes the fatbin.  Kernels are position-independent, so their bytes can be compared
one by one even when they move between translation units."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(path, tmp):
    """path is a host object with a .hip_fatbin section, or already a gfx950 code object."""
    with open(path, "rb") as f:
        head = f.read(20)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":  # EM_AMDGPU
        return path
    fat = os.path.join(tmp, os.path.basename(path) + ".fatbin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section=.hip_fatbin=" + fat, path])
    co = os.path.join(tmp, os.path.basename(path) + ".co")
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co, "--unbundle"])
    return co


def kernels(co):
    sec = {}
    for ln in subprocess.check_output([LLVM + "/llvm-readelf", "-S", "-W", co], text=True).splitlines():
        m = re.match(r"\s*\[\s*(\d+)\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", ln)
        if m:
            sec[int(m.group(1))] = (m.group(2), int(m.group(3), 16), int(m.group(4), 16))
    data = open(co, "rb").read()
    out = {}
    for ln in subprocess.check_output([LLVM + "/llvm-readelf", "-s", "-W", co], text=True).splitlines():
        f = ln.split()
        if len(f) == 8 and f[3] in ("FUNC", "OBJECT") and f[6].isdigit():
            addr, size, ndx, name = int(f[1], 16), int(f[2]), int(f[6]), f[7]
            if (f[3] == "FUNC" or name.endswith(".kd")) and ndx in sec and size:
                _, saddr, soff = sec[ndx]
                blob = bytearray(data[soff + addr - saddr: soff + addr - saddr + size])
                if name.endswith(".kd") and size == 64:
                    blob[16:24] = b"\0" * 8  # kernel_code_entry_byte_offset: where the code lies relative to the descriptor (layout, not code)
                out[name] = hashlib.sha256(bytes(blob)).hexdigest()[:16] + ":%d" % size
    return out


def main():
    if sys.argv[1] == "dump":
        allk = {}
        with tempfile.TemporaryDirectory() as tmp:
            for p in sys.argv[3:]:
                for k, v in kernels(code_object(p, tmp)).items():
                    assert k not in allk or allk[k] == v, "kernel %s defined twice with different code" % k
                    allk[k] = v
        json.dump(allk, open(sys.argv[2], "w"), indent=0, sort_keys=True)
        print("%d symbols -> %s" % (len(allk), sys.argv[2]))
    else:
        a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
        # the clock build renames kernels; anonymous-namespace kernels keep their names across translation units
        only_a, only_b = sorted(set(a) - set(b)), sorted(set(b) - set(a))
        diff = sorted(k for k in set(a) & set(b) if a[k] != b[k])
        print("common %d, identical %d, different %d, only in A %d, only in B %d" % (len(set(a) & set(b)), len(set(a) & set(b)) - len(diff), len(diff), len(only_a), len(only_b)))
        for k in diff:
            print("  DIFFERENT", k, a[k], b[k])
        for k in only_a:
            print("  only A", k)
        for k in only_b:
            print("  only B", k)
        sys.exit(1 if (diff or only_a or only_b) else 0)


if __name__ == "__main__":
    main()
