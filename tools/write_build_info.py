"""Written by the Makefile rule that installs libaslam_core.so (awesomeslam_amd/csrc/Makefile), AFTER the guards have passed and the library
has been moved into place: csrc/build_info.json records what that very file is -- its sha256, a hash of the sources it was compiled from, the
guards that scanned its assembly, compiler and time.  awesomeslam_amd.core.build_info() recomputes the sha256 of the library it is about to
load and reports a mismatch (the record then describes some other binary); __graft_entry__.smoke() refuses to run on one.

    python tools/write_build_info.py <library> <guards: 0|1> <ukf: 0|1> <source> [<source> ...]
"""
import hashlib
import json
import os
import subprocess
import sys
import time


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def sources_hash(paths):
    """one hash over (base name, content) of every source, in name order: independent of the checkout directory"""
    h = hashlib.sha256()
    for p in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(p).encode() + b"\0" + sha256(p).encode() + b"\n")
    return h.hexdigest()


def main():
    lib, guards, ukf, srcs = sys.argv[1], sys.argv[2] == "1", sys.argv[3] == "1", sys.argv[4:]
    try:
        ver = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        ver = next((ln.strip() for ln in ver.splitlines() if "HIP version" in ln), ver.strip().splitlines()[0] if ver.strip() else "?")
    except OSError:
        ver = "?"
    info = {
        "library": os.path.basename(lib),
        "sha256": sha256(lib),
        "sources_sha256": sources_hash(srcs),
        "sources": sorted(os.path.basename(p) for p in srcs),
        "built": time.strftime("%Y-%m-%d %H:%M:%S"),
        "host": os.uname().nodename,
        "hipcc": ver,
        "arch": "gfx950",
        "ukf": ukf,
        "guards": ("check_spill_exec + check_agpr_strip + check_vmcnt_protocol + check_dpp_hazard clean on the assembly of this compilation" if guards
                   else "NOT RUN (GUARDS=0): diagnostic library, never loaded by awesomeslam_amd.core"),
    }
    out = os.path.join(os.path.dirname(os.path.abspath(lib)), "build_info.json" if guards else "build_info_unguarded.json")
    with open(out, "w") as f:
        json.dump(info, f, indent=1)
    print(f"write_build_info: {out}: {info['library']} sha256 {info['sha256'][:16]}... sources {info['sources_sha256'][:16]}...")


if __name__ == "__main__":
    main()
