"""Builds libpymasc_hip.so (the C-ABI library of include/pymasc_amd.h) in-tree with hipcc for gfx950, and
libpymasc_io.so (include/pymasc_amd_io.h, host-side readers) with g++.

The .so is git-ignored but travels with the gpurun snapshot; hipcc cross-compiles without a GPU.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpymasc_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_hash():
    """sha256 (first 16 hex digits) over the sources libpymasc_hip.so is built from.  build() compiles it into the library
    (pmx_build_id), and the profile summaries under profiles/ carry the hash of the build they were measured on, so that
    bench.py only quotes counters that belong to the library it loaded."""
    import hashlib
    h = hashlib.sha256()
    files = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "pymasc_amd.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "pymasc_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-atomic-optimizer-strategy=DPP: the record cursors are bumped with one LDS atomic per wave; the
    # default "Iterative" strategy serialises over the active lanes with a scalar loop (~2k cycles per atomic).
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP",
           "-Wall", "-Wno-unused-function", '-DPMX_SRC_SHA16="%s"' % source_hash(), *extra_flags, "-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


IO_LIB = os.path.join(HERE, "libpymasc_io.so")


def io_sources():
    return sorted(glob.glob(os.path.join(CSRC, "io", "*.cpp")))


def build_io(force=False, verbose=False):
    """libpymasc_io.so (include/pymasc_amd_io.h): the host-side BAM / BigWig readers, plain g++ + zlib."""
    deps = io_sources() + glob.glob(os.path.join(CSRC, "io", "*.h")) + [os.path.join(HERE, "..", "include",
                                                                                      "pymasc_amd_io.h")]
    if not force and os.path.exists(IO_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(IO_LIB) for d in deps):
        return IO_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", IO_LIB,
           *io_sources(), "-lz", "-pthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return IO_LIB


INGEST_LIB = os.path.join(HERE, "libpymasc_ingest.so")


def ingest_sources():
    return sorted(glob.glob(os.path.join(CSRC, "ingest", "*.hip")))


def build_ingest(force=False, verbose=False):
    """libpymasc_ingest.so (include/pymasc_amd_ingest.h): BGZF inflate + BAM record decode on the device, hipcc for gfx950.
    A library of its own: libpymasc_hip.so's build id (source_hash) covers the cross-correlation kernels only."""
    deps = ingest_sources() + glob.glob(os.path.join(CSRC, "ingest", "*.inc")) + [os.path.join(HERE, "..", "include", "pymasc_amd_ingest.h"),
                                                                                     os.path.abspath(__file__)]
    if not force and os.path.exists(INGEST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(INGEST_LIB) for d in deps):
        return INGEST_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP",
           # the inflate kernel's decoder is uniform control flow; left alone by the structurizer (the backend's default is to
           # structurize uniform regions too: flags and exec-mask tests around scalar branches) it runs matches 40 -> 48 GB/s,
           # BAM-like members 28 -> 33 GB/s (same-box A/B, tools/ingest_paths.py; no effect on libpymasc_hip.so's kernels)
           "-mllvm", "-structurizecfg-skip-uniform-regions=1",
           "-Wall", "-Wno-unused-function", "-pthread",
           "-o", INGEST_LIB] + ingest_sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return INGEST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_io(force="--force" in sys.argv, verbose=True))
    print(build_ingest(force="--force" in sys.argv, verbose=True))
