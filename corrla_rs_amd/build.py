"""Build script: compiles the HIP product library for gfx950 (hipcc cross-compiles without a GPU)
and, for tests only, the host emulation library.  Usage: python -m corrla_rs_amd.build [--emu]"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcorrla_rsvd.so")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_PATH = os.path.join(EMU_DIR, "libcorrla_emu.so")
EMU_ASAN_PATH = os.path.join(EMU_DIR, "libcorrla_emu_asan.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources(d, exts):
    out = []
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return out


def build_product(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> corrla_rs_amd/lib/libcorrla_rsvd.so"""
    srcs = _sources(CSRC, (".hip", ".hpp", ".h")) + [os.path.join(ROOT, "include", "corrla_rsvd.h")]
    if not force and _newer(LIB_PATH, srcs):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcorrla_rsvd.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    # hipcc parses the translation unit twice (device pass, then host pass, about a minute apart): a header edited in between
    # gives a library whose host code launches kernels its device code object does not hold (seen twice in round 3 as
    # "Cannot find Symbol" at the first launch on the GPU box).  So: compile a private copy of the sources, and put the
    # result in place with one rename -- a snapshot of the tree taken during a build sees the old library or the new one.
    import tempfile
    with tempfile.TemporaryDirectory(prefix=".corrla_build_", dir=LIB_DIR) as tmp:  # inside the tree: nothing is written elsewhere
        csrc = os.path.join(tmp, "corrla_rs_amd", "csrc")  # same relative layout: the sources include "../../include/..."
        shutil.copytree(CSRC, csrc)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
        out_tmp = os.path.join(LIB_DIR, ".libcorrla_rsvd.so.%d" % os.getpid())
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-Xarch_host", "-mavx2", "-Xarch_host", "-mfma",  # host-side l x l factorizations (small_linalg.hpp)
               "-I" + os.path.join(tmp, "include"), os.path.join(csrc, "corrla_rsvd.hip"),
               "-o", out_tmp, "-L" + os.path.join(ROCM, "lib"), "-lrccl",
               "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd)
            os.replace(out_tmp, LIB_PATH)
        finally:
            if os.path.exists(out_tmp):
                os.remove(out_tmp)
    return LIB_PATH


def build_emu(force=False, verbose=False, asan=False):
    """g++ -> tests/emu/libcorrla_emu.so (test infrastructure: host emulation of the device backend).
    asan=True builds tests/emu/libcorrla_emu_asan.so with -fsanitize=address,undefined: the host-side driver and C-ABI
    glue (driver.hpp, capi_impl.hpp, small_linalg.hpp) under the sanitizers -- CPU only, never on the GPU box.  Run the
    CPU suite against it with
        CORRLA_EMU_ASAN=1 LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 \
            python -m pytest tests -m "not gpu" """
    srcs = _sources(CSRC, (".hpp", ".h")) + [os.path.join(EMU_DIR, "emu_backend.cpp"),
                                             os.path.join(ROOT, "include", "corrla_rsvd.h")]
    out = EMU_ASAN_PATH if asan else EMU_PATH
    if not force and _newer(out, srcs):
        return out
    opt = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if asan else ["-O2"]
    cmd = ["g++", *opt, "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
           os.path.join(EMU_DIR, "emu_backend.cpp"), "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_product(force=force, verbose=True))
    if "--emu" in sys.argv:
        print(build_emu(force=force, verbose=True))
    if "--emu-asan" in sys.argv:
        print(build_emu(force=force, verbose=True, asan=True))
