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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Xarch_host", "-mavx2", "-Xarch_host", "-mfma",  # host-side l x l factorizations (small_linalg.hpp)
           "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "corrla_rsvd.hip"),
           "-o", LIB_PATH, "-L" + os.path.join(ROCM, "lib"), "-lrccl",
           "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


def build_emu(force=False, verbose=False):
    """g++ -> tests/emu/libcorrla_emu.so (test infrastructure: host emulation of the device backend)"""
    srcs = _sources(CSRC, (".hpp", ".h")) + [os.path.join(EMU_DIR, "emu_backend.cpp"),
                                             os.path.join(ROOT, "include", "corrla_rsvd.h")]
    if not force and _newer(EMU_PATH, srcs):
        return EMU_PATH
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
           os.path.join(EMU_DIR, "emu_backend.cpp"), "-o", EMU_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return EMU_PATH


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_product(force=force, verbose=True))
    if "--emu" in sys.argv:
        print(build_emu(force=force, verbose=True))
