"""oracle/make_ref.py -- build the two reference-native pieces that compile in this image, from the
sources where they lie under /root/reference, into oracle/_ref/ (git-ignored; travels to the GPU
box like any built .so).  Never copies reference sources into the repo.

  * models/utils/src/Array_Index.cpp              -> oracle/_ref/Array_Index<ext>.so (pybind11 module;
                                                     built WITHOUT -fopenmp, as the shipped setup.py:49-55 does)
  * models/bbox_post_process/src/iou3d_cpu.cpp    -> oracle/_ref/libref_iou3d.so (+ oracle/ref_bind.cpp shim) and
                                                     oracle/_ref/libref_overlap.so (+ ref_bind_overlap.cpp: box_overlap)
    needs <cuda.h>/<cuda_runtime_api.h>: the genuine NVIDIA headers shipped inside this image's
    triton wheel are used (no stand-in headers are written).
TEST INFRASTRUCTURE ONLY.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")


def build(verbose=True):
    if not os.path.isdir(REF):
        return False
    os.makedirs(OUT, exist_ok=True)
    import pybind11
    import torch
    from torch.utils import cpp_extension
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    py_inc = sysconfig.get_paths()["include"]
    a_src = os.path.join(REF, "models/utils/src/Array_Index.cpp")
    a_out = os.path.join(OUT, "Array_Index" + ext)
    if not os.path.exists(a_out):
        cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-I" + pybind11.get_include(), "-I" + py_inc, a_src,
               "-o", a_out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    i_out = os.path.join(OUT, "libref_iou3d.so")
    if not os.path.exists(i_out):
        import triton
        cuda_inc = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
        assert os.path.exists(os.path.join(cuda_inc, "cuda_runtime_api.h")), "no CUDA headers in this image"
        incs = ["-I" + p for p in cpp_extension.include_paths()] + ["-I" + py_inc, "-I" + cuda_inc]
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-w", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(
            torch._C._GLIBCXX_USE_CXX11_ABI)] + incs + [
            os.path.join(REF, "models/bbox_post_process/src/iou3d_cpu.cpp"), os.path.join(HERE, "ref_bind.cpp"),
            "-L" + libdir, "-Wl,-rpath," + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-o", i_out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    o_out = os.path.join(OUT, "libref_overlap.so")
    if not os.path.exists(o_out):
        import triton
        cuda_inc = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
        incs = ["-I" + p for p in cpp_extension.include_paths()] + ["-I" + py_inc, "-I" + cuda_inc]
        libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
        # the shim and the reference file form one translation unit (box_overlap is `inline` there)
        cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-w", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(
            torch._C._GLIBCXX_USE_CXX11_ABI)] + incs + [
            "-include", os.path.join(REF, "models/bbox_post_process/src/iou3d_cpu.cpp"),
            os.path.join(HERE, "ref_bind_overlap.cpp"), "-L" + libdir, "-Wl,-rpath," + libdir, "-ltorch", "-ltorch_cpu",
            "-lc10", "-o", o_out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return True


if __name__ == "__main__":
    ok = build()
    print("oracle/_ref built" if ok else "no /root/reference here; nothing built")
    sys.exit(0)
