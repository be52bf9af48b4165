"""Builds the compiled PyTorch binding in-tree: tinycudann/_tcnn_ext.so (g++ against torch's headers; no device code, nothing hipified --
ext/torch_module.cpp is plain C++ over the C ABI).  Called by __graft_entry__.build(); `python -m tinycudann.ext.build_ext` from tiny-cuda-nn_amd/ works too."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "torch_module.cpp")
OUT = os.path.join(os.path.dirname(HERE), "_tcnn_ext.so")


def build(force=False):
    import torch
    from torch.utils import cpp_extension as ce
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_tcnn_ext",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    cmd += ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(rocm, "include")]
    cmd += [SRC, "-o", OUT, "-L" + torch_lib, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-ltorch_hip", "-lc10", "-lc10_hip", "-ldl", "-Wl,-rpath," + torch_lib]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
