#!/usr/bin/env python3
"""Builds oracle/_ref/libtcnn_ref.so: the REFERENCE'S OWN device code for the hot path, compiled for the host.

TEST INFRASTRUCTURE.  Purpose: pin the restated oracle (oracle/tcnn_oracle.c) against the reference's arithmetic, bit for bit,
wherever the reference's code can be compiled without nvcc -- everything on the path except the CUTLASS GEMMs (the submodule is
not in /root/reference); the fully fused network kernels of src/fully_fused_mlp.cu compile against oracle/ref_shim/mma.h, a host
nvcuda::wmma (second translation unit, oracle/ref_driver_mlp.cpp).

How: the reference's kernels are plain C++ between `__global__` and a handful of cuda_fp16 intrinsics.
  * include/tiny-cuda-nn/{common.h, vec.h, common_device.h} and dependencies/pcg32/pcg32.h are compiled WHOLE, where they lie under
    /root/reference, against oracle/ref_shim/cuda_fp16.h (a host __half with device rounding, threadIdx / blockIdx, serial atomicAdd);
    __CUDA_ARCH__ = 610 selects the device code paths (the __hfma2 / __half2-atomic forms of vec.h:327-395; the sm_70+ variants
    of the same operations are PTX `red` instructions with identical arithmetic);
  * from the headers whose classes need the CUDA runtime (encodings/grid.h, optimizers/adam.h, losses/*.h, random.h,
    encodings/{frequency,oneblob,identity}.h, encodings/multi_level_interface.h) only the kernels named in KERNELS below are taken:
    this script locates each definition by name in the file where it lies, brace-matches it, and hands the text to the compiler
    through a temporary file that is deleted after the compile.  No reference source is written into the repository; the only
    outputs are oracle/_ref/libtcnn_ref.so and oracle/_ref/manifest.json (file, kernel, line range of every definition compiled:
    what the parity tests cite).
  * oracle/ref_driver.cpp (ours) gives the kernels an extern "C" face: it loops (blockIdx, threadIdx) over the launch grid the
    reference's host code would use and calls the kernel body for every thread; oracle/ref_driver_mlp.cpp does the same for the
    kernels that synchronise (a block's threads are fibers, __syncthreads() switches between them).

usage: python oracle/build_ref.py [--reference /root/reference] [--force]     (a no-op when the reference tree is absent)
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libtcnn_ref.so")
MANIFEST = os.path.join(OUT_DIR, "manifest.json")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"  # a host compiler with _Float16 in C++ (g++ 11 has none)

INC = "include/tiny-cuda-nn/"
# header -> definitions compiled from it (kernels, and the helper structs / device functions they use)
KERNELS = [
    (INC + "encodings/multi_level_interface.h", ["line:static constexpr uint32_t MAX_N_LEVELS", "ParamsOffsetTable"]),
    (INC + "encodings/grid.h", ["kernel_grid", "kernel_grid_backward", "kernel_grid_backward_input", "kernel_grid_backward_input_backward_grid",
                                "kernel_grid_backward_input_backward_input", "kernel_grid_backward_input_backward_dLdoutput"]),
    (INC + "optimizers/adam.h", ["adam_step"]),
    (INC + "losses/l2.h", ["l2_loss"]),
    (INC + "losses/relative_l2.h", ["relative_l2_loss"]),
    (INC + "losses/l1.h", ["l1_loss"]),
    (INC + "losses/relative_l1.h", ["relative_l1_loss"]),
    (INC + "losses/mape.h", ["mape_loss"]),
    (INC + "losses/smape.h", ["smape_loss"]),
    (INC + "losses/relative_l2_luminance.h", ["relative_l2_luminance_loss"]),
    (INC + "losses/cross_entropy.h", ["cross_entropy_loss"]),
    (INC + "losses/variance_is.h", ["variance_is_loss"]),
    (INC + "optimizers/ema.h", ["ema_step_full_precision", "ema_step_half_precision"]),
    (INC + "random.h", ["generate_random_kernel"]),
    (INC + "encodings/identity.h", ["identity", "identity_backward"]),
    (INC + "encodings/frequency.h", ["frequency_encoding", "frequency_encoding_backward"]),
    (INC + "encodings/oneblob.h", ["kernel_one_blob_soa", "kernel_one_blob_backward"]),
]
# third translation unit (oracle/ref_driver_host.cpp): host-side member functions, compiled inside stand-ins for their classes; one
# include file per place they are included at
KERNELS_HOST = {
    "ref_extracted_host_functions.inc": [(INC + "common_host.h", ["decl:uint32_t powi"]),
                                         (INC + "encodings/multi_level_interface.h", ["line:static constexpr uint32_t MAX_N_LEVELS", "ParamsOffsetTable"])],
    "ref_extracted_matrix_init.inc": [(INC + "gpu_matrix.h", ["initialize_uniform", "initialize_xavier_uniform", "initialize_siren_uniform", "initialize_siren_uniform_first"])],
    "ref_extracted_mlp_init.inc": [("src/fully_fused_mlp.cu", ["FullyFusedMLP<T, WIDTH>::initialize_params"])],
    "ref_extracted_grid_ctor.inc": [(INC + "encodings/grid.h", ["ctor:GridEncodingTemplated"])],
}
# second translation unit (oracle/ref_driver_mlp.cpp): the fully fused network kernels, against oracle/ref_shim/mma.h (nvcuda::wmma for the
# host) and with a thread block's threads as fibers; in source order (each is declared before it is used)
KERNELS_MLP = [
    ("src/fully_fused_mlp.cu", ["threadblock_layer", "threadblock_load_input_static", "kernel_mlp_fused_backward", "threadblock_input_layer_forward_dynamic",
                                "threadblock_last_layer_forward", "threadblock_write_output_static", "kernel_mlp_fused"]),
]


def extract(path, name):
    """-> (text, first_line, last_line) of the definition of `name` (function or struct) in `path`, template header included."""
    lines = open(path).read().split("\n")
    if name.startswith("line:"):  # a one-line definition (a constant)
        for i, line in enumerate(lines):
            if line.startswith(name[5:]):
                return line, i + 1, i + 1
        raise RuntimeError(f"{name} not found in {path}")
    ctor = name.startswith("ctor:")  # a constructor: `Name(` alone on its line, an initialiser list, then the body between lines that are just braces
    if ctor:
        pat = re.compile(r"^\s*" + re.escape(name[5:]) + r"\($")
    elif name.startswith("decl:"):  # a function with another return type: "decl:uint32_t powi"
        rtype, fname = name[5:].rsplit(" ", 1)
        pat = re.compile(r"\b" + re.escape(rtype) + r"\s+" + re.escape(fname) + r"\b\s*\(")
    else:
        pat = re.compile(r"(\b(void|struct)\s+" + re.escape(name) + r"\b\s*[({])|(\bstruct\s+" + re.escape(name) + r"\s*$)")
    for i, line in enumerate(lines):
        if not pat.search(line) or line.lstrip().startswith("//"):
            continue
        first = i
        if ctor:
            j = i
            while lines[j].strip() != "{":
                j += 1
            depth = 0
            while True:
                code = lines[j].split("//")[0]
                depth += code.count("{") - code.count("}")
                if depth == 0:
                    return "\n".join(lines[first:j + 1]), first + 1, j + 1
                j += 1
        while first > 0 and (lines[first - 1].startswith("template") or lines[first - 1].startswith("__global__") or lines[first - 1].startswith("__device__")
                             or lines[first - 1].startswith("TCNN_")):
            first -= 1
        depth, seen, j = 0, False, i
        while j < len(lines):
            code = lines[j].split("//")[0]
            depth += code.count("{") - code.count("}")
            seen = seen or "{" in code
            if seen and depth == 0:
                last = j
                text = "\n".join(lines[first:last + 1])
                if lines[i].lstrip().startswith("struct") or "struct " + name in lines[i]:
                    text += ";" if not text.rstrip().endswith(";") else ""
                return text, first + 1, last + 1
            j += 1
    raise RuntimeError(f"definition of {name} not found in {path}")


def build(reference="/root/reference", force=False, verbose=True):
    if not os.path.isdir(os.path.join(reference, "include", "tiny-cuda-nn")):
        if verbose:
            print(f"[build_ref] no reference tree at {reference}: keeping whatever {LIB} exists")
        return os.path.exists(LIB)
    srcs = [os.path.join(HERE, "ref_driver.cpp"), os.path.join(HERE, "ref_driver_mlp.cpp"), os.path.join(HERE, "ref_driver_host.cpp"), os.path.join(HERE, "ref_shim", "cuda_fp16.h"),
            os.path.join(HERE, "ref_shim", "mma.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(LIB) and os.path.exists(MANIFEST) and os.path.getmtime(LIB) >= max(os.path.getmtime(s) for s in srcs):
        return True
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest = []

    def gather(kernels, wrap=True):
        parts = []
        for rel, names in kernels:
            path = os.path.join(reference, rel)
            for name in names:
                text, first, last = extract(path, name)
                manifest.append({"file": rel, "name": name, "lines": [first, last]})
                parts.append(f"// ---- {rel}:{first}-{last} ({name})\n{text}\n")
        return "namespace tcnn {\n" + "\n".join(parts) + "\n}  // namespace tcnn\n" if wrap else "\n".join(parts)

    with tempfile.TemporaryDirectory(prefix="tcnn_ref_") as tmp:
        with open(os.path.join(tmp, "ref_extracted_kernels.inc"), "w") as f:
            f.write(gather(KERNELS))
        with open(os.path.join(tmp, "ref_extracted_mlp.inc"), "w") as f:
            f.write(gather(KERNELS_MLP))
        for inc_name, kernels in KERNELS_HOST.items():  # included inside namespace tcnn / inside a class body: no namespace of their own
            with open(os.path.join(tmp, inc_name), "w") as f:
                f.write(gather(kernels, wrap=False))
        flags = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                 "-Wno-keyword-macro", "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-unused-value",
                 "-D__CUDACC__", "-D__CUDA_ARCH__=610", "-DTCNN_HALF_PRECISION=1", "-DTCNN_MIN_GPU_ARCH=61",
                 "-I" + os.path.join(HERE, "ref_shim"), "-I" + tmp, "-I" + os.path.join(reference, "include"), "-I" + os.path.join(reference, "dependencies")]
        # the two translation units compile side by side
        objs = [os.path.join(tmp, "ref_driver.o"), os.path.join(tmp, "ref_driver_mlp.o"), os.path.join(tmp, "ref_driver_host.o")]
        procs = [subprocess.Popen([CLANG, *flags, "-c", os.path.join(HERE, src), "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for src, obj in zip(["ref_driver.cpp", "ref_driver_mlp.cpp", "ref_driver_host.cpp"], objs)]
        for proc in procs:
            _, err = proc.communicate()
            if proc.returncode != 0:
                sys.stderr.write(err[-6000:])
                raise RuntimeError("oracle/_ref: compiling the reference's kernels for the host failed")
        r = subprocess.run([CLANG, "-shared", "-fopenmp", *objs, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-6000:])
            raise RuntimeError("oracle/_ref: linking failed")
    json.dump({"reference": reference, "compiled": manifest,
               "whole_files": ["include/tiny-cuda-nn/common.h", "include/tiny-cuda-nn/vec.h", "include/tiny-cuda-nn/common_device.h", "dependencies/pcg32/pcg32.h"]},
              open(MANIFEST, "w"), indent=1)
    if verbose:
        print(f"[build_ref] {LIB}: {len(manifest)} definitions of the reference compiled for the host")
    return True


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    sys.exit(0 if build(a.reference, a.force) else 1)
