#!/usr/bin/env python3
"""Builds samples/_generated/mlp_learning_an_image.hip: this repository's template (samples/mlp_learning_an_image.hip.in -- image IO, the
bilinear lookup that stands where the CUDA texture stands, the PSNR report) with the REFERENCE'S OWN training loop spliced in, read from
/root/reference/samples/mlp_learning_an_image.cu where it lies.  The point of the sample is that the reference's caller code compiles and
runs unchanged against the facade headers (include/tiny-cuda-nn/*.h); the loop is therefore taken from the reference at build time, as
oracle/build_ref.py takes the reference's kernels -- never committed.  Edits applied to the extracted text, all mechanical:
  * `cuda` -> `hip`, `CUDA_CHECK_THROW` -> `HIP_CHECK_THROW` (the runtime's names);
  * the default number of training steps 10000000 -> 1000 (the sample terminates and reports a PSNR);
  * the blocks that print the JIT notice and write a JPEG every few steps (fmt + stbi, third-party loaders that are not in this tree) are dropped.
Without /root/reference (the GPU box) nothing is generated: the prebuilt binary travels with the snapshot."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.path.join(os.environ.get("REFERENCE_DIR", "/root/reference"), "samples", "mlp_learning_an_image.cu")
TEMPLATE = os.path.join(ROOT, "samples", "mlp_learning_an_image.hip.in")
OUT_DIR = os.path.join(ROOT, "samples", "_generated")
OUT = os.path.join(OUT_DIR, "mlp_learning_an_image.hip")


def drop_block(lines, opener):
    """Removes the statement that starts on the line containing `opener` through its closing brace (incl. `} else { ... }`)."""
    out, i = [], 0
    while i < len(lines):
        if opener in lines[i]:
            depth, seen = 0, False
            while i < len(lines):
                depth += lines[i].count("{") - lines[i].count("}")
                seen = seen or "{" in lines[i]
                i += 1
                if seen and depth == 0 and not (i < len(lines) and lines[i].strip().startswith("else")):
                    break
            while out and out[-1].strip() == "" and i < len(lines) and lines[i].strip() == "":
                i += 1
            continue
        out.append(lines[i])
        i += 1
    return out


def reference_loop():
    text = open(REFERENCE, encoding="utf-8").read().split("\n")
    a = next(i for i, l in enumerate(text) if "Fourth step: train the model" in l)
    b = next(i for i, l in enumerate(text) if "Dump final image if a name was specified" in l)
    lines = text[a:b]
    lines = drop_block(lines, "if (network->jit_fusion())")
    lines = drop_block(lines, "if (visualize_learned_func)")
    lines = [l for l in lines if "bool visualize_learned_func" not in l]
    body = "\n".join(lines)
    body = body.replace("CUDA_CHECK_THROW", "HIP_CHECK_THROW")
    body = re.sub(r"\bcuda(?=[A-Z])", "hip", body)
    body = body.replace("10000000", "1000")
    return body.rstrip("\n") + "\n"


def generate():
    if not os.path.exists(REFERENCE):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    text = open(TEMPLATE, encoding="utf-8").read().replace("@REFERENCE_TRAINING_LOOP@\n", reference_loop())
    if not os.path.exists(OUT) or open(OUT, encoding="utf-8").read() != text:
        open(OUT, "w", encoding="utf-8").write(text)
    return OUT


if __name__ == "__main__":
    p = generate()
    print(p if p else f"{REFERENCE} not found: nothing generated (a binary built earlier, e.g. the one that travelled to the GPU box, keeps working)")
    sys.exit(0)
