"""Build the gfx950 kernels into ``vqattack_amd/lib/libvqattack_hip.so`` (C ABI: include/vqattack_hip.h).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to the GPU
box with the tree (it is git-ignored, not gpurun-ignored).  ``python -m vqattack_amd.build [--force]``.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvqattack_hip.so")
# the same sources with -DVQA_TUNING: launch-shape knobs behind vqa_set_option() + the A/B kernel variants (tools/ only)
TUNING_LIB_PATH = os.path.join(LIB_DIR, "libvqattack_hip_tuning.so")
SOURCES = ["linf.hip", "lnorm.hip", "loss.hip", "ce.hip", "text.hip", "image.hip", "attn.hip", "block.hip"]
# -ffp-contract=off: the reference's op chain rounds after every add/mul; keep it that way (bit-exact parity).
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


# which source file a kernel of the library lives in (by a substring of its demangled name) -- for records that must be
# tied to the code that produced them (profiles/*/pmc_traffic.json, bench.py's `traffic` field)
KERNEL_SOURCES = (("stream4_kernel", "linf.hip"), ("neg_cos_rows", "loss.hip"), ("ce_rows", "ce.hip"), ("ce_count", "ce.hip"),
                  ("attn_", "attn.hip"), ("sumsq", "lnorm.hip"), ("absmax", "lnorm.hip"), ("per_sample", "lnorm.hip"),
                  ("sum_stage2", "lnorm.hip"), ("ln_fwd", "block.hip"), ("ln_bwd", "block.hip"), ("gelu", "block.hip"),
                  ("resize", "image.hip"), ("gather_rows", "text.hip"), ("cand_dir_sim", "text.hip"),
                  ("embed_tokens", "text.hip"), ("greedy_accept", "text.hip"))


def kernel_source_digest(kernel_name):
    """sha256 over the source file a kernel is compiled from + ``common.hpp`` + the C-ABI header + the compiler flags:
    changes whenever that kernel's code object can have changed.  None for a kernel this table does not know."""
    import hashlib
    src = next((f for key, f in KERNEL_SOURCES if key in kernel_name), None)
    if src is None:
        return None
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in (os.path.join(SRC_DIR, src), os.path.join(SRC_DIR, "common.hpp"),
                 os.path.join(HERE, "..", "include", "vqattack_hip.h")):
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP kernels of vqattack_amd cannot be built")


def _stale(path=LIB_PATH):
    if not os.path.exists(path):
        return True
    built = os.path.getmtime(path)
    deps = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR)]
    deps.append(os.path.join(HERE, "..", "include", "vqattack_hip.h"))
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False, tuning=False):
    """``tuning=True`` builds ``libvqattack_hip_tuning.so`` (select it with VQA_TUNING_LIB=1) instead of the product."""
    target = TUNING_LIB_PATH if tuning else LIB_PATH
    if not force and not _stale(target):
        return target
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc()] + FLAGS + (["-DVQA_TUNING"] if tuning else []) + \
        [os.path.join(SRC_DIR, s) for s in SOURCES] + ["-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    if verbose and proc.stderr.strip():
        print(proc.stderr)
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    if "--digest" in sys.argv:                       # python -m vqattack_amd.build --digest <kernel name>
        print(kernel_source_digest(sys.argv[sys.argv.index("--digest") + 1]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv))
