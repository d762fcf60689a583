"""Builds csrc/*.hip into the in-tree C-ABI shared library ``csrc/libsmd_hip.so`` for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the repo snapshot
to the GPU box.  Usage: ``python -m smd_amd.build`` or ``build_library()``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsmd_hip.so")
SOURCES = ["gemm_nt.hip", "gemm_nt256.hip", "gemm_tn.hip", "gemm_tn256.hip", "norm.hip", "ln128.hip", "attention.hip", "encoder_fused.hip", "diffusion.hip", "rng_jax.hip", "optim.hip",
           "engine.hip", "capi.hip"]
HEADERS = ["smd_common.h", "smd_kernels.h", "gemm_epilogue.h", "engine.h", "rng.h", "rng_threefry.h",
           os.path.join("..", "..", "include", "smd_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]
# SMD_ABLATIONS=1 in the environment: also compile the kernel-ablation variants (wrong results by construction; only
# tools/kbench.py --gemm-ab uses them).  The shipped library never contains them.
if os.environ.get("SMD_ABLATIONS") == "1":
    FLAGS.append("-DSMD_ABLATIONS")
# experiment builds: SMD_EXTRA_DEFS="-DX=1 -DY" adds preprocessor definitions, SMD_LIB_SUFFIX="_tag" names the output
# csrc/libsmd_hip_tag.so (objects under csrc/build_tag/); lib.py loads that library when SMD_LIB_SUFFIX is set
FLAGS += [f for f in os.environ.get("SMD_EXTRA_DEFS", "").split() if f]
_SUFFIX = os.environ.get("SMD_LIB_SUFFIX", "")
LIB_PATH = os.path.join(CSRC, f"libsmd_hip{_SUFFIX}.so")


# per-file code generation switches
EXTRA_FLAGS = {"encoder_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # jax.random streams: every mul/add rounds separately (HIP's __fmul_rn/__fadd_rn are plain operators)
               "rng_jax.hip": ["-ffp-contract=off"]}
# NO PACKED-FP32 VALU ARITHMETIC in the small-LDS kernels -- LayerNorms, elementwise, optimiser (SMD_SLP=1 in the environment
# restores hipcc's default for A/B runs).  tools/rsq_repro.hip: a v_pk_mul_f32 that the SLP vectoriser
# forms on a register pair reads a STALE source in lanes 48..63 when the VALU instruction that wrote that register is two
# instructions ahead of it and a two-buffer weight-gradient workgroup shares the CU -- with a transcendental or a plain FMA as
# the producer alike; the same statement as two v_mul_f32 is never wrong.  Per-kernel times are unchanged (profiles/r4k_*).
# encoder_fused.hip keeps the default: its workgroups take 136-160 KiB of LDS, no GEMM workgroup can join them on a CU, and
# its GELU phases are 3 % faster with the packed forms.
if os.environ.get("SMD_SLP") != "1":
    for _f in ("norm.hip", "ln128.hip", "diffusion.hip", "optim.hip", "attention.hip"):
        EXTRA_FLAGS.setdefault(_f, []).append("-fno-slp-vectorize")


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _newest_input() -> float:
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return max(os.path.getmtime(f) for f in files)


def is_stale() -> bool:
    return not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < _newest_input()


def build_library(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB_PATH
    hipcc = find_hipcc()
    objdir = os.path.join(CSRC, "build" + _SUFFIX)
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), hdr_time):
            return obj
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", srcp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB_PATH + ".tmp"
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB_PATH)
    if verbose:
        print(f"[smd_amd.build] built {LIB_PATH}", file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
