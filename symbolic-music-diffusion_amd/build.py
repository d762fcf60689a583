"""Builds csrc/*.hip into the in-tree C-ABI shared library ``csrc/libsmd_hip.so`` for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the repo snapshot
to the GPU box.  Usage: ``python -m smd_amd.build`` or ``build_library()``.
"""
from __future__ import annotations

import hashlib
import os
import re
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
           os.path.join("..", "..", "include", "smd_hip.h"), os.path.join("..", "..", "include", "smd_hip_lab.h")]
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


# ---- build identity: a hash of what the library is built FROM (sources, headers, flags), embedded in the .so as the string
# "SMD_BUILD_ID=<16 hex>" and returned by smd_build_id().  Staleness is decided by comparing that id with the id of the tree --
# never by file times, which an rsync / clone / snapshot can put in any order.
_ID_RE = re.compile(rb"SMD_BUILD_ID=([0-9a-f]{16})")


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def _headers_digest() -> bytes:
    h = hashlib.sha256()
    for name in HEADERS:
        h.update(name.encode() + b"\0" + _read(os.path.join(CSRC, name)))
    return h.digest()


def _object_id(src: str, hdr: bytes) -> str:
    h = hashlib.sha256()
    h.update(hdr + _read(os.path.join(CSRC, src)) + " ".join(FLAGS + EXTRA_FLAGS.get(src, [])).encode())
    return h.hexdigest()[:12]


def source_id() -> str:
    """The id a library built from the current tree (and the current SMD_* build environment) carries."""
    hdr = _headers_digest()
    h = hashlib.sha256()
    for src in SOURCES:
        h.update(src.encode() + _object_id(src, hdr).encode())
    return h.hexdigest()[:16]


def built_id(path: str = None) -> str:
    """The id embedded in an existing library ('' when the file is missing or carries none), read without loading it."""
    path = LIB_PATH if path is None else path
    if not os.path.exists(path):
        return ""
    m = _ID_RE.search(_read(path))
    return m.group(1).decode() if m else ""


def is_stale() -> bool:
    return built_id() != source_id()


def build_library(force: bool = False, verbose: bool = True) -> str:
    """Compiles what is missing and links; returns the library path.  ``build_library.last`` says what happened:
    'reused' (the library's embedded id matches the tree), 'linked' (objects reused, library re-linked) or 'compiled N'."""
    sid = source_id()
    if not force and built_id() == sid:
        build_library.last = "reused"
        if verbose:
            print(f"[smd_amd.build] {LIB_PATH} is current (build id {sid}): reused", file=sys.stderr)
        return LIB_PATH
    hipcc = find_hipcc()
    objdir = os.path.join(CSRC, "build" + _SUFFIX)
    os.makedirs(objdir, exist_ok=True)
    hdr = _headers_digest()
    compiled = []

    def compile_one(src: str) -> str:
        stem = src.replace(".hip", "")
        obj = os.path.join(objdir, f"{stem}.{_object_id(src, hdr)}.o")       # content-addressed: no file times involved
        if not force and os.path.exists(obj):
            return obj
        for old in os.listdir(objdir):
            if old.startswith(stem + ".") and old.endswith(".o"):
                os.remove(os.path.join(objdir, old))
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        os.replace(obj + ".tmp", obj)
        compiled.append(src)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    idsrc = os.path.join(objdir, "build_id.cpp")
    with open(idsrc, "w") as f:
        f.write('extern "C" const char* smd_build_id(void) { static const char id[] = "SMD_BUILD_ID=%s"; return id + 13; }\n' % sid)
    idobj = os.path.join(objdir, "build_id.o")
    r = subprocess.run([hipcc, "-O1", "-fPIC", "-c", idsrc, "-o", idobj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on build_id.cpp:\n{r.stdout}\n{r.stderr}")
    tmp = LIB_PATH + ".tmp"
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs, idobj],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, LIB_PATH)
    build_library.last = f"compiled {len(compiled)}" if compiled else "linked"
    if verbose:
        print(f"[smd_amd.build] built {LIB_PATH} (build id {sid}; {len(compiled)} of {len(SOURCES)} translation units compiled: "
              f"{', '.join(compiled) or 'none'})", file=sys.stderr)
    return LIB_PATH


build_library.last = ""


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
