#!/bin/bash
# HISTORICAL (round 3): the SMD_TRANS_SETTLE / SMD_LOAD_SETTLE library variants this script compares were removed in round 4 --
# the instruction-level cause they were probing was refuted by tools/rsq_repro.hip; kept as the record of
# how profiles/r3_det_root_cause.txt was produced.  The engine-free reproducer is tools/build_rsq_repro.sh + tools/rsq_repro.
# Round-3 root-cause matrix for the co-residency non-repeatability.  Run on the GPU box from the repo root:
#   tools/r3_det_root_cause.sh <tag>   -> gpurun_out/<tag>_det.txt
# Libraries: default; _ts2 / _ts8 = SMD_TRANS_SETTLE 2 / 8 (bare v_rsq_f32 + s_nop 1 / 7 before its first consumer in every
# LayerNorm); _tsm1 = bare v_rsq_f32 without extra wait states (code-shape control).
# Knob tn_exclusive_cu: 1 = wgrad workgroups padded to the whole LDS (shipped in round 2), 0 = two-buffer kernels, no pad,
# 2 = the exclusive launch's four-buffer kernels without the pad.
set -u
TAG=${1:-r3a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_det.txt
mkdir -p $R/gpurun_out
: > $OUT
run() {   # suffix, extra env, args...
  local sfx=$1; shift
  local envs=$1; shift
  echo "== lib '${sfx}' env '${envs}' args $*" >> $OUT
  env SMD_LIB_SUFFIX=$sfx $envs ITERS=${ITERS:-40} timeout 300 python $R/tools/det_matrix.py "$@" 2>&1 | grep -v amdgpu.ids >> $OUT
}
run ""      ""                         T:tn_exclusive_cu=1
run ""      ""                         T:tn_exclusive_cu=0
run "_ts8"  ""                         T:tn_exclusive_cu=0
run "_ts2"  ""                         T:tn_exclusive_cu=0
run "_tsm1" ""                         T:tn_exclusive_cu=0
run ""      ""                         T:tn_exclusive_cu=2
run "_ts8"  ""                         T:tn_exclusive_cu=2
run ""      "SMD_SIDE_PRIORITY=normal" T:tn_exclusive_cu=0
cat $OUT
