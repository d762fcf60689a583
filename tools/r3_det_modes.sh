#!/bin/bash
# HISTORICAL (round 3): needs a -DSMD_TN_EXPERIMENTS build for every tn_mode but 48x since round 4 (tools/build_rsq_repro.sh);
# the engine-free form of this matrix is tools/rsq_repro.
# wgrad-kernel variants against the co-residency non-repeatability (tn_mode = NS*100 + NW*10 + pad code), library suffix $1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
SFX=${1:-_tsm1}
for m in 240 242 241 280 440 480; do
  echo "== lib $SFX tn_mode $m"
  SMD_LIB_SUFFIX=$SFX ITERS=${ITERS:-100} python $R/tools/det_matrix.py T:tn_exclusive_cu=0 T:tn_mode=$m 2>&1 | grep "repeats differ"
done
echo "== lib $SFX tn_mode 240 side priority normal"
SMD_SIDE_PRIORITY=normal SMD_LIB_SUFFIX=$SFX ITERS=${ITERS:-100} python $R/tools/det_matrix.py T:tn_exclusive_cu=0 T:tn_mode=240 2>&1 | grep "repeats differ"
