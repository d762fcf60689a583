#!/bin/bash
# Builds csrc/libsmd_hip_old.so from the library sources of an earlier commit (default HEAD~1), for the two-library A/B tools
# (lds_swizzle_ab.py, reverse_step_ab.py, ln_bwd_ab.py): both builds are loaded into one process and timed interleaved.
#   tools/build_old_lib.sh [commit]
set -eu
C=${1:-HEAD~1}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$R" archive "$C" symbolic-music-diffusion_amd include | tar -x -C "$T"
python - "$T" <<'PY'
import importlib.util, sys
spec = importlib.util.spec_from_file_location("oldbuild", sys.argv[1] + "/symbolic-music-diffusion_amd/build.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
m.build_library(force=True)
PY
cp "$T/symbolic-music-diffusion_amd/csrc/libsmd_hip.so" "$R/symbolic-music-diffusion_amd/csrc/libsmd_hip_old.so"
rm -rf "$T"
echo "built $R/symbolic-music-diffusion_amd/csrc/libsmd_hip_old.so from $C"
