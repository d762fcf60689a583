#!/bin/bash
# Builds what tools/rsq_repro needs next to the shipped library (run here; the .so files travel to the GPU box):
#   libsmd_hip_tnx.so   the experiment instantiations of the 128-wide weight-gradient kernel (aggressors <2,4>, <2,8>, <4,4>)
#   libsmd_hip_tsm1.so  the same + the BARE v_rsq_f32 in every LayerNorm (victim form 3), shipped code generation (no packed fp32)
#   libsmd_hip_slpbare.so / _slp.so  round 3's code generation (SMD_SLP=1) with the bare / the guarded instruction (forms 13 / 14)
#   tools/rsq_repro
set -e
cd "$(dirname "$0")/.."
SMD_EXTRA_DEFS="-DSMD_TN_EXPERIMENTS" SMD_LIB_SUFFIX=_tnx python -m smd_amd.build
SMD_EXTRA_DEFS="-DSMD_TN_EXPERIMENTS -DSMD_LN_RSTD_BARE" SMD_LIB_SUFFIX=_tsm1 python -m smd_amd.build
# the round-3 code generation (packed-fp32 arithmetic everywhere) + the bare instruction: the victim that fails
SMD_SLP=1 SMD_EXTRA_DEFS="-DSMD_TN_EXPERIMENTS -DSMD_LN_RSTD_BARE" SMD_LIB_SUFFIX=_slpbare python -m smd_amd.build
SMD_SLP=1 SMD_LIB_SUFFIX=_slp python -m smd_amd.build
python -m smd_amd.build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/rsq_repro tools/rsq_repro.hip -ldl
