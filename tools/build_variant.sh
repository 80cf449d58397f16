#!/bin/bash
# Build ONE lane-program instance of the env library (full C-ABI) into robot_lab_amd/csrc/variants/<name>_<inst>.so for the
# one-call A/B harness (tools/ab_bench.py, RL_ENV_LIB):   tools/build_variant.sh <name> <inst: 34|44|1044|74> [git rev | -] [extra hipcc flags]
# rev "-" (default) = the working tree; a git rev builds that revision's csrc/ + include/ from a scratch checkout.
NAME=$1; INST=$2; REV=${3:--}; shift; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/robot_lab_amd/csrc/variants
mkdir -p $OUT
SRC=$ROOT
if [ "$REV" != "-" ]; then
  SRC=$(mktemp -d /tmp/variant_src.XXXXXX)
  git -C $ROOT archive $REV robot_lab_amd/csrc include | tar -x -C $SRC
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-remove-redundant-endcf=false -std=c++17 -shared -fPIC -DRL_ENV_SINGLE_TU -DRL_ENV_ONLY=$INST "$@" \
  -o $OUT/${NAME}_${INST}.so $SRC/robot_lab_amd/csrc/rl_env.hip && echo "built $OUT/${NAME}_${INST}.so from $REV $*"
[ "$REV" != "-" ] && rm -rf $SRC
