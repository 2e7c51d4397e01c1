#!/bin/bash
# A/B builds of libcusrl_hip.so that differ in compile-time knobs of ppo_loss.hip (build/variants/, git-ignored, shipped to the
# GPU box with the snapshot): run one with CUSRL_HIP_LIBRARY=build/variants/libcusrl_hip_<tag>.so python scripts/kernel_bench.py ...
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/build/variants; mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I$R/include -I$R/cusrl_amd/csrc"
OBJS=$(ls $R/build/obj/*.o | grep -v ppo_loss.o)
build() { tag=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $R/cusrl_amd/csrc/ppo_loss.hip -o $OUT/ppo_loss_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/ppo_loss_$tag.o -ldl -o $OUT/libcusrl_hip_$tag.so
  rm -f $OUT/ppo_loss_$tag.o; echo built $tag; }
build base &
build r2 -DCUSRL_LOSS_ROUNDS_CAP=2 &
build f32 -DCUSRL_LOSS_F32_WAVE_SUMS &
wait
build r2f32 -DCUSRL_LOSS_ROUNDS_CAP=2 -DCUSRL_LOSS_F32_WAVE_SUMS &
build r2f32wpe7 -DCUSRL_LOSS_ROUNDS_CAP=2 -DCUSRL_LOSS_F32_WAVE_SUMS -DCUSRL_LOSS_WAVES_PER_EU=7 &
build r2f32wpe6 -DCUSRL_LOSS_ROUNDS_CAP=2 -DCUSRL_LOSS_F32_WAVE_SUMS -DCUSRL_LOSS_WAVES_PER_EU=6 &
wait
ls -la $OUT
