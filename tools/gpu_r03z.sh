#!/bin/bash
# Round 3: where does the G1 step kernel wait?  PC sampling (rocprofv3 beta) of the phase-marked build; stochastic first, host-trap as fallback
OUT=$PWD/gpurun_out/r03z
mkdir -p $OUT
export TMPDIR=/tmp
export RL_ENV_LIB=$PWD/robot_lab_amd/csrc/variants/g1_marks_74.so
cd /tmp
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv -d $OUT/stoch -- python $GRAFT_REPO_ROOT/tools/pcs_run.py > $OUT/stoch.log 2>&1
echo "stochastic rc=$?"; tail -3 $OUT/stoch.log | cut -c1-300
find $OUT/stoch -type f | head; du -sh $OUT/stoch
if ! find $OUT/stoch -name "*pc_sampling*" | grep -q .; then
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --output-format csv -d $OUT/trap -- python $GRAFT_REPO_ROOT/tools/pcs_run.py > $OUT/trap.log 2>&1
  echo "host_trap rc=$?"; tail -3 $OUT/trap.log | cut -c1-300
  find $OUT/trap -type f | head; du -sh $OUT/trap
fi
# keep the merge small: the sample CSVs compress well
for f in $(find $OUT -name "*.csv" -size +8M); do gzip -9 $f; done
du -sh $OUT
