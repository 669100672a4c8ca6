#!/bin/bash
# Runs on the GPU box: everything profiles/TAG_* is made from.  Two gpurun calls (a call is limited to 20 minutes):
#   gpurun -- 'bash tools/refresh_all.sh TAG attncut'   the default bench line; per precision mode the rocprofv3 kernel stats, per-call traces,
#                                                       PMC traffic and SQ counters of the AttnCut step (BASELINE configs[1])
#   gpurun -- 'bash tools/refresh_all.sh TAG rest'      the same for the Choopy step in the default mode (configs[2]); the side configurations
# then `python tools/collect_profiles.py TAG` copies the summaries into profiles/.
TAG=${1:-r06}
PART=${2:-attncut}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
if [ "$PART" = "attncut" ]; then
  bash $R/tools/refresh_profiles.sh $TAG nochoopy > $O/${TAG}_refresh.log 2>&1
  echo "attncut modes done"; tail -2 $O/${TAG}_refresh.log
  for MODE in bf16x6 fp32 bf16x3; do
    PMC_SQ_ARGS="--precision $MODE" bash $R/tools/pmc_sq_step.sh ${TAG}_$MODE > /dev/null 2>&1
    cp $O/${TAG}_${MODE}_pmc_sq.txt $O/${TAG}_pmc_sq_$MODE.txt 2>/dev/null
    echo "sq $MODE done"
  done
else
  bash $R/tools/choopy_profile.sh $TAG > $O/${TAG}_choopy_refresh.log 2>&1
  echo "choopy done"; tail -3 $O/${TAG}_choopy_refresh.log
  bash $R/tools/side_configs.sh $TAG bf16x6 > $O/${TAG}_side_refresh.log 2>&1
  echo "side configurations done"; tail -14 $O/${TAG}_side_configs.txt
fi
