#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/refresh_all.sh TAG'): everything profiles/TAG_* is made from - the default bench line, per precision mode the rocprofv3 kernel stats /
# per-call traces / PMC traffic / SQ counters of the AttnCut step (BASELINE configs[1]), the same for the Choopy step in the default mode
# (configs[2]), and the side configurations.  tools/collect_profiles.py TAG then copies the summaries into profiles/.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/refresh_profiles.sh $TAG nochoopy > $O/${TAG}_refresh.log 2>&1
echo "attncut modes done"; tail -2 $O/${TAG}_refresh.log
for MODE in bf16x6 fp32 bf16x3; do
  PMC_SQ_ARGS="--precision $MODE" bash $R/tools/pmc_sq_step.sh ${TAG}_$MODE > /dev/null 2>&1
  cp $O/${TAG}_${MODE}_pmc_sq.txt $O/${TAG}_pmc_sq_$MODE.txt 2>/dev/null
done
echo "sq done"
bash $R/tools/choopy_profile.sh $TAG > $O/${TAG}_choopy_refresh.log 2>&1
echo "choopy done"; tail -3 $O/${TAG}_choopy_refresh.log
bash $R/tools/side_configs.sh $TAG bf16x6 > $O/${TAG}_side_refresh.log 2>&1
echo "side configurations done"; tail -14 $O/${TAG}_side_configs.txt
