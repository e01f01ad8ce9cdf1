# What differs between a context whose device memory was never handed out before (slower predictor launch)
# and the next context of the same process (memory freed and allocated again): one rocprofv3 counter pass per
# run over tools/predict_time_series.py (two contexts one after the other, priming off), summarised per
# context by tools/placement_pmc_summary.py.  Run on the GPU box from the repository root.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/placement
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TEDD_PRIME_MEMORY=0
i=0
while read -r pass; do
  [ -z "$pass" ] && continue
  i=$((i+1))
  rm -rf $OUT/p$i
  timeout -s KILL 170 rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/predict_time_series.py 16384 3 2 > $OUT/p$i.log 2>&1 || { echo "pass $i ($pass) failed"; tail -5 $OUT/p$i.log; exit 1; }
  echo "pass $i done: $pass"
done <<PASSES
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_EA_BUSY GRBM_TC_BUSY
TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUBBLE_sum
TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum
PASSES
cd $R
python3 tools/placement_pmc_summary.py $OUT/summary.json $(find $OUT -name '*counter_collection.csv' | sort)
