# The round's profiles: default bench line, rocprofv3 kernel statistics of the bench (16384^2) and of the
# N = 256 study-mesh loop.  Run on the GPU box from the repository root; results under gpurun_out/round/.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/round
mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py > $OUT/bench_16384.json 2> $OUT/bench_16384.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof16384 -o p -- python3 $R/bench.py --no-extras > $OUT/prof16384.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof256 -o p -- python3 $R/tools/study_mesh_profile_target.py > $OUT/prof256.log 2>&1
find $OUT -name '*kernel_stats.csv'
