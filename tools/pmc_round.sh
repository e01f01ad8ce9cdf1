# Counter passes of the bench configuration (16384^2), one rocprofv3 run per pass (the passes cannot share a
# run), summarised by tools/pmc_summary.py.  Run on the GPU box from the repository root; TEDD_GIT_REV (the
# revision of the tree that was sent, the box has no .git) goes into the summary's stamp beside the library's hash.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $OUT/$tag
  timeout -s KILL 280 rocprofv3 --pmc $pass --output-format csv -d $OUT/$tag -o p -- python3 $R/bench.py --no-extras --steps 4 --warmup 1 --settle 6 > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; exit 1; }
  echo "pass $tag done"
done
cd $R
python3 tools/pmc_summary.py $OUT/summary.json 16384 strip 10 "arrays per time level" $(find $OUT -name '*counter_collection.csv')
ls -la $OUT/summary.json
