# per-kernel durations of the device-resident step loop on small meshes (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for N in "$@"; do
  rm -rf $R/gpurun_out/prof_small_$N
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_small_$N -o p -- python3 $R/tools/small_mesh_profile_target.py $N > $R/gpurun_out/prof_small_$N.log 2>&1 || exit 1
  grep steps_per_s $R/gpurun_out/prof_small_$N.log
  python3 $R/tools/rocpd_stats.py $R/gpurun_out/prof_small_$N/p_results.db | cut -c1-100,180-230 | head -8
done
