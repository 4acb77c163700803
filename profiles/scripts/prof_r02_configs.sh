# rocprofv3 kernel stats of the other BASELINE configurations (bash profiles/scripts/prof_r02_configs.sh on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
for c in c2 c4 c5; do
  rm -rf $R/prof_r02_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_r02_$c -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --steps 10 --warmup 2 --no-cpu > $R/prof_r02_$c.log 2>&1 || { echo "$c failed"; exit 1; }
  cp $R/prof_r02_$c/bench_kernel_stats.csv $R/r02_${c}_kernel_stats.csv
  echo "$c ok"; head -6 $R/r02_${c}_kernel_stats.csv | cut -c1-150
done
