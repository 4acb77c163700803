# round-3 profiles (run on the GPU box from the repo root: bash profiles/scripts/prof_r03.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
rm -rf $R/prof_r03* $R/pmc_r03*
run() { name=$1; shift; timeout -k 10 400 rocprofv3 "$@" > $R/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/$name.log; exit 1; }; echo "$name ok"; }
# 1. kernel stats: the driver's command (headline + every sub-record), the headline alone on one stream, the other configs
run prof_r03_full --kernel-trace --stats -d $R/prof_r03_full -o bench --output-format csv -- python3 $B --steps 20 --warmup 5 --no-cpu
run prof_r03d --kernel-trace --stats -d $R/prof_r03d -o bench --output-format csv -- python3 $B --steps 50 --warmup 5 --no-cpu --no-extra
export TTSK_SINGLE_STREAM=1
run prof_r03s --kernel-trace --stats -d $R/prof_r03s -o bench --output-format csv -- python3 $B --steps 20 --warmup 3 --no-cpu --no-extra --inflight 1
# 2. counters of the headline (separate passes)
run pmc_r03f --kernel-trace --pmc FETCH_SIZE -d $R/pmc_r03f -o p --output-format csv -- python3 $B --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
run pmc_r03w --kernel-trace --pmc WRITE_SIZE -d $R/pmc_r03w -o p --output-format csv -- python3 $B --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
run pmc_r03sq --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/pmc_r03sq -o p --output-format csv -- python3 $B --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
unset TTSK_SINGLE_STREAM
# 3. the other configurations: stats + traffic per sketch (steps + warm-up sketches per run: 5 + 2, + 1 first call at c4)
for c in c2 c4 c5 ref150; do
  run prof_r03_$c --kernel-trace --stats -d $R/prof_r03_$c -o bench --output-format csv -- python3 $B --config $c --steps 10 --warmup 2 --no-cpu
done
for c in c2 c4 ref150; do
  run pmc_r03f_$c --kernel-trace --pmc FETCH_SIZE -d $R/pmc_r03f_$c -o p --output-format csv -- python3 $B --config $c --steps 5 --warmup 2 --no-cpu
  run pmc_r03w_$c --kernel-trace --pmc WRITE_SIZE -d $R/pmc_r03w_$c -o p --output-format csv -- python3 $B --config $c --steps 5 --warmup 2 --no-cpu
done
# 4. the sequential variants at C3: kernel stats and the launch timeline of the last call
for w in orth hmt; do
  run prof_r03_$w --kernel-trace --stats -d $R/prof_r03_$w -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/profiles/scripts/orth_bench.py $w
  tail -1 $R/prof_r03_$w.log
done
cd $GRAFT_REPO_ROOT
for w in orth hmt; do
  python3 profiles/scripts/timeline.py gpurun_out/prof_r03_$w/bench_kernel_trace.csv 13 > gpurun_out/r03_${w}_timeline.txt 2>&1
  cp gpurun_out/prof_r03_$w/bench_kernel_stats.csv gpurun_out/r03_${w}_kernel_stats.csv
done
python3 profiles/by_grid.py gpurun_out/prof_r03d/bench_kernel_trace.csv gpurun_out/r03_kernel_by_grid.csv > gpurun_out/r03_by_grid.txt 2>&1
python3 profiles/by_grid.py gpurun_out/prof_r03s/bench_kernel_trace.csv gpurun_out/r03_kernel_by_grid_single_stream.csv > gpurun_out/r03_by_grid_single.txt 2>&1
rm -f gpurun_out/r03_traffic.json
python3 profiles/collect_traffic.py gpurun_out/pmc_r03f gpurun_out/pmc_r03w gpurun_out/r03_traffic.json > gpurun_out/r03_traffic.txt 2>&1
python3 profiles/collect_traffic.py gpurun_out/pmc_r03f_c2 gpurun_out/pmc_r03w_c2 gpurun_out/r03_traffic.json --total c2_sketch 7 >> gpurun_out/r03_traffic.txt 2>&1
python3 profiles/collect_traffic.py gpurun_out/pmc_r03f_c4 gpurun_out/pmc_r03w_c4 gpurun_out/r03_traffic.json --total c4_sketch 11 --only 'sg_pass|sg_psi_reduce|sg_om_reduce|fillBuffer' >> gpurun_out/r03_traffic.txt 2>&1
python3 profiles/collect_traffic.py gpurun_out/pmc_r03f_ref150 gpurun_out/pmc_r03w_ref150 gpurun_out/r03_traffic.json >> gpurun_out/r03_traffic.txt 2>&1
python3 profiles/sq_counters.py gpurun_out/pmc_r03sq/p_counter_collection.csv > gpurun_out/r03_sq_counters.txt 2>&1
cp gpurun_out/prof_r03_full/bench_kernel_stats.csv gpurun_out/r03_bench_full_kernel_stats.csv
cp gpurun_out/prof_r03d/bench_kernel_stats.csv gpurun_out/r03_bench_kernel_stats.csv
for c in c2 c4 c5 ref150; do cp gpurun_out/prof_r03_$c/bench_kernel_stats.csv gpurun_out/r03_${c}_kernel_stats.csv; done
# keep the merge small: traces and databases stay on the box
find gpurun_out/prof_r03* gpurun_out/pmc_r03* -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_r03* gpurun_out/pmc_r03* -name "*trace.csv" -delete 2>/dev/null
find gpurun_out/pmc_r03* -name "*counter_collection.csv" -delete 2>/dev/null
tail -20 gpurun_out/r03_traffic.txt; head -14 gpurun_out/r03_by_grid_single.txt
