# round-4 profiles (run on the GPU box from the repo root: bash profiles/scripts/prof_r04.sh [part]); parts keep one gpurun call short
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
PART=${1:-all}
run() { name=$1; shift; rm -rf $R/$name; timeout -k 10 500 rocprofv3 "$@" > $R/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/$name.log; exit 1; }; echo "$name ok"; }
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ_B="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"
SQ_C="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS"
if [ $PART = all ] || [ $PART = stats ]; then
  # 1. kernel stats: the driver's command (headline + every sub-record), the headline alone on one stream, the other configs
  run prof_r04_full --kernel-trace --stats -d $R/prof_r04_full -o bench --output-format csv -- python3 $B --steps 20 --warmup 5 --no-cpu
  run prof_r04d --kernel-trace --stats -d $R/prof_r04d -o bench --output-format csv -- python3 $B --steps 50 --warmup 5 --no-cpu --no-extra
  export TTSK_SINGLE_STREAM=1
  run prof_r04s --kernel-trace --stats -d $R/prof_r04s -o bench --output-format csv -- python3 $B --steps 20 --warmup 3 --no-cpu --no-extra --inflight 1
  unset TTSK_SINGLE_STREAM
  for c in c2 c2g c4 c5; do
    run prof_r04_$c --kernel-trace --stats -d $R/prof_r04_$c -o bench --output-format csv -- python3 $B --config $c --steps 10 --warmup 2 --no-cpu
  done
  # C5 with ONE sketch in flight: the launch sequence of a single sketch (the default record has two in flight)
  export TTSK_BENCH_C5_INFLIGHT=1
  run prof_r04_c5one --kernel-trace -d $R/prof_r04_c5one -o bench --output-format csv -- python3 $B --config c5 --steps 10 --warmup 2 --no-cpu
  unset TTSK_BENCH_C5_INFLIGHT
fi
if [ $PART = all ] || [ $PART = pmc1 ]; then
  # 2. counters of the headline (separate passes): traffic, SQ tables incl. the wait attribution
  export TTSK_SINGLE_STREAM=1
  run pmc_r04f --kernel-trace --pmc FETCH_SIZE -d $R/pmc_r04f -o p --output-format csv -- python3 $B --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
  run pmc_r04w --kernel-trace --pmc WRITE_SIZE -d $R/pmc_r04w -o p --output-format csv -- python3 $B --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
  run pmc_r04sq --kernel-trace --pmc $SQ_A -d $R/pmc_r04sq -o p --output-format csv -- python3 $B --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
  run pmc_r04sqb --kernel-trace --pmc $SQ_B -d $R/pmc_r04sqb -o p --output-format csv -- python3 $B --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
  unset TTSK_SINGLE_STREAM
fi
if [ $PART = all ] || [ $PART = pmc2 ]; then
  # 3. the other configurations: traffic per sketch (steps + warm-up sketches per run: 5 + 2, + 1 first call at c4; c5: + 18 drained
  # and 20 back-to-back public calls = 45) and SQ tables of their kernels
  for c in c2 c2g c4 c5; do
    run pmc_r04f_$c --kernel-trace --pmc FETCH_SIZE -d $R/pmc_r04f_$c -o p --output-format csv -- python3 $B --config $c --steps 5 --warmup 2 --no-cpu
    run pmc_r04w_$c --kernel-trace --pmc WRITE_SIZE -d $R/pmc_r04w_$c -o p --output-format csv -- python3 $B --config $c --steps 5 --warmup 2 --no-cpu
  done
  for c in c2g c4 c5; do
    run pmc_r04sq_$c --kernel-trace --pmc $SQ_A -d $R/pmc_r04sq_$c -o p --output-format csv -- python3 $B --config $c --steps 3 --warmup 1 --no-cpu
    run pmc_r04sqb_$c --kernel-trace --pmc $SQ_B -d $R/pmc_r04sqb_$c -o p --output-format csv -- python3 $B --config $c --steps 3 --warmup 1 --no-cpu
  done
fi
cd $GRAFT_REPO_ROOT
if [ $PART = all ] || [ $PART = stats ]; then
  python3 profiles/by_grid.py gpurun_out/prof_r04d/bench_kernel_trace.csv gpurun_out/r04_kernel_by_grid.csv > gpurun_out/r04_by_grid.txt 2>&1
  python3 profiles/by_grid.py gpurun_out/prof_r04s/bench_kernel_trace.csv gpurun_out/r04_kernel_by_grid_single_stream.csv > gpurun_out/r04_by_grid_single.txt 2>&1
  python3 profiles/scripts/timeline.py gpurun_out/prof_r04_c5/bench_kernel_trace.csv 12 > gpurun_out/r04_c5_timeline.txt 2>&1
  python3 profiles/scripts/timeline.py gpurun_out/prof_r04_c5one/bench_kernel_trace.csv 12 > gpurun_out/r04_c5_timeline_single.txt 2>&1
  python3 profiles/scripts/timeline.py gpurun_out/prof_r04_c2g/bench_kernel_trace.csv 12 > gpurun_out/r04_c2g_timeline.txt 2>&1
  cp gpurun_out/prof_r04_full/bench_kernel_stats.csv gpurun_out/r04_bench_full_kernel_stats.csv
  cp gpurun_out/prof_r04d/bench_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv
  for c in c2 c2g c4 c5; do cp gpurun_out/prof_r04_$c/bench_kernel_stats.csv gpurun_out/r04_${c}_kernel_stats.csv; done
fi
if [ $PART = all ] || [ $PART = pmc1 ]; then
  rm -f gpurun_out/r04_traffic.json
  python3 profiles/collect_traffic.py gpurun_out/pmc_r04f gpurun_out/pmc_r04w gpurun_out/r04_traffic.json > gpurun_out/r04_traffic.txt 2>&1
  python3 profiles/sq_counters.py gpurun_out/pmc_r04sq/p_counter_collection.csv > gpurun_out/r04_sq_counters.txt 2>&1
  python3 profiles/sq_counters.py gpurun_out/pmc_r04sqb/p_counter_collection.csv >> gpurun_out/r04_sq_counters.txt 2>&1
fi
if [ $PART = all ] || [ $PART = pmc2 ]; then
  python3 profiles/collect_traffic.py gpurun_out/pmc_r04f_c2 gpurun_out/pmc_r04w_c2 gpurun_out/r04_traffic.json --total c2_sketch 7 >> gpurun_out/r04_traffic.txt 2>&1
  python3 profiles/collect_traffic.py gpurun_out/pmc_r04f_c2g gpurun_out/pmc_r04w_c2g gpurun_out/r04_traffic.json --total c2_gaussian_sketch 7 --only 'dense_left_pass|rows_longk|skinny_r|skinny_s|copy_strided|gemm_f64|small_gemm' >> gpurun_out/r04_traffic.txt 2>&1
  python3 profiles/collect_traffic.py gpurun_out/pmc_r04f_c4 gpurun_out/pmc_r04w_c4 gpurun_out/r04_traffic.json --total c4_sketch 11 --only 'sg_pass|sg_psi_reduce|sg_om_reduce|fillBuffer' >> gpurun_out/r04_traffic.txt 2>&1
  python3 profiles/collect_traffic.py gpurun_out/pmc_r04f_c5 gpurun_out/pmc_r04w_c5 gpurun_out/r04_traffic.json --total c5_sketch 45 --only 'chain_sum|skinny_r_reduce|stream_small|sum_slices|small_gemm|gemm_f64' >> gpurun_out/r04_traffic.txt 2>&1
  for c in c2g c4 c5; do
    python3 profiles/sq_counters.py gpurun_out/pmc_r04sq_$c/p_counter_collection.csv "bench.py --config $c --steps 3 --warmup 1 --no-cpu" > gpurun_out/r04_sq_counters_$c.txt 2>&1
    python3 profiles/sq_counters.py gpurun_out/pmc_r04sqb_$c/p_counter_collection.csv "bench.py --config $c --steps 3 --warmup 1 --no-cpu" >> gpurun_out/r04_sq_counters_$c.txt 2>&1
  done
fi
# keep the merge small: traces and databases stay on the box
find gpurun_out/prof_r04* gpurun_out/pmc_r04* -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_r04* gpurun_out/pmc_r04* -name "*trace.csv" -delete 2>/dev/null
find gpurun_out/pmc_r04* -name "*counter_collection.csv" -delete 2>/dev/null
tail -25 gpurun_out/r04_traffic.txt 2>/dev/null
