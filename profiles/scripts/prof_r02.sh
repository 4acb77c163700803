# round-2 profile of the bench command (run on the GPU box from the repo root: bash profiles/scripts/prof_r02.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $R/prof_r02d $R/prof_r02s $R/pmc_r02f $R/pmc_r02w $R/pmc_r02sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_r02d -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $R/prof_r02d.log 2>&1 || { echo stats failed; exit 1; }
echo stats ok
export TTSK_SINGLE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/prof_r02s -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --inflight 1 > $R/prof_r02s.log 2>&1 || { echo single failed; exit 1; }
echo single ok
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/pmc_r02f -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --inflight 1 > $R/pmc_r02f.log 2>&1 || { echo fetch failed; exit 1; }
echo fetch ok
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/pmc_r02w -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --inflight 1 > $R/pmc_r02w.log 2>&1 || { echo write failed; exit 1; }
echo write ok
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/pmc_r02sq -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --inflight 1 > $R/pmc_r02sq.log 2>&1 || { echo sq failed; exit 1; }
echo sq ok
unset TTSK_SINGLE_STREAM
cd $GRAFT_REPO_ROOT
python3 profiles/by_grid.py gpurun_out/prof_r02d/bench_kernel_trace.csv gpurun_out/r02_kernel_by_grid.csv > gpurun_out/r02_by_grid.txt 2>&1
python3 profiles/by_grid.py gpurun_out/prof_r02s/bench_kernel_trace.csv gpurun_out/r02_kernel_by_grid_single_stream.csv > gpurun_out/r02_by_grid_single.txt 2>&1
python3 profiles/collect_traffic.py gpurun_out/pmc_r02f gpurun_out/pmc_r02w gpurun_out/r02_traffic.json > gpurun_out/r02_traffic.txt 2>&1
cp gpurun_out/prof_r02d/bench_kernel_stats.csv gpurun_out/r02_bench_kernel_stats.csv
cat gpurun_out/r02_by_grid_single.txt | head -14
