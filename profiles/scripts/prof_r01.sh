cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $R/prof_r01d $R/prof_r01s $R/pmc_fetch2 $R/pmc_write2
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/prof_r01d -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu > $R/prof_r01d.log 2>&1 || { echo stats failed; exit 1; }
echo stats ok
export TTSK_SINGLE_STREAM=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/prof_r01s -o bench --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --inflight 1 > $R/prof_r01s.log 2>&1 || { echo single failed; exit 1; }
unset TTSK_SINGLE_STREAM
echo single ok
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/pmc_fetch2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --inflight 1 > $R/pmc_fetch2.log 2>&1 || { echo fetch failed; exit 1; }
echo fetch ok
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/pmc_write2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --inflight 1 > $R/pmc_write2.log 2>&1 || { echo write failed; exit 1; }
echo write ok
