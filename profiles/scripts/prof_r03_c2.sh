# refresh of the C2 records only: kernel stats and counter traffic per sketch, merged into profiles/r03_traffic.json
# (run from the repo root on the GPU box: bash profiles/scripts/prof_r03_c2.sh; then copy gpurun_out/r03_traffic.json and
# gpurun_out/r03_c2_kernel_stats.csv into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
run() { name=$1; shift; timeout -k 10 400 rocprofv3 "$@" > $R/$name.log 2>&1 || { echo "$name failed"; tail -5 $R/$name.log; exit 1; }; echo "$name ok"; }
run prof_r03_c2 --kernel-trace --stats -d $R/prof_r03_c2 -o bench --output-format csv -- python3 $B --config c2 --steps 10 --warmup 2 --no-cpu
run pmc_r03f_c2 --kernel-trace --pmc FETCH_SIZE -d $R/pmc_r03f_c2 -o p --output-format csv -- python3 $B --config c2 --steps 5 --warmup 2 --no-cpu
run pmc_r03w_c2 --kernel-trace --pmc WRITE_SIZE -d $R/pmc_r03w_c2 -o p --output-format csv -- python3 $B --config c2 --steps 5 --warmup 2 --no-cpu
cd $GRAFT_REPO_ROOT
cp profiles/r03_traffic.json gpurun_out/r03_traffic.json
python3 - <<PY
import json
j = json.load(open("gpurun_out/r03_traffic.json")); j.pop("c2_sketch", None); json.dump(j, open("gpurun_out/r03_traffic.json", "w"), indent=1)
PY
python3 profiles/collect_traffic.py gpurun_out/pmc_r03f_c2 gpurun_out/pmc_r03w_c2 gpurun_out/r03_traffic.json --total c2_sketch 7 > gpurun_out/r03_traffic_c2.txt 2>&1
cp gpurun_out/prof_r03_c2/bench_kernel_stats.csv gpurun_out/r03_c2_kernel_stats.csv
find gpurun_out/prof_r03* gpurun_out/pmc_r03* -name "*.db" -delete 2>/dev/null
find gpurun_out/prof_r03* gpurun_out/pmc_r03* -name "*trace.csv" -delete 2>/dev/null
find gpurun_out/pmc_r03* -name "*counter_collection.csv" -delete 2>/dev/null
grep -n "c2_sketch\|dense_pass" gpurun_out/r03_traffic.json
