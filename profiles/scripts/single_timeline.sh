#!/bin/bash
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/prof_single
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/prof_single -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/profiles/scripts/single_timeline.py > $R/prof_single.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'EOP'
import csv, glob
f = glob.glob("gpurun_out/prof_single/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last sketch: kernels after the last big gap
# find fill_normal (DRM sampling) end; take last N kernels
K = [r for r in rows if 'fill_normal' not in r['Kernel_Name']]
per = len(K) // 20
last = K[-per:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    st, en = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{st/1e3:8.1f} {en/1e3:8.1f} {(en-st)/1e3:7.1f} us  q{r.get('Queue_Id','?'):>3s} {r['Kernel_Name'][:70]}")
EOP
rm -f gpurun_out/prof_single/*.db
