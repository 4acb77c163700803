"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only)
into profiles/r02_traffic.json (or the file named as third argument): HBM bytes per launch for every contraction-kernel instantiation.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch2 -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu --inflight 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write2 -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu --inflight 1
    python profiles/collect_traffic.py gpurun_out/pmc_fetch2 gpurun_out/pmc_write2

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB-like units of
1024 bytes as rocprofv3 reports them; on gfx950 FETCH_SIZE counts half of the bytes of wide
(16 B/lane) streaming reads, so it is doubled; WRITE_SIZE is exact for 16-byte streaming stores.
The 8-byte-per-lane accesses of the non-vector paths are uncalibrated (guide); the table is
therefore an estimate for kernels that use them.
"""
import json
import re
import sys

import pandas as pd


def per_kernel(path, counter):
    c = pd.read_csv(f"{path}/p_counter_collection.csv")
    c = c[c.Counter_Name == counter]
    c["kern"] = c.Kernel_Name.str.extract(
        r"((?:gemm_f64_kernel|skinny_s_kernel|skinny_r_kernel|chain_step_kernel|stream_small_kernel)<[^>]*>|splitk_reduce_kernel|skinny_r_reduce|small_gemm_kernel)")
    per_dispatch = c.groupby(["kern", "Dispatch_Id"]).Counter_Value.sum()
    return per_dispatch.groupby("kern").agg(["median", "max", "count"])


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch.index:
    # the biggest dispatches of an instantiation are the batched chain steps (the class the bench reports)
    f = float(fetch.loc[k, "max"]) * 1024 * 2
    w = float(write.loc[k, "max"]) * 1024 if k in write.index else 0.0
    out[k] = dict(fetch_bytes=f, write_bytes=w, bytes_per_launch=f + w, launches_sampled=int(fetch.loc[k, "count"]))
json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/r02_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
