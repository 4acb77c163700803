"""Turn rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into a traffic table:
HBM bytes per launch for every kernel instantiation, and (optionally) per step of a whole configuration.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o p --output-format csv -- python3 bench.py ... (same command)
    python profiles/collect_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r03_traffic.json [--total c4_sketch 11 [--only 'sg_']]

`--total NAME STEPS`: also the sum over the dispatches of the run (with `--only REGEX`: of the kernels whose name matches,
i.e. without the one-time set-up of the run: sorts, stream building, table sampling) divided by STEPS sketches (warm-up
included in STEPS), stored under NAME -- the counter traffic of one sketch of a configuration that is many kernels.
Entries are merged into an existing output file; a kernel that is already there keeps its entry (the headline run is
collected first: its batched launches are the ones the bench reports).

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB-like units of 1024 bytes as rocprofv3
reports them; on gfx950 FETCH_SIZE counts half of the bytes of wide (16 B/lane) streaming reads, so it is doubled;
WRITE_SIZE is exact for 16-byte streaming stores.  The 8-byte-per-lane accesses of the non-vector paths are
uncalibrated (guide); the table is therefore an estimate for kernels that use them.
"""
import json
import os
import sys

import pandas as pd

NAMES = (r"((?:gemm_f64_kernel|skinny_s_kernel|skinny_r_kernel|chain_step_kernel|chain_wide_kernel|stream_small_kernel|"
         r"sample_rows_kernel|sample_kernel|sparse_psi_mfma_kernel|dense_pass_kernel|chain_sum_kernel|dense_left_pass_kernel)<[^>]*>|rows_longk_kernel|dense_pass_reduce|dense_pass_zsum|splitk_reduce_kernel|skinny_r_reduce2?|small_gemm_kernel|"
         r"sg_pass_kernel|sg_psi_reduce_kernel|sg_om_reduce_kernel|expand_rows_kernel|chol_inv_kernel|hh_sign_scale_kernel)")


def load(path, counter):
    c = pd.read_csv(f"{path}/p_counter_collection.csv")
    c = c[c.Counter_Name == counter].copy()
    c["kern"] = c.Kernel_Name.str.extract(NAMES)
    return c


def per_kernel(c):
    per_dispatch = c.dropna(subset=["kern"]).groupby(["kern", "Dispatch_Id"]).Counter_Value.sum()
    return per_dispatch.groupby("kern").agg(["median", "max", "count"])


args = sys.argv[1:]
total, only = None, None
if "--only" in args:
    i = args.index("--only")
    only = args[i + 1]
    args = args[:i] + args[i + 2:]
if "--total" in args:
    i = args.index("--total")
    total = (args[i + 1], float(args[i + 2]))
    args = args[:i] + args[i + 3:]
fetch_c, write_c = load(args[0], "FETCH_SIZE"), load(args[1], "WRITE_SIZE")
fetch, write = per_kernel(fetch_c), per_kernel(write_c)
dst = args[2] if len(args) > 2 else "profiles/r04_traffic.json"
out = json.load(open(dst)) if os.path.exists(dst) else {}
for k in fetch.index:
    if k in out:
        continue
    # the biggest dispatches of an instantiation are the batched launches (the class the bench reports)
    f = float(fetch.loc[k, "max"]) * 1024 * 2
    w = float(write.loc[k, "max"]) * 1024 if k in write.index else 0.0
    out[k] = dict(fetch_bytes=f, write_bytes=w, bytes_per_launch=f + w, launches_sampled=int(fetch.loc[k, "count"]))
if total:
    fc = fetch_c[fetch_c.Kernel_Name.str.contains(only)] if only else fetch_c
    wc = write_c[write_c.Kernel_Name.str.contains(only)] if only else write_c
    f = float(fc.Counter_Value.sum()) * 1024 * 2 / total[1]
    w = float(wc.Counter_Value.sum()) * 1024 / total[1]
    out[total[0]] = dict(fetch_bytes=f, write_bytes=w, bytes_per_launch=f + w, sketches=total[1],
                         note=("sum over the dispatches matching %r" % only if only else "sum over every dispatch of the run")
                              + " / sketches")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: round(v["bytes_per_launch"] / 1e6, 1) for k, v in out.items()}, indent=1))
