"""SQ counter table of the batched launches from a `rocprofv3 --pmc SQ_...` pass (profiles/scripts/prof_r02.sh):
    python profiles/sq_counters.py gpurun_out/pmc_r02sq/p_counter_collection.csv > profiles/r02_sq_counters.txt
Median per (kernel, counter) over the launches with the largest grid of that kernel (the batched ones)."""
import re
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1])
df["kern"] = df["Kernel_Name"].str.replace(r"^void ", "", regex=True).str.replace(r"^ttsk::", "", regex=True).str.replace(r"\(.*$", "", regex=True)
keep = df.groupby("kern")["Grid_Size"].transform("max") == df["Grid_Size"]
t = df[keep].pivot_table(index="kern", columns="Counter_Name", values="Counter_Value", aggfunc="median")
t = t[t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0] if "SQ_VALU_MFMA_BUSY_CYCLES" in t else t
regs = df[keep].groupby("kern")[["VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"]].max()
t = t.join(regs, how="left")
if {"SQ_WAIT_ANY", "SQ_WAVE_CYCLES"} <= set(t.columns):
    t["wait_any%"] = (100 * t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"]).round(1)
    t["wait_inst%"] = (100 * t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"]).round(1)
    t["active%"] = (100 * t["SQ_ACTIVE_INST_ANY"] / t["SQ_WAVE_CYCLES"]).round(1)
if {"SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"} <= set(t.columns):
    t["lds_conflict%"] = (100 * t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"]).round(1)
print("# rocprofv3 --pmc (SQ counters) of `%s`; median over the largest launches of each kernel" % (sys.argv[2] if len(sys.argv) > 2 else "bench.py --steps 5 --warmup 2 --no-cpu --no-extra --inflight 1 (batch 32), TTSK_SINGLE_STREAM=1"))
print("# SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs (64 per 16x16x4, 16 per 4x4x4); the other SQ counters are quad-cycles")
with pd.option_context("display.width", 400, "display.max_columns", 30, "display.max_colwidth", 60):
    print(t.to_string())
