"""Split a rocprofv3 kernel trace by kernel instantiation and launch geometry.

    python profiles/by_grid.py gpurun_out/prof_r01d/bench_kernel_trace.csv profiles/r01_kernel_by_grid.csv

The batched chain steps of the bench are the launches with the largest grid of their instantiation;
durations in microseconds.
"""
import re
import sys

import pandas as pd

t = pd.read_csv(sys.argv[1])
t["us"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t["kernel"] = t.Kernel_Name.str.extract(
    r"((?:gemm_f64_kernel|skinny_s_kernel|skinny_r_kernel|chain_step_kernel|stream_small_kernel)<[^>]*>|[A-Za-z_0-9]*(?:reduce|copyBuffer|fill|probe|sum_slices|small_gemm|axpby|jacobi|chol|hh_sign)[A-Za-z_0-9]*)")[0]
t["kernel"] = t.kernel.fillna(t.Kernel_Name.str.slice(0, 40))
g = t.groupby(["kernel", "Grid_Size_X", "Grid_Size_Y"]).us.agg(["count", "mean", "median", "min", "max"]).round(2)
g.reset_index().to_csv(sys.argv[2], index=False)
print(g.sort_values("mean", ascending=False).head(16).to_string())
