"""Randomised sweeps of the kernels new in round 4 against numpy -- a tool, not part of the collected suite:
`python tests/fuzz_round4.py SEED SECONDS` on a GPU box.
  * ttsk_chain_step_sum (chain_sum.h): random term counts, slice counts, ranks inside and around its cover, both stride patterns,
    all three T layouts; shapes it declines are skipped;
  * ttsk_dense_left_pass (dense_left_pass.hip) and the long-K rows product behind ttsk_gemm (dense_right_pass.hip);
  * orthogonal_sketch_batch against the single calls (to rounding) on random signatures."""
import os
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tt_sketch_amd as tsa
from tt_sketch_amd import _native as nat
from oracle import ttsk_oracle as orc
from tests import test_gpu_parity as tp, test_gpu_dense_pass as td

nat.call("ttsk_init", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = time.time()
count = dict(chain_sum=0, chain_sum_declined=0, left_pass=0, rows=0, orth_batch=0)
while time.time() - t0 < budget:
    pick = rng.integers(0, 4)
    try:
        if pick == 0:
            case = (int(rng.integers(1, 41)), int(rng.integers(1, 60)), int(rng.integers(1, 21)), int(rng.integers(1, 129)),
                    2 * int(rng.integers(1, 65)), int(rng.integers(1, 21)), bool(rng.integers(0, 2)), int(rng.integers(0, 3)))
            try:
                tp._chain_sum_case(case, int(rng.integers(0, 1000)))
                count["chain_sum"] += 1
            except nat.TtskUnsupported:
                count["chain_sum_declined"] += 1
        elif pick == 1:
            n4 = int(rng.choice([64, 128, 256, 512]))
            n3 = 512 // n4 * int(rng.integers(1, 3))
            n0, n1, n2, l = 4 * int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(1, 9)), int(rng.integers(1, 21))
            td.test_left_pass_against_einsum(tsa, n0, n1, n2, n3, n4, l)
            count["left_pass"] += 1
        elif pick == 2:
            rows, N, K = int(rng.integers(1, 300)), int(rng.integers(1, 49)), 64 * int(rng.integers(64, 400))
            td.test_rows_against_a_matrix_long_k(tsa, rows, N, K, 2 * int(rng.integers(0, 40)))
            count["rows"] += 1
        else:
            d = int(rng.integers(3, 6))
            shape = tuple(int(x) for x in rng.integers(6, 40, d))
            s_in, l = int(rng.integers(4, 20)), int(rng.integers(2, min(10, min(shape))))
            r = l + int(rng.integers(1, 8))
            ld, rd = orc.random_tt_drm(shape, l, False, rng), orc.random_tt_drm(shape, r, True, rng)
            left = tsa.TensorTrainDRM(l, shape, False, seed=1, cores=[np.array(c) for c in ld.cores])
            right = tsa.TensorTrainDRM(r, shape, True, seed=2, cores=[np.array(c) for c in rd.cores])
            tts = [tsa.TensorTrain(orc.random_tt(shape, s_in, rng)) for _ in range(int(rng.integers(2, 11)))]
            got = tsa.orthogonal_sketch_batch(tts, (l,) * (d - 1), (r,) * (d - 1), left_drm=left, right_drm=right)
            for t, g in zip(tts, got):
                one = tsa.orthogonal_sketch(t, (l,) * (d - 1), (r,) * (d - 1), left_drm=left, right_drm=right)
                assert all(np.abs(np.asarray(a) - np.asarray(b)).max() <= 1e-8 * np.abs(np.asarray(a)).max() for a, b in zip(one.cores, g.cores))
                assert g.error(t, relative=True) < 1e-6 or s_in > l      # exact recovery where the sketch rank covers the TT rank
            count["orth_batch"] += 1
    except ValueError:
        pass                     # (a sketch rank beyond an unfolding's rows: the API's own refusal)
    except AssertionError as e:
        print("FAIL", pick, locals().get("case"), repr(e)[:300], flush=True)
print("fuzz_round4:", count)
