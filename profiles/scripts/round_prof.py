import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import tt_sketch_amd as tsa
from tt_sketch_amd import _native as nat
from oracle import ttsk_oracle as orc
rng = np.random.default_rng(0)
tt = tsa.TensorTrain(orc.random_tt((200,)*6, (100,)*5, rng)).to_device()
for _ in range(3):
    t = time.perf_counter(); o = tt.round_dev(max_rank=50); nat.call("ttsk_sync", -1); print("round_dev ms", (time.perf_counter()-t)*1e3)
for _ in range(3):
    t = time.perf_counter(); o = tt.orthogonalize_dev(); nat.call("ttsk_sync", -1); print("orth_dev ms", (time.perf_counter()-t)*1e3)
