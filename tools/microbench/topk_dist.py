"""Top-k kernel time against the key distribution (dev tool; run under rocprofv3 --kernel-trace --stats):
uniform keys spread over the radix bins, sigmoid / softmax-like keys pile into a few (LDS atomic contention)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
rng = np.random.default_rng(0)
cases = {
    "uniform_201600_k1000": (rng.uniform(0, 1, (2, 201600)), 1000),
    "sigmoid_201600_k1000": (1 / (1 + np.exp(-rng.normal(-2, 2, (2, 201600)))), 1000),
    "uniform_19248_k200x640": (rng.uniform(0, 1, (640, 19248)), 200),
    "softmaxlike_19248_k200x640": (np.exp(rng.normal(-6, 1.5, (640, 19248))), 200),
}
which = sys.argv[1]
keys, k = cases[which]
for _ in range(5): _ffi.topk(keys.astype(np.float32), k)
print(which, "done")
