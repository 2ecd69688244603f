"""Development-build engine (general kernels), a few fp32 train steps, close: the sequence of tests/test_gpu_bf16.py's "f32alt" run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from debvader_amd import engine as E
from debvader_amd._lib import check
from debvader_amd.data import synthetic_stamps
from tests import debug_lib

if os.environ.get("DV_PROBE_DEBUG_LIB"):
    debug_lib.DEBUG_LIB_PATH = os.environ["DV_PROBE_DEBUG_LIB"]

B = 64
x, y = synthetic_stamps(B, seed=0)
with debug_lib.debug_build() as dlib:
    check(dlib.dv_debug_general_kernels(1 if len(sys.argv) < 2 else int(sys.argv[1])))
    eng = E.Engine(E.make_config(max_batch=B, dtype=0))
    eng.optimizer_reset(1e-4)
    eng.upload(0, x, y)
    out = eng.train_steps(0, 0, B, 3, seed=1)
    print("loss", out["loss"], flush=True)
    eng.close()
    print("closed", flush=True)
    check(dlib.dv_debug_general_kernels(0))
print("done")
