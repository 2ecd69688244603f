"""Line-level timing of DeblendField.deblend_field (default path) on the GPU box: where the Python side of the reference's
call sequence spends its time beside the engine call.  python tools/probes/df_lines.py [n_per_call]"""
import collections
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def main():
    from debvader_amd import engine as E
    from debvader_amd.deblend.field_deblender import DeblendField
    from debvader_amd.model import model
    from tools.field_cutouts import synthetic_field

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    ctx = E.default_context()
    field = np.asarray(synthetic_field(), np.float64)
    field = field.reshape(field.shape[-3:])
    scene = np.ascontiguousarray(np.tile(field, (8, 8, 1)))
    F, cs = scene.shape[0], 59
    starts = np.random.default_rng(0).integers(0, F - cs + 1, size=(n, 2))
    dist = (starts + cs // 2 - F // 2).astype(np.float64)
    net, _, _, _ = model.create_model_vae((cs, cs, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=8192, ctx=ctx, seed=0)
    db = DeblendField(net, scene[None])
    db.deblend_field(dist[:16384])
    acc = collections.defaultdict(float)
    last = [None, 0.0]
    code = DeblendField.deblend_field.__code__

    def tracer(frame, event, arg):
        if frame.f_code is not code:
            return None

        def local(frame, event, arg):
            now = time.perf_counter()
            if last[0] is not None:
                acc[last[0]] += now - last[1]
            last[0] = frame.f_lineno
            last[1] = time.perf_counter()
            return local
        return local

    for rep in range(2):
        acc.clear()
        last[0] = None
        t0 = time.perf_counter()
        sys.settrace(tracer)
        res = db.deblend_field(dist)
        sys.settrace(None)
        t1 = time.perf_counter()
        del res
        t2 = time.perf_counter()
        print(f"rep {rep}: call {t1 - t0:.3f} s, del res {t2 - t1:.3f} s, {n / (t2 - t0):.0f} stamps/s")
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]:
            print(f"   line {k}: {v:.3f} s")


if __name__ == "__main__":
    main()
