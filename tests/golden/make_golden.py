"""Generates the committed fixtures under tests/golden/ (run once in the build container).

  dc2_b4.npz   inputs: the first 4 stamps of the reference's sample data
               (src/debvader/data/dc2_imgs/imgs_dc2.npy, cast to float32 as deblender.py:18 does), seeded eps;
               expected: outputs of the fp64 oracle (oracle/vae_oracle.py) for weights init_params(seed=11, perturb=0.03).
               The reference itself cannot produce network outputs here (TensorFlow absent, weight shard missing),
               so these vectors pin the ORACLE and are what the GPU box checks the HIP path against.
  helpers.npz  inputs/outputs of the reference's own importable helpers (normalize.py, metrics.mse),
               obtained by importing those two files by path.
  scene.npz    cutouts produced by the reference's own extract_cutouts (extract/extraction.py imported by path) for
               a seeded field, including windows that leave the field; plus residual / predicted fields of a small
               scene computed by oracle/scene_oracle.py (scipy.ndimage.shift with the reference's arguments:
               field_deblender.py itself needs `sep` and cannot be imported).

Usage:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/debvader"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def summarise(a):
    a = np.asarray(a, np.float64).ravel()
    step = max(1, a.size // 64)
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum()), np.abs(a).max()], a[::step][:64]])


def main():
    from oracle import vae_oracle as vo

    imgs = np.load(os.path.join(REF, "data/dc2_imgs/imgs_dc2.npy"))
    x = imgs[:4].astype(np.float32)
    y = imgs[4:8].astype(np.float32)          # any other real stamps as "isolated galaxy" labels
    arch = vo.Arch()
    p = vo.init_params(arch, seed=11, perturb=0.03)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    eps = np.random.default_rng(5).normal(size=(4, 32)).astype(np.float32)
    c = vo.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=True)
    L = vo.losses(arch, c, y.astype(np.float64))
    g = vo.backward(arch, p, c, y.astype(np.float64))
    ci = vo.forward(arch, p, x.astype(np.float64), eps.astype(np.float64), training=False)
    out = dict(x=x, y=y, eps=eps, param_seed=np.array(11), param_perturb=np.array(0.03),
               t=c["t"], z=c["z"], kl=c["kl"], loc_sum=summarise(c["loc"]), scale_sum=summarise(c["scale"]),
               loss=np.array([L["loss"], L["nll_mean"], L["kl_reg"], L["mse"]]),
               infer_t=ci["t"], infer_loc_sum=summarise(ci["loc"]), infer_scale_sum=summarise(ci["scale"]))
    for k, v in g.items():
        out["g/" + k] = summarise(v)
    np.savez_compressed(os.path.join(HERE, "dc2_b4.npz"), **out)

    nz = _load(os.path.join(REF, "normalize/normalize.py"), "ref_normalize")
    mt = _load(os.path.join(REF, "training/metrics.py"), "ref_metrics")
    rng = np.random.default_rng(0)
    a = rng.normal(0, 2, size=(3, 5, 5, 2))
    b = rng.normal(0, 2, size=(3, 5, 5, 2))
    n = nz.normalize_non_linear(a)
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), a=a, b=b, normalized=n,
                        denormalized=nz.denormalize_non_linear(n), mse=np.array(mt.mse(a, b)))
    ex = _load(os.path.join(REF, "extract/extraction.py"), "ref_extraction")
    from oracle import scene_oracle as so
    rng = np.random.default_rng(3)
    F, cs, nb = 41, 11, 2
    field = rng.normal(size=(1, F, F, nb))
    dists = [[-4, -3], [15, 15], [-15, -15], [16, 2], [0, -16], [3.7, -2.2], [-15.9, 14.99], [0, 0], [-40, 3], [-36, -41]]
    cut, idx = ex.extract_cutouts(field.copy(), F, dists, cs, nb)
    stamps = rng.random((6, cs, cs, nb))
    pos = np.array([[0, 0], [5, -7], [-14, 13], [2.5, 3.25], [-17.3, 16.8], [19, -19]], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "scene.npz"), field=field, dists=np.array(dists), cutouts=cut,
                        list_idx=np.array(idx), stamps=stamps, pos=pos,
                        residual=so.residual_field(field[0], stamps, pos, cs),
                        predicted=so.predicted_field(F, nb, stamps, pos, cs))
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
