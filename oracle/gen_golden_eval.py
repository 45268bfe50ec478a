"""Generates tests/golden/eval.npz by RUNNING the reference's own garg_crop / eigen_crop / compute_errors (depth_evaluation.py:L16-53) on
seeded inputs.  Runs only in the build container (needs /root/reference).  The module itself cannot be imported (cv2, fvcore are absent and
irrelevant to these three pure-numpy functions), so the three function definitions are compiled from the reference file's syntax tree,
unmodified, into a namespace that holds numpy -- the same "load the leaf, skip the package" approach as oracle/ref_harness.py.

    python -m oracle.gen_golden_eval
"""
import ast
import os

import numpy as np

REF = "/root/reference/detectron2/evaluation/depth_evaluation.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "eval.npz")
WANTED = ("garg_crop", "eigen_crop", "compute_errors")


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in keep) == sorted(WANTED)
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns


def synth(seed, gh, gw, ph, pw, density):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:ph, 0:pw].astype(np.float32)
    pred = (3.0 + 60.0 * (1.0 - yy / ph) ** 2 + 2.0 * np.sin(xx / 37.0) + r.random((ph, pw)) * 0.5).astype(np.float32)
    # ground truth: the up-sampled prediction distorted by a smooth factor + noise, LiDAR-sparse, some > 80 m and some tiny values
    up = pred[(np.arange(gh) * ph // gh)[:, None], (np.arange(gw) * pw // gw)[None, :]]
    gt = (up * (1.0 + 0.25 * np.sin(np.arange(gw) / 90.0))[None, :] * np.exp(r.normal(0, 0.08, (gh, gw)))).astype(np.float32) * 1.3
    gt[r.random((gh, gw)) > density] = 0.0
    gt[r.random((gh, gw)) < 0.002] = 85.0
    return pred, gt


def main():
    ref = reference_functions()
    out = {}
    cases = [("k0", 11, 375, 1242, 192, 640, 0.2), ("k1", 12, 370, 1226, 192, 640, 0.05), ("small", 13, 48, 160, 24, 80, 0.5)]
    for tag, seed, gh, gw, ph, pw, dens in cases:
        pred, gt = synth(seed, gh, gw, ph, pw, dens)
        out[f"{tag}.pred"], out[f"{tag}.gt"] = pred, gt
        # the evaluator's own steps around the three functions (process L74-104), with an integer-exact nearest up-sampling so that the
        # golden does not depend on cv2: full[y, x] = pred[y*ph//gh, x*pw//gw]
        ymap, xmap = (np.arange(gh) * ph // gh).astype(np.int32), (np.arange(gw) * pw // gw).astype(np.int32)
        out[f"{tag}.ymap"], out[f"{tag}.xmap"] = ymap, xmap
        full = pred[ymap[:, None], xmap[None, :]]
        for crop in ("garg", "eigen"):
            p, g = ref[f"{crop}_crop"](full, gt)
            out[f"{tag}.{crop}.shape"] = np.array(g.shape)
            for scale in (0, 1):
                pp = p
                valid = np.logical_and(g > 1e-3, g < 80)
                if scale:
                    pp = p * np.median(g[valid]) / np.median(p[valid])
                    out[f"{tag}.{crop}.medians"] = np.array([np.median(g[valid]), np.median(p[valid])], np.float64)
                for lo, hi in ((1e-3, 80), (1e-3, 30), (30, 50), (50, 80)):
                    valid = np.logical_and(g > lo, g < hi)
                    res = ref["compute_errors"](g[valid], pp[valid]) if valid.sum() > 0 else [np.nan] * 9
                    out[f"{tag}.{crop}.s{scale}.{lo:g}_{hi:g}"] = np.array([float(v) for v in res] + [float(valid.sum())], np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
