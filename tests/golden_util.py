"""Helpers for comparing tensors with the committed golden vectors (tests/golden/*.npz, produced by
oracle/gen_golden.py from the real HF SiglipVisionModel in the build container)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_32", "tiny_48_interp", "hostile_42", "hostile_98_interp", "so400m1_384", "so400m1_224_interp",
         "base1_224"]
SMALL_CASES = CASES[:4]


def load(case):
    return dict(np.load(os.path.join(GOLDEN_DIR, case + ".npz")))


def meta(rec):
    return dict(config=str(rec["meta.config"]), seed=int(rec["meta.seed"]), batch=int(rec["meta.batch"]),
                res=int(rec["meta.res"]), interp=bool(int(rec["meta.interp"])),
                taps=tuple(int(t) for t in rec["meta.taps"]))


def compare(rec, prefix, tensor, atol, rtol):
    """Return (max_abs_err, reference_scale); asserts shape and tolerance."""
    a = np.asarray(tensor, dtype=np.float32).reshape(-1)
    shape = tuple(int(s) for s in rec[prefix + ".shape"])
    assert int(np.prod(shape)) == a.size, f"{prefix}: size {a.size} vs golden shape {shape}"
    if prefix + ".full" in rec:
        ref = rec[prefix + ".full"]
        got = a
    else:
        ref = rec[prefix + ".samples"]
        got = a[rec[prefix + ".idx"]]
        s = float(a.astype(np.float64).sum())
        scale = float(rec[prefix + ".abssum"])
        assert abs(s - float(rec[prefix + ".sum"])) <= (rtol * scale + atol * a.size), \
            f"{prefix}: checksum {s} vs {float(rec[prefix + '.sum'])} (abssum {scale})"
    err = float(np.abs(got - ref).max())
    ref_scale = float(np.abs(ref).max())
    if rtol >= 1e-3:      # bf16-style call: atol is the whole element-wise bound, rtol only scales the checksum check
        assert err <= atol, f"{prefix}: max|err| {err:.3e} > {atol:.3e}"
        return err, ref_scale
    assert err <= atol + rtol * ref_scale, f"{prefix}: max|err| {err:.3e} > {atol} + {rtol}*{ref_scale:.3e}"
    return err, ref_scale


def err_stats(rec, prefix, tensor):
    """(max|err|, relative L2 error) against the fp32 golden values, on exactly the elements the fixture keeps — the same
    statistic oracle/gen_golden.py stores under ``bf16ac.<prefix>.*`` for the real HF model under CPU bf16 autocast."""
    a = np.asarray(tensor, dtype=np.float32).reshape(-1)
    if prefix + ".full" in rec:
        ref, got = rec[prefix + ".full"], a
    else:
        ref, got = rec[prefix + ".samples"], a[rec[prefix + ".idx"]]
    e = got.astype(np.float64) - ref.astype(np.float64)
    return float(np.abs(e).max()), float(np.sqrt((e * e).sum()) / (np.sqrt((ref.astype(np.float64) ** 2).sum()) + 1e-30))
