"""Shared comparison helpers for the GPU-vs-oracle parity tests."""
import numpy as np

# north_star: "ray-hit object indices match bit-exactly, and final 8-bit colour is within 1 LSB per
# channel on >= 99.9% of pixels"
COLOUR_LSB_TOL = 1
COLOUR_MIN_FRACTION = 0.999


def channels(fb):
    fb = np.asarray(fb, dtype=np.uint32)
    return np.stack([(fb >> 16) & 255, (fb >> 8) & 255, fb & 255], axis=-1).astype(np.int32)


def colour_report(gpu, ref):
    d = np.abs(channels(gpu) - channels(ref)).max(axis=-1)
    return {
        "exact_fraction": float((d == 0).mean()),
        "within_1lsb_fraction": float((d <= COLOUR_LSB_TOL).mean()),
        "max_abs_diff": int(d.max()),
        "n_pixels": int(d.size),
    }


def assert_parity(gpu_fb, gpu_planes, ref_fb, ref_planes, label=""):
    for k in ("hit_cube", "hit_sphere", "tree_hash"):
        bad = int((gpu_planes[k] != ref_planes[k]).sum())
        assert bad == 0, f"{label}: {k} differs from the oracle on {bad} pixels (must be bit-exact)"
    rep = colour_report(gpu_fb, ref_fb)
    assert rep["within_1lsb_fraction"] >= COLOUR_MIN_FRACTION, f"{label}: colour parity {rep}"
    return rep
