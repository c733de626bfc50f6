#!/usr/bin/env python3
"""tests/fuzz_soak.py [first_seed] [n_seeds] [max_cubes]: the fuzz of tests/test_gpu_parity.py::test_random_scenes
over many seeds (GPU through the C-ABI against the oracle: hit planes and ray-tree hashes bit-exact, colour
within 1 LSB on >= 99.9 % of pixels, ray counters equal).  Prints one line per failing seed and a summary.
RTX950_SOAK_HEAVY=k (read here, by the tool): also renders every scene three more times with
rt_set_tuning(RT_TUNE_HEAVY_CLASS, k), so that the side kernel (three-wave workgroups, general cull, bundles)
traces the tiles of cost class >= k; those frames must equal the debug variant's too.
RTX950_SOAK_WAVEFRONT=1: the plain frames are rendered by the wavefront launches (RT_TUNE_WAVEFRONT), twice each,
every third scene with the item capacity cut to 128 (RT_TUNE_WAVEFRONT_CAP: overflowing tiles take the redo list).
RTX950_SOAK_REFINE=1: also renders every scene with the side kernel off and each plain short-list kernel forced in turn
(RT_TUNE_CULL_REFINE 0, 1, 2: plain, cull refinement, refinement and the walk by bundles)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import raytracing_amd as rt  # noqa: E402
import oracle as orc  # noqa: E402
from parity import assert_parity, random_scene  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
max_cubes = int(sys.argv[3]) if len(sys.argv) > 3 else None
base = rt.default_scene()
ctx = rt.Context(0)
soak_heavy = os.environ.get("RTX950_SOAK_HEAVY")
soak_wf = os.environ.get("RTX950_SOAK_WAVEFRONT") == "1"
soak_refine = os.environ.get("RTX950_SOAK_REFINE") == "1"
bad = 0
worst = 1.0
for seed in range(first, first + count):
    sc, label = random_scene(rt, base, seed, max_cubes)
    ctx.upload(sc)
    f = sc.frame(96, 64)
    gfb, gpl = ctx.render(f, debug=True)
    if soak_wf:
        ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, 1)
        ctx.set_tuning(rt.RT_TUNE_WAVEFRONT_CAP, 128 if seed % 3 == 0 else 0)
        first_plain = ctx.render(f)
    plain = ctx.render(f)  # the kernel variant that ships
    if soak_wf:
        ctx.set_tuning(rt.RT_TUNE_WAVEFRONT, -1)
        if not (first_plain == plain).all():
            plain = first_plain  # report below
    ofb, opl, ocnt, _ = orc.OracleScene(sc).render(f, debug=True, reuse_secondary=True)
    try:
        rep = assert_parity(gfb, gpl, ofb, opl, label)
        assert (plain == gfb).all(), f"{label}: plain and debug kernels differ"
        if soak_heavy is not None:
            ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, int(soak_heavy))
            for k in range(3):
                assert (ctx.render(f) == gfb).all(), f"{label}: side kernel (class >= {soak_heavy}), launch {k}: frame differs"
            ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, -1)
            assert ctx.debug_status() == 0, f"{label}: a wait between the streams gave up"
        if soak_refine:
            ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, 8)
            for r in (0, 1, 2):
                ctx.set_tuning(rt.RT_TUNE_CULL_REFINE, r)
                assert (ctx.render(f) == gfb).all(), f"{label}: plain kernel with RT_TUNE_CULL_REFINE {r}: frame differs"
            ctx.set_tuning(rt.RT_TUNE_CULL_REFINE, -1)
            ctx.set_tuning(rt.RT_TUNE_HEAVY_CLASS, -1)
        worst = min(worst, rep["exact_fraction"])
        assert ctx.count_rays(f).uniq_dict() == ocnt.uniq_dict(), f"{label}: ray counters differ"
    except AssertionError as e:
        bad += 1
        print("FAIL", e, flush=True)
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} seeds, {bad} failing", flush=True)
print(f"fuzz_soak: seeds {first}..{first + count - 1}: {bad} failing; lowest exact-colour fraction {worst:.5f}")
sys.exit(1 if bad else 0)
