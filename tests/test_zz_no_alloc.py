"""rt_render_device_noalloc (the allocation-free entry point: fixed work buffer per GPU inside the library)
against rt_render_device, through the C-ABI: two calls in a row, the second with a smaller input; then a
different scene and view on a second context (the buffer is shared per GPU and keeps the last call's
contents), a band, a width the 16-byte paint stores do not fit, and the size limit."""
import numpy as np
import pytest

import raytracing_amd as rt

pytestmark = pytest.mark.gpu


def _pair(ctx, f):
    import torch
    rows = f.y1 - f.y0
    a = torch.full((rows, f.width), 0x55AA55, dtype=torch.int32, device="cuda")
    b = torch.full((rows, f.width), 0x33CC33, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # torch's fills are not ordered against the library's stream
    ctx.render_device(f, a.data_ptr())
    ctx.sync()
    ctx.render_device_noalloc(f, b.data_ptr())
    ctx.sync()
    return a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)


def test_noalloc_entry_point_matches_render_device(default_scene):
    with rt.Context(0) as ctx:
        ctx.upload(default_scene)
        # two calls in a row, the second with a smaller input
        for w, h in ((1080, 720), (328, 200)):
            want, got = _pair(ctx, default_scene.frame(w, h))
            assert np.array_equal(got, want), (w, h, int((got != want).sum()))
            assert (want != default_scene.sky_0rgb).any()
        # a band, an odd width (no 16-byte stores), the smallest frame
        for w, h, y0, y1 in ((200, 150, 37, 149), (37, 19, 0, None), (1, 1, 0, None)):
            want, got = _pair(ctx, default_scene.frame(w, h, y0=y0, y1=y1))
            assert np.array_equal(got, want), (w, h, y0, y1)
        # another context on the same GPU shares the buffer: a long-list scene with spheres, then the diorama again
        with rt.Context(0) as other:
            s = rt.synthetic_scene(n_cubes=700, n_spheres=5)
            other.upload(s)
            want, got = _pair(other, s.frame(320, 180))
            assert np.array_equal(got, want)
        cam = rt.rotate_cam(default_scene.camera, 0.9, -0.2)
        want, got = _pair(ctx, default_scene.frame(640, 360, camera=cam))
        assert np.array_equal(got, want)
        # the largest frame it takes (4K = 129 600 tiles) and one beyond the limit
        want, got = _pair(ctx, default_scene.frame(3840, 2160))
        assert np.array_equal(got, want)
        big = default_scene.frame(4096, 2304)
        assert (4096 // 8) * (2304 // 8) > rt.RT_NOALLOC_MAX_TILES
        with pytest.raises(rt.RtError) as e:
            ctx.render_device_noalloc(big, 8)
        assert e.value.code == rt._ffi.RT_ERR_LIMIT and "RT_NOALLOC_MAX_TILES" in str(e.value)
