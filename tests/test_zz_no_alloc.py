"""The entry point that allocates nothing (rt_gather_radiance_static) against rt_gather_radiance, through the C ABI:
same kernels, scratch in the library's static device array, so the results must be the same bits -- on two calls in a
row, the second with a smaller input (the array keeps the first call's contents).  (Radius gathers: the same bits as the
allocating entry point's walk lane by lane; its queued route for large inputs, which needs buffers that grow with the
input, sums the same photons in another order and is held to fp32 rounding.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_gather_equals_allocating_gather():
    import torch
    from raytracer_amd import capi, synth
    ctx = capi.Context(0)
    try:
        rec = synth.box_photons(200_000, seed=11)
        m = ctx.photon_map(rec)
        hits_all = synth.box_queries(640, 400)[:capi.GATHER_STATIC_MAX_QUERIES]
        # queries far apart in one group force the second and third passes (split groups, per-lane heaps): shuffle a block
        rng = np.random.default_rng(0)
        hits_all[64 * 10:64 * 40] = hits_all[rng.permutation(hits_all.shape[0])[:64 * 30]]
        for n in (hits_all.shape[0], 64 * 37 + 5):                  # the second call is smaller and ragged
            hits = np.ascontiguousarray(hits_all[:n])
            for kw in (dict(k=64), dict(k=1), dict(r=0.3)):
                with ctx.option("defer_gathers", 0, -1):
                    want = m.gather(hits, **kw)
                if "r" in kw:
                    queued = m.gather(hits, **kw)
                    assert np.abs(queued - want).max() <= 3e-5 * max(1.0, float(np.abs(want).max()))
                d_hits = torch.from_numpy(hits).cuda()
                d_out = torch.full((n, 3), -7.0, dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                free_before = torch.cuda.mem_get_info()[0]          # device memory free on the GPU: any hipMalloc would lower it
                m.gather_static(d_hits, d_out, **kw)
                assert torch.cuda.mem_get_info()[0] == free_before
                got = d_out.cpu().numpy()
                assert np.array_equal(got, want), (n, kw, float(np.abs(got - want).max()))
        # limits and misuse are errors, not faults
        too_many = torch.zeros((capi.GATHER_STATIC_MAX_QUERIES + 1, 9), dtype=torch.float32, device="cuda")
        out = torch.zeros((capi.GATHER_STATIC_MAX_QUERIES + 1, 3), dtype=torch.float32, device="cuda")
        with pytest.raises(capi.RtError) as e:
            m.gather_static(too_many, out, k=8)
        assert e.value.code == -8 and "limit" in str(e.value)
        with pytest.raises(capi.RtError) as e:                      # a host array would need a staging allocation
            m.gather_static(np.zeros((64, 9), np.float32), out[:64], k=8)
        assert e.value.code == -1
    finally:
        ctx.close()
