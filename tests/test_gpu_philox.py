"""Counter-based fast sampler (RT_SAMPLER_PHILOX) -- other samples than the reference's MT19937 stream,
reported separately.  What is checked: the generator is Philox4x32-10 (known answers from the Random123
distribution), the GPU offsets equal a numpy restatement, and a frame rendered with them equals the
oracle's frame for the same explicit offsets bit for bit (the traversal path is the parity path)."""
import numpy as np
import pytest

import orc
import raytracer_amd as ra
from raytracer_amd import meshgen

pytestmark = pytest.mark.gpu


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """numpy, vectorised over counters (uint64 arrays holding 32-bit values)."""
    M0, M1, lo32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) for x in (c0, c1, c2, c3))
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0), p1 & lo32, (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1), p0 & lo32
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_offsets(first_sample, n, seed):
    s = np.arange(first_sample, first_sample + n, dtype=np.uint64)
    z = np.zeros(n, np.uint64)
    w = philox4x32_10(s & np.uint64(0xFFFFFFFF), s >> np.uint64(32), z, z, seed, 0)
    w = [x.astype(np.float64) for x in w]
    ox = (w[0] + w[1] * 4294967296.0) / 18446744073709551616.0
    oy = (w[2] + w[3] * 4294967296.0) / 18446744073709551616.0
    ox = np.where(ox >= 1.0, np.nextafter(1.0, 0.0), ox)
    oy = np.where(oy >= 1.0, np.nextafter(1.0, 0.0), oy)
    return np.stack([ox - 0.5, oy - 0.5], -1)


def test_numpy_restatement_has_the_published_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd))]
    for c, k, want in kat:
        got = philox4x32_10(*[np.array([x], np.uint64) for x in c], *k)
        assert tuple(int(g[0]) for g in got) == want


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def test_philox_frame_equals_oracle_frame_with_the_same_offsets(ctx):
    c, k = meshgen.plate(40)
    H, W, spp, pos = 30, 44, 7, 4 * 1234
    cam = meshgen.camera_arrays(W, H)
    host = ra.HostScene([c], [k], 0)
    ctx.upload(host)
    ctx.rng_seed(777)
    ctx.rng_seek(pos)
    ctx.render(*cam, H, W, spp, hit_ids=True, philox=True)
    got = ctx.download()
    assert ctx.rng_tell() == pos + H * W * spp * 4
    jit = philox_offsets(pos // 4, H * W * spp, 777)
    assert jit.min() >= -0.5 and jit.max() < 0.5 and abs(jit.mean()) < 0.01
    want = orc.Scene([c], [k], 0).render(*cam, H, W, spp, jitter=jit)
    mesh, elem = host.map_prims(got["hit_prim"])
    assert np.array_equal(elem, want["hit_elem"])
    assert np.array_equal(got["intensity"].view(np.uint64), want["intensity"].view(np.uint64))
    # bands see the same samples; an MT19937 frame right after it starts at the right stream position
    ctx.rng_seed(777)
    ctx.rng_seek(pos + 10 * W * spp * 4)
    ctx.render(*cam, H, W, spp, rows=(10, 21), philox=True)
    band = ctx.download()
    assert np.array_equal(band["intensity"].view(np.uint64), got["intensity"][10:21].view(np.uint64))
    p1 = ctx.rng_tell()
    ctx.render(*cam, H, W, spp)
    a = ctx.download()["intensity"]
    g = orc.Rng(777)
    g.discard(p1)
    b = orc.Scene([c], [k], 0).render(*cam, H, W, spp, rng=g)["intensity"]
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
