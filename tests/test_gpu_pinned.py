"""Page-locked host arrays of the context (stardis_amd._lib.PinnedPool) and the DMA copies of the C ABI that move them."""
import gc

import numpy as np
import pytest

from stardis_amd import _lib

pytestmark = pytest.mark.gpu


def test_pinned_arrays_round_trip_and_return_to_the_pool(ctx):
    pool = _lib.PinnedPool(ctx, keep=64 << 10, limit=1 << 20)
    a = pool.empty((3, 1000))
    assert a.shape == (3, 1000) and a.dtype == np.float64 and a.flags.writeable and a.flags.c_contiguous
    a[:] = np.arange(3000.0).reshape(3, 1000)
    dev = ctx.empty((3, 1000))
    ctx.call("sdx_memcpy_h2d_pinned", dev.ptr, a.ctypes.data, a.nbytes)
    b = pool.empty((3, 1000))
    ctx.call("sdx_memcpy_d2h_pinned", b.ctypes.data, dev.ptr, b.nbytes)
    assert np.array_equal(a, b)
    assert np.array_equal(dev.numpy(), a)  # pageable destination, bounce buffer
    # a view keeps the block out of the pool; dropping the last one hands it back, and the next array of the class reuses it
    addr = b.ctypes.data
    view = b[1]
    del b
    gc.collect()
    assert pool._out == 2 * pool._capacity(a.nbytes)
    del view
    gc.collect()
    assert pool._out == pool._capacity(a.nbytes)
    c = pool.empty((3, 1000))
    assert c.ctypes.data == addr
    # beyond the limit the pool declines (the caller falls back to pageable memory)
    assert pool.empty(1 << 20, np.uint8) is None
    assert _lib.PinnedPool._capacity(5000) == 8192 and _lib.PinnedPool._capacity((3 << 20) + 1) == 4 << 20
    assert _lib.PinnedPool(ctx).empty(_lib.PinnedPool.BIGGEST + 1, np.uint8) is None  # too large to page-lock per call
    del a, c
    gc.collect()
    assert pool._out == 0


def test_planes_read_back_through_the_pool_equal_the_bounce_copy(ctx):
    rng = np.random.default_rng(3)
    host = rng.random((56, 4000))
    dev = ctx.upload(host)
    assert dev.nbytes >= 64 << 10
    got = dev.numpy()  # pinned path
    assert np.array_equal(got, host)
    plain = np.empty_like(host)
    ctx.call("sdx_memcpy_d2h", plain.ctypes.data, dev.ptr, dev.nbytes)
    assert np.array_equal(plain, host)
    assert ctx.lib.sdx_host_free(None) == 0


def test_closing_a_context_frees_its_idle_blocks():
    c = _lib.Context(0)
    a = c.pinned.empty((10, 1000))
    b = c.pinned.empty((10, 1000))
    del a
    gc.collect()
    assert c.pinned._idle > 0
    pool = c.pinned
    c.close()
    assert pool._idle == 0 and not pool._free
    b[:] = 1.0  # an array handed out earlier stays valid; its block is freed, not pooled, when it goes
    del b
    gc.collect()
    assert pool._idle == 0 and pool._out == 0
    assert pool.empty(16) is None  # no context left to allocate from
