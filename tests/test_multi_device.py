"""One context for several GPUs (flgpu_config::n_devices, reference analogue: the one Arc<State> behind all workers,
src/main.rs:108-112): every batch is cut into contiguous shards balanced by algorithmic bytes, results
come back in request order and are byte-identical to a single-device run.

CPU part: the shard map (flgpu_plan_shards is a pure function).  GPU part: a context over devices=[0, 0] on the 1-GPU
box -- two shard contexts with their own streams, arenas and scratch on the one card -- runs BASELINE config 4's
1:6:3 size mix and config 3's shape and must reproduce the single-device bytes and the oracle."""
import threading

import numpy as np
import pytest

import oracle_lib
import parity
import synth


# ------------------------------------------------------------------------------------------------ CPU: the shard map --

def test_shards_are_contiguous_cover_everything_and_balance_bytes(fl):
    p = fl.make_params(300, 200)
    shapes = [(1080, 1920, 3)] * 1024
    for k in (1, 2, 3, 4, 8):
        shard_of, shard_bytes = fl.plan_shards(k, shapes, p)
        assert shard_of[0] == 0 and shard_of[-1] == k - 1
        assert np.all(np.diff(shard_of.astype(np.int64)) >= 0)                       # contiguous runs
        counts = np.bincount(shard_of, minlength=k)
        assert counts.max() - counts.min() <= 1 and counts.sum() == 1024
        assert int(shard_bytes.sum()) == 1024 * (1080 * 1920 * 3 + 300 * 200 * 4)   # W*H*C + out_bytes per image


def test_mixed_sizes_are_balanced_by_bytes_not_by_count(fl):
    # BASELINE config 4: 4K / 1080p / thumbnails in ratio 1:6:3
    kinds = [(2160, 3840, 3)] * 100 + [(1080, 1920, 3)] * 600 + [(120, 160, 3)] * 300
    rng = np.random.default_rng(4)
    rng.shuffle(kinds)
    q = fl.Query.parse("w=300&h=200&webp=true&quality=85")
    params, _ = q.to_params(fl.Format.from_accept_header("image/webp"), input_is_jpeg=True)
    shard_of, shard_bytes = fl.plan_shards(8, [tuple(k) for k in kinds], params)
    assert np.all(np.diff(shard_of.astype(np.int64)) >= 0) and set(shard_of.tolist()) == set(range(8))
    mean = shard_bytes.mean()
    assert shard_bytes.max() <= mean + 2160 * 3840 * 3 + 1e6        # no shard is off by more than one largest image
    counts = np.bincount(shard_of, minlength=8)
    assert counts.max() > counts.min()                                # counts differ: the split follows bytes


def test_fewer_images_than_shards_and_per_image_params(fl):
    ps = [fl.make_params(300, 200), fl.make_params(64, 64, crop=True)]
    shard_of, shard_bytes = fl.plan_shards(8, [(100, 100, 3), (50, 50, 4)], ps)
    assert shard_of.tolist() == sorted(shard_of.tolist()) and len(set(shard_of.tolist())) == 2
    assert int((shard_bytes > 0).sum()) == 2
    shard_of, _ = fl.plan_shards(4, [], fl.make_params())
    assert shard_of.size == 0
    with pytest.raises(fl.FanlinError):
        fl.plan_shards(9, [(10, 10, 3)], fl.make_params())           # FLGPU_MAX_DEVICES = 8


# ------------------------------------------------------------------------------------------------------- GPU: [0, 0] --

@pytest.fixture(scope="module")
def two_shards(fl):
    from conftest import require_device
    require_device()
    st = fl.State(devices=[0, 0], profile=True)
    yield st
    st.close()


def _device_batch(fl, st, pool, order, params, cap):
    """Runs images pool[order[i]] (device tensors) through process_batch_device; returns the host copy of every output."""
    import torch
    n = len(order)
    dst = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
    st.process_batch_device([pool[k].data_ptr() for k in order], [tuple(pool[k].shape) for k in order], params,
                            [dst.data_ptr() + i * cap for i in range(n)], [cap] * n, stream=torch.cuda.current_stream().cuda_stream)
    res = st.batch_results()
    torch.cuda.synchronize()
    return dst.cpu().numpy(), res


@pytest.mark.gpu
def test_config4_mix_on_two_shards_equals_one_device_and_the_oracle(fl, gpu_state, two_shards, oracle):
    import torch
    assert two_shards.devices() == [0, 0] and gpu_state.devices() == [0]
    # 1,000 images, 4K : 1080p : thumbnail = 1 : 6 : 3 (a few distinct pictures of each size, referenced many times)
    pool = [torch.from_numpy(synth.uniform(2160, 3840, 3, index=900 + i)).cuda() for i in range(3)] + \
           [torch.from_numpy(synth.photo(1080, 1920, 3, index=910 + i)).cuda() for i in range(6)] + \
           [torch.from_numpy(synth.uniform(120, 160, 3, index=920 + i)).cuda() for i in range(4)]
    kinds = [0] * 100 + [1] * 600 + [2] * 300
    rng = np.random.default_rng(4)
    rng.shuffle(kinds)
    order = [(i % 3) if k == 0 else (3 + i % 6) if k == 1 else (9 + i % 4) for i, k in enumerate(kinds)]
    q = fl.Query.parse("w=300&h=200&webp=true&quality=85")
    params, _ = q.to_params(fl.Format.from_accept_header("image/webp"), input_is_jpeg=True)
    assert params.front_end == fl.FE_WEBP420
    cap = (int(fl.plan_output(params, 1920, 1080, 3).out_bytes) + 255) // 256 * 256
    one, res1 = _device_batch(fl, gpu_state, pool, order, params, cap)
    two, res2 = _device_batch(fl, two_shards, pool, order, params, cap)
    assert np.array_equal(one, two) and res1 == res2
    stats = two_shards.stats()
    assert stats["images"] >= 1000 and stats["batches"] >= 2           # both shard contexts launched
    # request order: output i belongs to input order[i] -- every picture of the pool maps to one distinct output
    firsts = {}
    for i, k in enumerate(order):
        firsts.setdefault(k, i)
        assert np.array_equal(two[i], two[firsts[k]])
    # ... and the planes are what the oracle's chain (resize + letterbox, libwebp's YUV420 front end) makes of them
    for k in (0, 3, 9):
        img = pool[k].cpu().numpy()
        want_px = parity.expected_pixels(fl, gpu_state, oracle, img, w=300, h=200)   # (checked against the oracle in there)
        y, u, v, _a = oracle.webp_yuv420(want_px)
        got = two[firsts[k]]
        assert np.array_equal(got[:60000].reshape(200, 300), y)
        assert np.array_equal(got[60000:75000].reshape(100, 150), u) and np.array_equal(got[75000:90000].reshape(100, 150), v)


@pytest.mark.gpu
def test_config3_shape_on_two_shards(fl, gpu_state, two_shards, oracle):
    # BASELINE config 3's per-GPU shape: >= 1,024 x 1080p, resize + `rgb=` fill, sharded
    import torch
    pool = [torch.from_numpy(synth.uniform(1080, 1920, 3, index=940 + i)).cuda() for i in range(16)]
    order = [(7 * i) % 16 for i in range(1024)]
    params = fl.make_params(300, 200, fill=(12, 200, 77))
    cap = 300 * 200 * 4
    one, _ = _device_batch(fl, gpu_state, pool, order, params, cap)
    two, _ = _device_batch(fl, two_shards, pool, order, params, cap)
    assert np.array_equal(one, two)
    for i in (0, 511, 512, 1023):                                        # both sides of the shard boundary
        img = pool[order[i]].cpu().numpy()
        want = parity.expected_pixels(fl, gpu_state, oracle, img, w=300, h=200, fill=(12, 200, 77))   # within 1 LSB of the reference arithmetic
        got = two[i].reshape(200, 300, 4)
        assert np.array_equal(got, want)
        assert tuple(got[0, 0]) == (12, 200, 77, 255)


@pytest.mark.gpu
def test_config3_true_per_gpu_share_in_one_batch(fl, gpu_state, two_shards, oracle):
    """BASELINE config 3 on 8 GPUs is 65,536 x 1080p = 8,192 pictures per GPU: 51 GB of sources resident on the one device, ONE
    flgpu_transform_batch_device call with an `rgb=` fill (24,576 items on 256 persistent workgroups, a 3 MB descriptor block,
    every picture its own 6.2 MB source -- no 32-bit offset may wrap), then the same over two shards of 4,096.  Pictures from
    all over the batch must equal the same request sent alone."""
    import torch
    n, h, w = 8192, 1080, 1920
    free, _ = torch.cuda.mem_get_info()
    if free < n * h * w * 3 + n * 240128 + (6 << 30):
        pytest.skip("needs 60 GB of free device memory")
    src = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    for k in range(0, n, 512):
        src[k:k + 512].random_(0, 256)
    cap = 300 * 200 * 4
    dst = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
    params = fl.make_params(300, 200, fill=(12, 200, 77))
    picks = [0, 1, 255, 256, 2047, 4095, 4096, 4097, 6000, n - 2, n - 1]
    alone = {k: parity.expected_pixels(fl, gpu_state, oracle, src[k].cpu().numpy(), w=300, h=200, fill=(12, 200, 77)) for k in picks}
    for st in (gpu_state, two_shards):
        dst.zero_()
        st.process_batch_device([src.data_ptr() + k * h * w * 3 for k in range(n)], [(h, w, 3)] * n, params, [dst.data_ptr() + k * cap for k in range(n)], [cap] * n)
        st.batch_results()
        for k in picks:
            assert np.array_equal(dst[k].cpu().numpy().reshape(200, 300, 4), alone[k]), k
        assert int((dst[:, 3] != 255).sum().item()) == 0                # every picture's first pixel is opaque frame: nothing was skipped
    del src, dst
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_host_batches_and_the_request_queue_on_two_shards(fl, gpu_state, two_shards, oracle):
    imgs = [synth.photo(200 + 13 * i, 260 + 7 * i, 3 + (i % 2), index=960 + i) for i in range(24)]
    # (ratios 2 .. 5: the window-tile matrix-pipe kernel and the fused ones side by side; every fifth request with a blur behind it)
    ps = [fl.make_params(100 + i, 80, crop=bool(i % 3 == 0), quality=60 + i, blur_sigma=3.0 if i % 5 == 0 else 0.0,
                         front_end=fl.FE_JPEG if i % 4 == 0 else fl.FE_NONE) for i in range(24)]
    one = gpu_state.process_batch(imgs, ps)
    two = two_shards.process_batch(imgs, ps)
    for a, b in zip(one, two):
        assert (a == b) if isinstance(a, bytes) else np.array_equal(a, b)
    # concurrent callers of flgpu_transform: flushed batches are split between the two device slots
    out = [None] * 96
    def worker(t):
        for i in range(t, 96, 8):
            out[i] = two_shards.process_pixels(imgs[i % 24], ps[i % 24])
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(96):
        a = one[i % 24]
        assert (out[i] == a) if isinstance(a, bytes) else np.array_equal(out[i], a)


@pytest.mark.gpu
def test_cmyk_table_reaches_every_shard(fl, two_shards):
    # shards that share a physical GPU take a copy (1); distinct GPUs would take one RCCL broadcast (2)
    clut = np.random.default_rng(5).integers(0, 65536, (17, 17, 17, 17, 3), dtype=np.uint16)
    two_shards.set_cmyk_clut(clut)
    assert two_shards.cmyk_distribution() == 1
    px = np.random.default_rng(6).integers(0, 256, (1 << 18, 4), dtype=np.uint8)     # large enough to be cut per device
    with fl.State(device=0) as one:
        one.set_cmyk_clut(clut)
        assert one.cmyk_distribution() == 0
        want = one.cmyk_to_rgb(px)
    assert np.array_equal(two_shards.cmyk_to_rgb(px), want)
    assert np.array_equal(two_shards.get_cmyk_clut(), clut)


@pytest.mark.gpu
def test_rccl_path_of_the_table_distribution_on_one_device(fl):
    """A context over several distinct GPUs hands its CMYK table round with one ncclBroadcast through a dlopen-ed RCCL and five
    hand-declared prototypes (csrc/fl_cmyk_ctx.cpp).  No multi-GPU box has run that yet; everything but the wires is proven here
    with one rank: library found, symbols resolved, a communicator from ncclCommInitAll, 250,563 "ncclUint8" elements broadcast
    out of place between group calls arrive as exactly 250,563 bytes, the communicator is destroyed."""
    from conftest import require_device
    require_device()
    r = fl.rccl_selftest(0)
    if r["status"] == fl.ERR_UNSUPPORTED:
        pytest.skip("no librccl on this host: the distribution uses copies")
    assert r["status"] == fl.OK, r
    assert r["bytes_intact"] == 250563 and r["tail_untouched"] and r["communicator_destroyed"], r
    assert r["rccl_version"] > 20000, r
