"""N > 1 path of bench.py (one process per GPU, torch.distributed).

CPU part (gloo, world size 2, no device): the shard map, the max-over-ranks timing contract, and the rule that a rank
which disagrees about the planned tables takes every rank down together (an all-reduced flag), instead of leaving its
peers waiting in a broadcast.

GPU part (gloo, two ranks sharing the box's one GPU): each rank opens a real context, plans the bench batch, rank 0's
table blob goes out through flgpu_copy_tables -> broadcast -> flgpu_import_tables, rank 0's baked CMYK table through
flgpu_get_cmyk_clut -> broadcast -> flgpu_set_cmyk_clut; both ranks then produce identical pixels and conversions.
These are the very calls bench.py issues over RCCL/xGMI on a multi-GPU node."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(target, world, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=300) for _ in range(world))
    [p.join(timeout=120) for p in procs]
    return res, [p.exitcode for p in procs]


def _agree(rank, world, port, out, sizes):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    lo, hi = bench.shard_batches(2048, world)[rank]
    nbytes = sizes[rank]                                   # what this rank planned
    size_t = torch.tensor([nbytes], dtype=torch.int64)
    dist.broadcast(size_t, src=0)
    same = torch.tensor([int(int(size_t.item()) == nbytes)], dtype=torch.int64)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)            # bench.py: every rank learns the verdict before moving on
    agreed = int(same.item())
    blob0 = -1
    if agreed:
        blob = torch.full((nbytes,), rank + 1, dtype=torch.uint8)
        dist.broadcast(blob, src=0)
        blob0 = int(blob[0])
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out.put((rank, lo, hi, agreed, blob0, float(t[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_agree_on_tables_and_timing():
    res, codes = _spawn(_agree, 2, (1000, 1000))
    assert codes == [0, 0]
    assert [(r[1], r[2]) for r in res] == [(0, 1024), (1024, 2048)]      # disjoint, covering shards
    assert all(r[3] == 1 and r[4] == 1 for r in res)                      # every rank holds rank 0's blob
    assert all(r[5] == 2.0 for r in res)                                  # elapsed = max over ranks


def test_a_rank_with_different_tables_stops_every_rank_together():
    res, codes = _spawn(_agree, 2, (1000, 1001))
    assert codes == [0, 0]                                                # nobody hangs in a broadcast
    assert all(r[3] == 0 and r[4] == -1 for r in res)                     # both saw the mismatch, neither broadcast


def test_shard_map_properties():
    import bench
    for world in (1, 2, 4, 8):
        s = bench.shard_batches(1024 * world, world)
        assert len(s) == world and s[0][0] == 0 and s[-1][1] == 1024 * world
        assert all(a[1] == b[0] for a, b in zip(s, s[1:])) and all(hi - lo == 1024 for lo, hi in s)


# ---------------------------------------------------------------------------------- GPU: real contexts, real tables --

def _gpu_rank(rank, world, port, out):
    import hashlib
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_package
    import synth
    fl = load_package()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    st = fl.State(device=0)
    # the same request mix on every rank (tables are planned from it), different pixels per rank
    params = [fl.make_params(300, 200), fl.make_params(300, 200, crop=True), fl.make_params(160, 90, grayscale=True, blur_sigma=10.0)]
    common = torch.from_numpy(synth.photo(360, 640, 3, index=5)).to(dev)
    own = torch.from_numpy(synth.photo(360, 640, 3, index=50 + rank)).to(dev)
    n = 6
    dst = torch.zeros((n, 300 * 200 * 4), dtype=torch.uint8, device=dev)
    srcs = [common.data_ptr()] * 3 + [own.data_ptr()] * 3
    st.process_batch_device(srcs, [(360, 640, 3)] * n, params * 2, [dst.data_ptr() + i * dst.shape[1] for i in range(n)], [dst.shape[1]] * n)
    torch.cuda.synchronize()
    before = hashlib.sha256(dst[:3].cpu().numpy().tobytes()).hexdigest()
    # rank 0's table blob -> every rank (bench.py's sequence, gloo instead of RCCL)
    blob = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
    nbytes = st.copy_tables(blob.data_ptr(), blob.numel())
    size_t = torch.tensor([nbytes], dtype=torch.int64)
    dist.broadcast(size_t, src=0)
    same = torch.tensor([int(int(size_t.item()) == nbytes)], dtype=torch.int64)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    payload = blob[:nbytes].cpu()
    table_digest_local = hashlib.sha256(payload.numpy().tobytes()).hexdigest()
    payload.numpy().tofile(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"fl_tables_rank{rank}.bin"))   # (for the diagnosis below)
    dist.broadcast(payload, src=0)
    blob[:nbytes] = payload.to(dev)
    torch.cuda.synchronize()
    st.import_tables(blob.data_ptr(), nbytes)
    dst.zero_()
    st.process_batch_device(srcs, [(360, 640, 3)] * n, params * 2, [dst.data_ptr() + i * dst.shape[1] for i in range(n)], [dst.shape[1]] * n)
    torch.cuda.synchronize()
    after = hashlib.sha256(dst[:3].cpu().numpy().tobytes()).hexdigest()
    # the CMYK -> sRGB device-link table: made on rank 0 only, broadcast, installed everywhere
    lut = torch.zeros(17 ** 4 * 3, dtype=torch.int32)
    if rank == 0:
        st.set_cmyk_clut(np.random.default_rng(11).integers(0, 65536, (17, 17, 17, 17, 3), dtype=np.uint16))
        lut = torch.from_numpy(st.get_cmyk_clut().astype(np.int32).reshape(-1))
    dist.broadcast(lut, src=0)
    st.set_cmyk_clut(lut.numpy().astype(np.uint16))
    px = np.random.default_rng(7).integers(0, 256, (1 << 15, 4), dtype=np.uint8)
    rgb = hashlib.sha256(st.cmyk_to_rgb(px).tobytes()).hexdigest()
    out.put((rank, int(same.item()), nbytes, table_digest_local, before, after, rgb, hashlib.sha256(dst[3:].cpu().numpy().tobytes()).hexdigest()))
    st.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_share_rank0_tables_and_clut_on_the_device():
    from conftest import require_device
    require_device()
    res, codes = _spawn(_gpu_rank, 2)
    assert codes == [0, 0]
    r0, r1 = res
    assert r0[1] == 1 and r1[1] == 1 and r0[2] == r1[2] > 4096           # both planned tables of the same size ...
    if r0[3] != r1[3]:                                                     # say WHERE the two arenas differ before failing
        import numpy as np
        d = os.environ.get("TMPDIR", "/tmp")
        a, b = (np.fromfile(os.path.join(d, f"fl_tables_rank{k}.bin"), np.uint32) for k in (0, 1))
        w = np.nonzero(a != b)[0]
        print("table blobs differ in", len(w), "words; first:", [(int(i), hex(int(a[i])), hex(int(b[i]))) for i in w[:24]])
    assert r0[3] == r1[3]                                                  # ... and, built independently, the same bytes
    assert r0[4] == r0[5] == r1[4] == r1[5]                                # common picture: same pixels before and after the import, on both ranks
    assert r0[6] == r1[6]                                                  # conversions through the broadcast CLUT agree
    assert r0[7] != r1[7]                                                  # (each rank really processed its own pictures too)
