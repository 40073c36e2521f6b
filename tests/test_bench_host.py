"""Host logic of bench.py that needs no GPU: the communicator proof of the JSON line (gloo, world 2) and the roofline
arithmetic (executed vs algorithmic flops, one entry per kernel family, every fraction <= 1 for a physically possible time)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _comm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    c = bench.comm_check(world, rank, torch.device("cpu"), "gloo", rank)
    q.put((rank, c))
    dist.destroy_process_group()


def test_comm_field_counts_the_ranks_of_the_process_group():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        c = got[r]
        assert c["ranks"] == world and c["backend"] == "gloo" and c["devices"] == [0, 1] and c["distinct_devices"] == 2


def test_comm_field_single_process():
    import bench
    c = bench.comm_check(1, 0, torch.device("cpu"), "nccl", 0)
    assert c == {"backend": None, "ranks": 1, "devices": [0], "distinct_devices": 1}


class _Lib:
    def __init__(self, conv_kind, wgrad_kind):
        self.ck, self.wk = conv_kind, wgrad_kind

    def hpvg_conv_fwd_kernel_kind(self, *a):
        return self.ck

    def hpvg_conv_bwd_weight_kernel_kind(self, *a):
        return self.wk


@pytest.mark.parametrize("conv_kind,wgrad_kind", [(0, 0), (1, 2), (2, 2), (2, 3)])
def test_roofline_entries_price_executed_flops(conv_kind, wgrad_kind):
    import bench
    shape = (2, 64, 64, 13, 144, 256)
    alg = 2.0 * 2 * 13 * 144 * 256 * 64 * 64 * 27
    by_shape = {("conv_fwd",) + shape: (1.05, 25, 0), ("weight_gradient",) + shape: (1.33, 30, 0),
                ("conv_fwd", 4, 64, 64, 13, 144, 256): (2.08, 10, 0),      # the merged generator pass: not the N = 1 shape
                ("conv_fwd", 2, 64, 64, 7, 114, 204): (0.35, 10, 0),       # a coarser level
                ("head_fwd", 2, 3, 64, 13, 144, 256): (0.21, 5, 0), ("tail_fwd", 2, 64, 3, 13, 144, 256): (0.184, 5, 0)}
    roof = bench.roofline_entries(by_shape, 3, _Lib(conv_kind, wgrad_kind))
    fams = {e["family"]: e for e in roof["families"]}
    assert list(fams) == ["conv_fwd", "weight_gradient", "head_fwd", "tail_fwd"]
    assert roof["family"] == "conv_fwd" and roof["shape"] == [2, 64, 13, 144, 256] and roof["avg_ms"] == 1.05
    ratio = {0: 1.0, 1: 2 / 3, 2: 4 / 9}[conv_kind]
    c = fams["conv_fwd"]
    assert abs(c["flops_per_launch"] - alg) < 1 and abs(c["executed_flops_per_launch"] - alg * ratio) < 1e3
    assert abs(c["achieved"] - alg * ratio / 1.05e-3 / 1e12) < 1e-2 and abs(c["frac"] - c["achieved"] / 157.3) < 1e-3
    assert abs(c["algorithmic_tflops"] - alg / 1.05e-3 / 1e12) < 1e-2
    w = fams["weight_gradient"]
    wr = {0: 1.0, 2: 2 / 3, 3: 4 / 9}[wgrad_kind]
    assert abs(w["work_ratio"] - wr) < 1e-3 and abs(w["achieved"] - alg * wr / 1.33e-3 / 1e12) < 1e-2
    # a direct kernel at these times would be above the peak - the point of pricing executed flops is that Winograd is not
    for e in (c, w):
        if e["work_ratio"] < 1:
            assert e["frac"] < 1.0 and e["algorithmic_tflops"] > e["achieved"]
    h, t = fams["head_fwd"], fams["tail_fwd"]
    assert h["bound"] == "hbm" and t["bound"] == "hbm" and h["unit"] == "GB/s"
    vox = 2 * 13 * 144 * 256
    assert abs(h["achieved"] - (4.0 * vox * 67 + 4 * 3 * 64 * 27) / 0.21e-3 / 1e9) < 1 and 0 < h["frac"] < 1 and 0 < t["frac"] < 1
    assert abs(t["flop_per_byte"] - 2.0 * vox * 64 * 3 * 27 / (4.0 * vox * 67 + 4 * 3 * 64 * 27)) < 0.01
