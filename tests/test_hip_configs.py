"""Every BASELINE config bench.py names runs on one GPU: configs[2] and configs[1] (ten scales), configs[3] (8 pyramid scales, --min-size 48) and configs[4]
(train_video_baselines GeneratorSG, 8 scales) at their full sizes, one iteration per stage, with the checks bench.py
makes (finite losses) and the size-independent ones below; plus 2-rank rehearsals of their N > 1 pipelines."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, nproc=1, env=None):
    e = dict(os.environ)
    e.update(env or {})
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                "--master-port", "29611"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_roofline(line, KT):
    """every fraction of the line is a fraction of a roof: executed matrix-core flops (or algorithmic bytes) per second over
    the peak, <= 1 by construction; the algorithmic rate of a Winograd kernel is a separate key and may exceed the peak"""
    roof = line["roofline"]
    fams = {e["family"]: e for e in roof["families"]}
    assert roof["family"] == "conv_fwd" and "conv_fwd" in fams and "weight_gradient" in fams
    for e in roof["families"]:
        assert 0.0 < e["frac"] < 1.0, e
        assert abs(e["frac"] - e["achieved"] / e["peak"]) < 2e-3
        if e["bound"] == "mfma":
            assert e["executed_flops_per_launch"] <= e["flops_per_launch"] and e["work_ratio"] in (1.0, 0.6667, 0.4444)
            assert abs(e["algorithmic_tflops"] * e["work_ratio"] - e["achieved"]) < 0.05 * e["achieved"]
            assert e["flops_per_launch"] == 2.0 * e["shape"][0] * e["shape"][2] * e["shape"][3] * e["shape"][4] * 64 * 64 * 9 * KT
        else:
            assert e["unit"] == "GB/s" and e["peak"] == 8000.0
    assert 0.2 < fams["conv_fwd"]["frac"] and 0.2 < fams["weight_gradient"]["frac"]
    assert line["comm"]["ranks"] == line["n_gpus"]
    return fams


@pytest.mark.parametrize("config", ["video", "image"])
def test_ten_scale_configs_run_on_one_gpu(config):
    """configs[2] (the metric's config) and configs[1] (the 2-D path): one step, every stage, the roofline families of the line"""
    line = _bench(["--config", config, "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert line["config"]["stages"] == list(range(10))
    its = [line["per_stage_it_s"][str(s)] for s in range(10)]
    assert all(v > 0 for v in its) and its[9] < its[5] < its[0]
    fams = _check_roofline(line, 3 if config == "video" else 1)
    W = 256
    for e in fams.values():
        assert e["shape"][0] in (2, 4) and e["shape"][4] == W and e["shape"][2] == (13 if config == "video" else 1)
    # the 3 -> 64 heads and 64 -> 3 tails are the HBM-bound convs of the path (12.9 / 4.4 FLOP/B in 2-D)
    assert fams["head_fwd"]["bound"] == "hbm" and fams["tail_fwd"]["bound"] == "hbm"
    if config == "video":
        assert fams["conv_fwd"]["traffic"] is not None and fams["weight_gradient"]["traffic"] is not None


@pytest.mark.parametrize("config", ["video8", "baseline"])
def test_eight_scale_configs_run_on_one_gpu(config):
    line = _bench(["--config", config, "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert line["config"]["stages"] == list(range(8)), "--min-size 48 gives 8 pyramid scales (SURVEY Appendix A)"
    assert sorted(line["per_stage_it_s"], key=int) == [str(s) for s in range(8)]
    assert all(v > 0 for v in line["per_stage_it_s"].values())
    its = [line["per_stage_it_s"][str(s)] for s in range(8)]
    assert its[7] < its[3] < its[0], "iterations get slower as the pyramid grows"
    roof = line["roofline"]
    # the dominant kernel ran at the finest stage: 13 frames at 256 wide (the baseline's valid convs run on padded volumes)
    assert roof["shape"][0] in (2, 4) and roof["shape"][2] >= 13 and roof["shape"][4] >= 256 and 0.2 < roof["frac"] < 1.0
    _check_roofline(line, 3)


@pytest.mark.parametrize("config", ["video8", "baseline"])
def test_eight_scale_pipelines_rehearse_on_two_ranks(config):
    """The N > 1 path of both configs (level / stage pipeline) on two processes sharing the card (gloo, host-staged
    messages), stages 2-4: finite losses on every stage, one JSON line, the roofline taken from the rank that owns the
    finest level."""
    line = _bench(["--gpus", "2", "--config", config, "--steps", "1", "--warmup", "1", "--stages", "2-4"], nproc=2,
                  env={"HPVG_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and "pipeline" in line["config"]["parallelism"]
    assert all(v > 0 for v in line["per_stage_it_s"].values())
    assert line["roofline"]["rank"] == 1 and line["roofline"]["shape"][0] == 2
    assert line["comm"] == {"backend": "gloo", "ranks": 2, "devices": [0, 0], "distinct_devices": 1, "device_name": line["comm"]["device_name"]}
