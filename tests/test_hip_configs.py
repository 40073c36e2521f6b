"""Every BASELINE config bench.py names runs on one GPU: configs[3] (8 pyramid scales, --min-size 48) and configs[4]
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
    # (`frac` prices the conv's ALGORITHMIC flops: the Winograd kernel executes 2/3 of them, so it may pass 1; what the
    # matrix cores actually ran stays under their peak)
    assert roof["shape"][0] in (2, 4) and roof["shape"][2] >= 13 and roof["shape"][4] >= 256 and 0.2 < roof["frac"] < 1.5
    assert 0.2 < roof["matrix_pipe_frac"] < 1.0 and roof["executed_flops_per_launch"] <= roof["flops_per_launch"]


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
