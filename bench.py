"""bench.py - HP-VAE-GAN train-step throughput on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], the config the metric is quoted on): train_video.py air_balloons 13 frames @
256x144, --vae-levels 3, defaults (nfc 64, latent 128, batch 2, fp32) => 10 pyramid stages.
The mp4 is absent from the reference checkout, so inputs are synthetic U(-1,1) tensors of the exact stage shapes and
weights are random-init (default torch init), as SURVEY.md 8(d) prescribes.

One "step" = one train iteration (train_video.py:111-202) at EVERY pyramid stage 0..9, inputs resident in HBM.
value = stage-iterations per second over the timed region (10*K / seconds); per-stage it/s are reported beside it.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def measured_traffic(family, shape):
    """HBM bytes per launch of a roofline kernel family at `shape`, from the rocprofv3 PMC passes committed under profiles/
    (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, gfx950 x2 fetch correction: tools/pmc_summary.py writes
    profiles/roofline_traffic.json beside the per-kernel csv).  bench.py cannot run the profiler itself; a (family, shape)
    the profile does not cover yields None."""
    try:
        with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, None
    for e in rec.get("entries", []):
        if e.get("family", "conv_fwd") == family and tuple(e["shape"]) == tuple(shape):
            return e["bytes"], e["source"]
    return None, None


# --config: "video" = BASELINE configs[2] (the metric's config), "image" = configs[1] (2-D path), "video8" = configs[3]
# (8 pyramid scales: --min-size 48; with --gpus N the level pipeline, one level per GPU at N = 8), "baseline" = configs[4]
# (train_video_baselines.py --generator GeneratorSG, 8 scales; with --gpus N the stage pipeline)
CONFIG = "video"
WORKLOADS = {
    "video": "train_video air_balloons 13f@256x144 vae_levels=3 B=2 nfc=64 (BASELINE configs[2])",
    "image": "train_image air_balloons.jpg 256x192 vae_levels=3 B=2 nfc=64 (BASELINE configs[1], 2-D path)",
    "video8": "train_video air_balloons 13f@256x144 --min-size 48 => 8 pyramid scales, vae_levels=3 B=2 nfc=64 (BASELINE configs[3])",
    "baseline": "train_video_baselines --generator GeneratorSG 13f@256x144 --min-size 48 => 8 scales, B=2 nfc=64 (BASELINE configs[4])",
}


def video_opt(device, **kw):
    if CONFIG == "image":
        return image_opt(device, **kw)
    if CONFIG == "video8":
        return _video_opt(device, **dict(dict(min_size=48), **kw))
    if CONFIG == "baseline":
        # train_video_baselines.py:216-270 defaults (Dsteps 1, Gsteps 1, alpha 10, train-depth 1) + the config's generator
        return _video_opt(device, **dict(dict(min_size=48, generator="GeneratorSG", Dsteps=1, Gsteps=1, alpha=10.0, nc_z=3), **kw))
    return _video_opt(device, **kw)


def image_opt(device, **kw):
    """BASELINE configs[1]: train_image.py air_balloons.jpg (248x186 -> ar 0.75), --vae-levels 3, 2-D conv path."""
    o = _video_opt(device, **kw)
    o.dims = 2
    o.ar = 186.0 / 248.0
    o.discriminator = "WDiscriminator2D"
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _video_opt(device, **kw):
    o = types.SimpleNamespace(
        dims=3, nc_im=3, nfc=64, latent_dim=128, enc_blocks=2, ker_size=3, num_layer=5, padd_size=1, stride=1,
        vae_levels=3, train_all=False, train_depth=1, scale_factor_init=0.75, min_size=32, max_size=256, img_size=256,
        ar=144.0 / 256.0, sampling_rates=[4, 3, 2, 1], org_fps=24, fps_lcm=12, batch_size=2, lr_g=5e-4, lr_d=5e-4, beta1=0.5,
        lambda_grad=0.1, rec_weight=10.0, kl_weight=1.0, disc_loss_weight=1.0, lr_scale=0.2, grad_clip=5.0,
        noise_amp_init=0.1, const_amp=False, device=device, generator="GeneratorHPVAEGAN", discriminator="WDiscriminator3D",
        niter=1)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def stage_shapes(opt, geom):
    geom.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    return [geom.level_shape(i, opt) for i in range(opt.stop_scale + 1)]


class _HipGeom:
    @staticmethod
    def adjust_scales2image(size, opt):
        from hp_vae_gan_amd import utils as hu
        hu.adjust_scales2image(size, opt)

    @staticmethod
    def level_shape(i, opt):
        from hp_vae_gan_amd import utils as hu
        return hu.images.level_shape_3d(i, opt) if opt.dims == 3 else hu.images.level_shape_2d(i, opt)


def build_gpu_stages(device, stages):
    """One StageTrainer per pyramid stage with synthetic resident inputs (setup is outside the timed region).
    Returns [(stage, trainer, step callable, real, real_zero)]."""
    import copy
    from hp_vae_gan_amd import train as hp_train
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    torch.manual_seed(0)
    base = video_opt(device)
    shapes = stage_shapes(base, _HipGeom)
    baseline = CONFIG == "baseline"
    nets = networks_3d if base.dims == 3 else networks_2d
    proto = getattr(nets, base.generator)(base)
    z_init = None
    out = []
    for s in range(base.stop_scale + 1):
        if s > 0:
            proto.init_next_stage()
        if s not in stages:
            continue
        opt = video_opt(device)
        _HipGeom.adjust_scales2image(opt.img_size, opt)
        opt.stop_scale_time = opt.stop_scale
        opt.scale_idx = s
        opt.Noise_Amps = [1] + [0.05] * max(0, s - 1)
        netG = copy.deepcopy(proto)
        netG.opt = opt
        netG.to(device)
        g = torch.Generator().manual_seed(100 + s)
        real = (torch.rand(opt.batch_size, 3, *shapes[s], generator=g) * 2 - 1).to(device)
        real_zero = (torch.rand(opt.batch_size, 3, *shapes[0], generator=g) * 2 - 1).to(device) if s > 0 else real
        if baseline:
            # fixed reconstruction noise of the whole run (train_video_baselines.py:39-44), the same tensor at every stage
            if z_init is None:
                z_init = torch.randn(opt.batch_size, 3, *shapes[0], generator=torch.Generator().manual_seed(99)).to(device)
            opt.Z_init = z_init
            trainer = hp_train.BaselineStageTrainer(opt, netG)
            step = (lambda tr, r: (lambda: tr.step(r)))(trainer, real)
        else:
            trainer = hp_train.StageTrainer(opt, netG)
            step = (lambda tr, r, rz: (lambda: tr.step(r, rz)))(trainer, real, real_zero)
        out.append((s, trainer, step, real, real_zero))
    return out, shapes


def cpu_iters(s):
    """timed iterations of the CPU leg at stage s (after one warm-up iteration): BASELINE.md section 4's plan - every stage of
    the sweep is RUN (no extrapolation): 20 iterations at stages 0-3, 3 at 4-7, 1 at 8-9 (~3 minutes on the box's 16 threads,
    most of it the two finest stages)"""
    return 20 if s <= 3 else (3 if s <= 7 else 1)


def cpu_baseline(stages, threads):
    """The CPU oracle (port of the reference path, oracle/hpvg_oracle.py) timed on the host: one warm-up, then cpu_iters(s)
    timed iterations per sampled stage, same shapes / hyper-parameters, torch CPU fp32 with `threads` threads.
    Returns {stage: mean seconds per iteration}."""
    from oracle import hpvg_oracle as O
    from hp_vae_gan_amd.modules import networks_3d
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    base = video_opt("cpu")
    O.adjust_scales2image(base.img_size, base)
    base.stop_scale_time = base.stop_scale
    proto = networks_3d.GeneratorHPVAEGAN(base)  # parameter containers only (state_dict layout); compute is the oracle's
    per_stage = {}
    for s in range(max(stages) + 1):
        if s > 0:
            proto.init_next_stage()
        if s not in stages:
            continue
        opt = video_opt("cpu")
        O.adjust_scales2image(opt.img_size, opt)
        opt.stop_scale_time = opt.stop_scale
        PG = {k: v.clone() for k, v in proto.state_dict().items()}
        for k in PG:
            if O.is_param(k):
                PG[k].requires_grad_(True)
        PD = None
        if opt.vae_levels < s + 1:
            D = networks_3d.WDiscriminator3D(opt)
            PD = {k: v.clone() for k, v in D.state_dict().items()}
            for k in PD:
                if O.is_param(k):
                    PD[k].requires_grad_(True)
        g = torch.Generator().manual_seed(100 + s)
        shp = O.level_shape(s, opt, 3)
        real = torch.rand(opt.batch_size, 3, *shp, generator=g) * 2 - 1
        real_zero = torch.rand(opt.batch_size, 3, *O.level_shape(0, opt, 3), generator=g) * 2 - 1 if s > 0 else real
        amps = [1] + [0.05] * s
        zshape = [opt.batch_size, opt.latent_dim, *O.level_shape(0, opt, 3)]
        adam_g, adam_d = {}, {}

        def one_iter():
            ni = torch.randn(zshape)
            shapes_needed = [zshape] + [[opt.batch_size, 3, *O.level_shape(i, opt, 3)] for i in range(1, s + 1) if opt.vae_levels <= i]
            noises = iter([torch.randn(x) for x in shapes_needed])
            return O.train_step(PG, PD, opt, 3, s, real, real_zero, ni, noises, torch.rand(()), amps, adam_g, adam_d)

        one_iter()  # warm-up
        n = cpu_iters(s)
        t0 = time.perf_counter()
        for _ in range(n):
            one_iter()
        per_stage[s] = (time.perf_counter() - t0) / n
    return per_stage


def parallelism_mode():
    """N > 1 schedule: HPVG_PARALLELISM=levels|schedules; default: the level pipeline for configs[3] (the partition that
    config names), the pass / sample / row-slab schedules of multigpu.py for configs[2]."""
    env = os.environ.get("HPVG_PARALLELISM", "")
    if env:
        return env
    return "levels" if CONFIG == "video8" else "schedules"


def _parallelism(world):
    """What the ranks do (hp_vae_gan_amd/multigpu.py); the stage thresholds are the env knobs the runner reads."""
    if world == 1:
        return "single GPU"
    if CONFIG == "baseline":
        return ("stage pipeline: contiguous GeneratorSG stages per rank (frozen stages forward-only; the newest stage + D on the last), "
                "stage outputs sent forward point to point (%d ranks)" % world)
    if parallelism_mode() == "levels":
        return "level pipeline: contiguous pyramid levels per rank (finest level + D on the last), level outputs sent forward, gradients back (%d ranks)" % world
    vae = "VAE stages: rank 0 alone (hipGraph replay); "
    pair = vae + "GAN stages: rec/rand generator passes on ranks 0/1 + discriminator work split over the batch"
    if world < 4:
        return "%s (2 working ranks of %d)" % (pair, world)
    qmin = os.environ.get("HPVG_QUAD_MIN_STAGE", "5")
    quad = vae + "stages >= %s: generator passes x batch samples on 4 ranks with batch-split BatchNorm, discriminator work by sample and task" % qmin
    if world < 8:
        return "%s; earlier GAN stages: 2 ranks (4 working ranks of %d)" % (quad, world)
    omin = os.environ.get("HPVG_OCT_MIN_STAGE", "7")
    return ("stages >= %s: the 4 (pass, sample) jobs x 2 row slabs on 8 ranks (boundary-row swaps per conv, BatchNorm over samples and "
            "slabs); %s; earlier GAN stages: 2 ranks (8 working ranks of %d)" % (omin, quad[len(vae):], world)).join([vae, ""])


def comm_check(world, rank, device, backend, dev_index):
    """Proof, in the JSON line, of what the run's communicator was: one all_reduce of ones over the process group the
    schedules use (RCCL when backend == "nccl") must count `world` ranks, and every rank reports the device it sits on.
    N = 1: no communicator."""
    if world == 1:
        return {"backend": None, "ranks": 1, "devices": [dev_index], "distinct_devices": 1}
    one = torch.ones(1, device=device if backend == "nccl" else "cpu", dtype=torch.float32)
    dist.all_reduce(one)
    ranks = int(round(float(one.item())))
    devs = [None] * world
    name = torch.cuda.get_device_properties(dev_index).name if torch.cuda.is_available() else "cpu"
    dist.all_gather_object(devs, (rank, dev_index, name))
    devs.sort()
    if ranks != world:
        raise RuntimeError("the %s communicator counted %d ranks, the launch has %d" % (backend, ranks, world))
    return {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "ranks": ranks,
            "devices": [d[1] for d in devs], "distinct_devices": len({d[1] for d in devs}), "device_name": devs[0][2]}


# roofline kernel families of a step (ops.KernelTimer descriptors -> family name)
def roofline_family(g, KT):
    if g["KT"] != KT:
        return None
    if g["op"] == "wgrad":
        return "weight_gradient" if g["Cin"] == 64 and g["Cout"] == 64 else None
    if g["flip"] or g["var"] != "plain":
        return None          # (the plain template instance: forward convs of the generator blocks)
    if g["Cin"] == 64 and g["Cout"] == 64:
        return "conv_fwd"
    if g["Cin"] <= 4 and g["Cout"] == 64:
        return "head_fwd"
    if g["Cin"] == 64 and g["Cout"] <= 4:
        return "tail_fwd"
    return None


CONV_KIND = {0: ("conv_mfma_kernel + conv_fixup_kernel (direct implicit GEMM, stream-K)", 1.0),
             1: ("conv_wino_kernel + conv_wino_fixup_kernel (Winograd F(2,3) along W, stream-K)", 2.0 / 3.0),
             2: ("conv_wino2r_kernel (Winograd F(2x2,3x3) over H and W, one software-pipelined workgroup per CU, a point row per wave)", 4.0 / 9.0),
             3: ("conv_narrow2_kernel (narrow output: taps in the GEMM N axis)", 1.0)}
WGRAD_KIND = {0: ("conv_wgrad_kernel + conv_wgrad_reduce_kernel (direct, a workgroup per time tap)", 1.0),
              1: ("conv_wgrad3_kernel + conv_wgrad3_reduce_kernel (direct, all taps per workgroup)", 1.0),
              2: ("conv_wgradw_kernel + conv_wgradw_reduce_kernel (transposed Winograd F(2,3) along W)", 2.0 / 3.0),
              3: ("conv_wgradw2_kernel + conv_wgradw2_reduce_kernel (transposed Winograd F(2x2,3x3) over H and W)", 4.0 / 9.0),
              4: ("conv_wgrad_narrow2_kernel + conv_wgrad_narrow_reduce_kernel", 1.0)}


def roofline_entries(by_shape, KT, lib):
    """One roofline entry per kernel family from the live HIP-event timings {(family, B, Cin, Cout, T, H, W): (ms, launches, rank)}.
    MFMA-bound families (64 -> 64 convs and their weight gradient): `achieved` = matrix-core flops the kernel EXECUTES per
    launch / average launch time, `frac` = achieved / the fp32 MFMA peak (<= 1 by construction); the conv's algorithmic flops
    (2 * Cin * Cout * taps per output voxel) are reported beside it as `algorithmic_tflops` with `work_ratio` =
    executed / algorithmic (1 direct, 2/3 one-axis Winograd, 4/9 two-axis).  HBM-bound families (3 -> 64 heads, 64 -> 3 / 1
    tails): `achieved` = algorithmic bytes / time in GB/s against the 8 TB/s HBM peak.  The returned object is the forward
    conv family's entry (the step's dominant kernel) with every family under "families"."""
    fams = {}
    for k in by_shape:
        fams.setdefault(k[0], []).append(k)
    out = {}
    taps = 9 * KT
    for fam, keys in fams.items():
        # finest level; among its launches the per-pass batch B = batch_size when some rank ran it (the shape the PMC traffic
        # figures are measured on, and the shape an N = 1 run times: the merged generator pass runs the same kernel at B = 4,
        # the batch-split schedules at B = 1)
        key = max(keys, key=lambda k: (k[4] * k[5] * k[6], k[1] == 2, -k[1], k[3]))
        ms, n, owner = by_shape[key]
        _, B, Cin, Cout, T, H, W = key
        shape5 = [B, Cin, T, H, W]
        vox = float(B) * T * H * W
        sec = ms * 1e-3
        traffic, traffic_src = measured_traffic(fam, shape5)
        if fam in ("conv_fwd", "weight_gradient"):
            if fam == "conv_fwd":
                kind = lib.hpvg_conv_fwd_kernel_kind(B, Cin, Cout, T, H, W, KT)
                kname, ratio = CONV_KIND[kind]
            else:
                kind = lib.hpvg_conv_bwd_weight_kernel_kind(B, Cin, Cout, T, H, W, KT)
                kname, ratio = WGRAD_KIND[kind]
            flops = 2.0 * vox * Cin * Cout * taps
            executed = flops * ratio
            achieved = executed / sec / 1e12
            ent = {"bound": "mfma", "kernel": "%s, 64->64 %s, fp32 v_mfma_f32_32x32x2_f32" % (kname, "3x3x3" if KT == 3 else "3x3"),
                   "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                   "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                   "algorithmic_tflops": round(flops / sec / 1e12, 3), "work_ratio": round(ratio, 4),
                   "flops_per_launch": flops, "executed_flops_per_launch": executed,
                   "algorithmic_bytes": 4.0 * vox * (Cin + Cout) + 4.0 * Cin * Cout * taps,
                   "hbm_frac_of_algorithmic_bytes": round((4.0 * vox * (Cin + Cout)) / sec / 1e9 / PEAK_HBM_GBS, 4)}
        else:
            kind = lib.hpvg_conv_fwd_kernel_kind(B, Cin, Cout, T, H, W, KT)
            kname = CONV_KIND[kind][0] if fam == "tail_fwd" else "conv_mfma_kernel<4, ...> (3-channel input chunk) + conv_fixup_kernel"
            nbytes = 4.0 * vox * (Cin + Cout) + 4.0 * Cin * Cout * taps
            flops = 2.0 * vox * Cin * Cout * taps
            achieved = nbytes / sec / 1e9
            ent = {"bound": "hbm", "kernel": "%s, %d->%d %s" % (kname, Cin, Cout, "3x3x3" if KT == 3 else "3x3"),
                   "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4),
                   "algorithmic_bytes": nbytes, "flops_per_launch": flops, "flop_per_byte": round(flops / nbytes, 2),
                   "mfma_frac": round(flops / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
        ent.update({"family": fam, "traffic": traffic, "traffic_source": traffic_src, "shape": shape5, "cout": Cout,
                    "avg_ms": round(ms, 4), "launches": n, "rank": owner})
        out[fam] = ent
    if not out:
        return None
    order = ["conv_fwd", "weight_gradient", "head_fwd", "tail_fwd"]
    main = dict(out.get("conv_fwd") or out[sorted(out, key=order.index)[0]])
    main["families"] = [out[f] for f in order if f in out]
    return main


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--stages", type=str, default="all", help="pyramid stages in a step, e.g. 0-9 or 9 (default: every stage of the config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stages", type=str, default="all",
                    help="stages the CPU leg times (default: every stage of the step; 1 warm-up + 20 timed iterations at stages "
                         "0-3, 3 at 4-7, 1 at 8-9; nothing is extrapolated - a stage left out here is reported as not run)")
    ap.add_argument("--graph-stages", type=str, default="0-8",
                    help="stages whose iteration is captured once and replayed as a hipGraph (single-GPU path; the stage "
                         "holding the roofline kernel always stays eager so that its launches can be bracketed by events); "
                         "'none' = every stage eager.  Measured r01: +52 % at stage 0, +8 % at stage 3, +2 % at stage 7")
    ap.add_argument("--config", choices=["video", "image", "video8", "baseline"], default="video",
                    help="video = BASELINE configs[2] (the metric's config, default); image = configs[1] (2-D path); video8 = "
                         "configs[3] (8 pyramid scales, --min-size 48; N > 1: level pipeline, one level per GPU at N = 8); "
                         "baseline = configs[4] (train_video_baselines GeneratorSG, 8 scales; N > 1: stage pipeline)")
    args = ap.parse_args()
    global CONFIG
    CONFIG = args.config

    def parse(r):
        a, _, b = r.partition("-")
        return list(range(int(a), int(b or a) + 1))

    if args.stages == "all":
        from hp_vae_gan_amd import utils as _hu
        _o = video_opt("cpu")
        _hu.adjust_scales2image(_o.img_size, _o)
        args.stages = "0-%d" % _o.stop_scale
    stages = parse(args.stages)
    # keep torch's CPU thread pool small: the host side only launches kernels; a 128-thread OpenMP pool spinning after
    # tiny CPU ops exhausts the box's CPU share and the launching thread gets throttled for ~75 ms at a time
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # rehearsal on a one-GPU box: HPVG_DIST_BACKEND=gloo puts every rank on cuda:0 and stages messages through the host
    backend = os.environ.get("HPVG_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import hp_vae_gan_amd  # noqa: F401  (fails loudly if libhpvg.so is missing)
    from hp_vae_gan_amd import ops, lib as hplib
    comm = comm_check(world, rank, device, backend, dev_index)

    if world > 1:
        from hp_vae_gan_amd import multigpu
        runner = multigpu.build_bench_runner(video_opt, stages, device, rank, world, config=CONFIG, mode=parallelism_mode())
    else:
        built, shapes = build_gpu_stages(device, stages)

        class Runner:
            last = {}

            def timed_stage(self, idx):
                s, trainer, step, real, real_zero = built[idx]
                self.last[s] = step()
                return s

            def check_finite(self):
                for s, out in self.last.items():
                    for k, v in (out or {}).items():
                        if torch.is_tensor(v) and v.numel() <= 2 and not bool(torch.isfinite(v).all()):
                            raise RuntimeError("stage %d: %s is not finite after the timed iterations: %s" % (s, k, v))

            n = len(built)
        runner = Runner()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # hipGraph replay for the launch-bound small stages (single-GPU path): one eager iteration (noise-amplitude
    # calibration, workspace sizing), then capture.  The stage holding the roofline kernel stays eager so that the
    # kernel's launches can be bracketed by events.
    graph_stages = [] if args.graph_stages == "none" else [g for g in parse(args.graph_stages) if g < max(stages)]
    if world == 1:
        for s, trainer, step, real, real_zero in built:
            if s in graph_stages and s != max(stages) and hasattr(trainer, "enable_graph"):
                step()
                trainer.enable_graph(real, real_zero)
    # Stage-major order, as training proceeds (train_video.py:414-417: each stage runs its iterations before the next
    # stage starts): W warm-up then K timed iterations of every stage = K "steps" of one iteration per stage.
    for i in range(runner.n):
        for _ in range(args.warmup):
            runner.timed_stage(i)
    barrier()
    # ---- timed region: exactly K steps; the dominant kernel's launches are bracketed by HIP events on the launch stream
    KT = 1 if CONFIG == "image" else 3
    timer = ops.KernelTimer(match=lambda g: roofline_family(g, KT))
    ops.set_kernel_timer(timer)
    stage_ev = {}
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(runner.n):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(args.steps):
            s = runner.timed_stage(i)
        e1.record()
        stage_ev[s] = (e0, e1)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    ops.set_kernel_timer(None)
    # outside the timed region: the iterations just timed must have produced finite losses (a fast run that trained
    # NaNs - as hipGraph replays once did, DESIGN.md section 4 - is not a measurement)
    runner.check_finite()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the roofline kernels' launches, from every rank: on N > 1 the finest level's convs run on the rank(s) that own that
    # level (the last rank of a pipeline), not necessarily on rank 0
    by_shape = {k: v + (rank,) for k, v in timer.summary().items()}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, {tuple(k): v for k, v in by_shape.items()})
        by_shape = {}
        for d in gathered:
            for k, v in d.items():
                if k not in by_shape or v[1] > by_shape[k][1]:
                    by_shape[k] = v

    if rank == 0:
        nstage = len(stages)
        per_stage = {str(s): 1000.0 * args.steps / ev[0].elapsed_time(ev[1]) for s, ev in stage_ev.items()}
        roof = roofline_entries(by_shape, KT, hplib.load())
        cpu = None
        if not args.no_cpu_baseline and world == 1 and CONFIG == "video":
            cs = stages if args.cpu_stages == "all" else [s for s in parse(args.cpu_stages) if s in stages]
            if cs:
                threads = min(os.cpu_count() or 1, 16)
                per = cpu_baseline(cs, threads)
                tot = sum(per.values())
                gpu_same = sum(1.0 / per_stage[str(s)] for s in cs)
                cpu = {"value": round(len(cs) / tot, 4), "unit": "stage-iterations/s", "cores": threads, "kind": "port",
                       "sample": ("oracle train step (oracle/hpvg_oracle.py, torch CPU fp32, %d threads) at stages %s, same shapes as the GPU "
                                  "run (B=2): 1 warm-up + %s timed iterations per stage, every listed stage RUN (nothing extrapolated)" %
                                  (threads, cs, {s: cpu_iters(s) for s in cs})),
                       "per_stage_it_s": {str(s): round(1.0 / t, 5) for s, t in per.items()},
                       "stages_not_run": [s for s in stages if s not in per],
                       "seconds": round(sum(t * (cpu_iters(s) + 1) for s, t in per.items()), 1),
                       "gpu_value_same_sample": round(len(cs) / gpu_same, 3)}
        line = {
            "metric": "train iters/sec per pyramid scale, air_balloons 13f@144p",
            "value": round(nstage * args.steps / elapsed, 4), "unit": "stage-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[CONFIG] + "; step = 1 train iteration at each pyramid stage %s" % args.stages,
                       "stages": stages,
                       "hipgraph_stages": ([s for s in graph_stages if s in stages] if world == 1 else
                                           [s for s in stages if s < 3 and os.environ.get("HPVG_VAE_ON_RANK0", "1") != "0"
                                            and parallelism_mode() != "levels" and CONFIG != "baseline"]),
                       "parallelism": _parallelism(world)},
            "per_stage_it_s": {k: round(v, 4) for k, v in sorted(per_stage.items(), key=lambda kv: int(kv[0]))},
            "roofline": roof, "cpu_baseline": cpu, "comm": comm,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
