"""Diagnostic (not a test): per-quantity error report of the HIP train step vs a golden stage fixture."""
import sys
import torch
from helpers import flat_to_named, load_golden, rel_err, run_hip_stage

fname = sys.argv[1] if len(sys.argv) > 1 else "step3d_gan_s3.pt"
fx = load_golden(fname)
for it, (rec, out, netG, netD, trainer) in enumerate(run_hip_stage(fx)):
    print("== iter", it)
    for k in ("total_loss", "rec_vae_loss", "kl_loss", "errD_real", "errD_fake", "gradient_penalty", "rec_loss", "errG", "generated", "fake"):
        if k in rec and k in out:
            print("%-18s rel %.3e" % (k, rel_err(out[k], rec[k])))
    if "gradsD" in rec:
        gotD = flat_to_named(out["gradD_flat"], trainer.arenaD, netD)
        worst = max(((rel_err(gotD[k], g), k) for k, g in rec["gradsD"].items() if g is not None))
        print("gradD worst", worst)
        sd = netD.state_dict()
        for k, v in rec["D_after"].items():
            d = (sd[k].float().cpu() - v.float()).abs().max().item()
            if d > 1e-6:
                n = ((sd[k].float().cpu() - v.float()).abs() > 1e-5).sum().item()
                print("D_after %-32s maxabs %.3e  count(>1e-5) %d / %d" % (k, d, n, v.numel()))
    gotG = flat_to_named(out["gradG_flat"], trainer.arenaG, netG)
    errs = sorted(((rel_err(gotG[k], g), k) for k, g in rec["gradsG"].items() if g is not None), reverse=True)
    print("gradG worst 6 (all)", errs[:6])
    errs = [e for e in errs if not e[1].endswith("conv.bias")]
    print("gradG worst 8 (no conv.bias)", errs[:8])
    print("total_norm rel", rel_err(out["clip_info"][1], rec["total_norm"]))

# oracle on this host vs golden
from smoke_step import _oracle_first_iter
want = _oracle_first_iter(fx)
rec = fx["iters"][0]
errs = sorted(((rel_err(want["gradsG"][k], g), k) for k, g in rec["gradsG"].items() if g is not None), reverse=True)
print("ORACLE-vs-golden gradG worst 6 (all)", errs[:6])
errs = [e for e in errs if not e[1].endswith("conv.bias")]
print("ORACLE-vs-golden gradG worst 8", errs[:8], "threads", torch.get_num_threads())
if "gradsD" in rec:
    errs = sorted(((rel_err(want["gradsD"][k], g), k) for k, g in rec["gradsD"].items() if g is not None), reverse=True)
    print("ORACLE-vs-golden gradD worst 3", errs[:3])
for k in ("errG", "errD_real", "gradient_penalty", "rec_loss"):
    if k in want:
        print(k, rel_err(want[k], rec[k]))
