import sys, os
sys.path.insert(0, os.getcwd())
import torch, hp_vae_gan_amd
from hp_vae_gan_amd import ops, lib as hplib
lib = hplib.load()
T,H,W = (int(v) for v in os.environ.get('HPVG_PERF_SHAPE','13,144,256').split(','))
B = int(os.environ.get('HPVG_PERF_B','2'))
x = torch.randn(B,64,T,H,W,device='cuda'); w = torch.randn(64,64,3,3,3,device='cuda')*0.05; b=torch.randn(64,device='cuda')
def bench(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for mode in (6,5):
    lib.hpvg_conv_wino_config(mode,-1)
    ya,bits = ops.conv_fwd_raw(x,w,b,out_lrelu=True,want_bits=True)
    t_plain = bench(lambda: ops.conv_fwd_raw(x,w,b))
    t_lrelu = bench(lambda: ops.conv_fwd_raw(x,w,b,out_lrelu=True))
    t_bits = bench(lambda: ops.conv_fwd_raw(x,w,b,out_lrelu=True,want_bits=True))
    t_mask = bench(lambda: ops.conv_fwd_raw(x,w,None,flip=True,mask_bits=bits))
    t_maskf = bench(lambda: ops.conv_fwd_raw(x,w,None,flip=True,out_mask=x))
    print((B,T,H,W),'mode',mode,'plain %.4f lrelu %.4f bits %.4f mask_bits %.4f mask_f32 %.4f'%(t_plain,t_lrelu,t_bits,t_mask,t_maskf))
