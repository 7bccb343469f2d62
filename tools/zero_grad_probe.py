import sys, os, argparse
sys.path.insert(0, os.getcwd())
import torch, bench
"""Fraction of ray-samples whose loss gradient (draw) is exactly zero at cfg2, and how they cluster along rays (run on the GPU box)."""
args = argparse.Namespace(keyframes=16, height=480, width=640, rays=4096, log2_T=19, mlp='baseline', precision='fp16x3', finest=256)
torch.cuda.set_device(0)
runner, cfg = bench.build_runner(args, 0, 1, torch.device('cuda', 0))
for n in (20, 300, 1500):
    while runner.field.global_step < n:
        runner.train_loop(); runner.global_step += 1
    torch.cuda.synchronize()
    b = runner.field._buffers(4096, 192)
    df = b['dfeat']
    z = (df == 0).all(0).all(-1)
    dr = b['draw']
    print(n, 'samples with all-zero dfeat:', float(z.float().mean()), '| zero draw rows:', float((dr == 0).all(-1).float().mean()), '| valid', float(b['valid'].float().mean()))
    print('   all-zero tiles of 64 / 32 / 16 samples:', float(z.view(-1, 64).all(1).float().mean()), float(z.view(-1, 32).all(1).float().mean()), float(z.view(-1, 16).all(1).float().mean()),
          '| tiles of 64 with any zero:', float(z.view(-1, 64).any(1).float().mean()))
    zr = z.view(4096, 192).float()
    print('   zero fraction by sample position (8 bins of 24):', [round(float(zr[:, i * 24:(i + 1) * 24].mean()), 2) for i in range(8)])
