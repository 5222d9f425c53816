import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, numpy as np
import test_gpu_training_loop as T
for enc, mode, nn, nl in [("hash","nerf",64,4),("hash","compat",128,2),("freq","nerf",128,8),("freq","compat",128,8),("freq","nerf",64,2)]:
    batches = T._det_batches(torch)
    a = T._small_trainer(torch, enc, mode, nn, nl, deterministic=True)
    c = T._small_trainer(torch, enc, mode, nn, nl, deterministic=True)
    B = batches[0][0].shape[0]
    c.entry_args(B)
    worst = []
    for i,(o,d,t) in enumerate(batches):
        a.step(o,d,t)
        c.graph_rays_o.copy_(o); c.graph_rays_d.copy_(d); c.graph_targets.copy_(t)
        c.step_entry()
        torch.cuda.synchronize()
        r = [float((x.float()-y.float()).norm())/ (float(x.float().norm())+1e-30) for x,y in zip(T._state(a), T._state(c)) if x.dtype in (torch.float32, torch.float16)]
        worst.append(max(r))
    print(enc, mode, nn, nl, ["%.2e" % w for w in worst], flush=True)
