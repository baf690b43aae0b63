import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nele_gan_amd import metrics as mt, synth
B = 256
c, v = synth.batch(B, 64000, start=1)
cw = torch.from_numpy(c).cuda(); yw = torch.from_numpy((c + v).astype(np.float32)).cuda()
for _ in range(3):
    mt.batch_haspi_quality(cw, yw, 16000, noise=True, seed=1)
torch.cuda.synchronize()
