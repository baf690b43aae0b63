cd $GRAFT_REPO_ROOT
python -m pytest tests/test_metrics_gpu.py tests/test_varlen_gpu.py -q -x -m gpu -k "estoi or mixed or metrics" 2>&1 | tail -2
cat > /tmp/estoi_t.py <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from nele_gan_amd import metrics as mt, synth
c, v = synth.batch(16, 64000, start=40)
x = torch.from_numpy(np.tile(c, (16, 1))).cuda(); y = torch.from_numpy(np.tile(0.7 * c + v, (16, 1))).cuda()
for _ in range(3): mt.batch_estoi(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): r, _ = mt.batch_estoi(x, y)
torch.cuda.synchronize(); print('estoi B=256: %.3f ms' % ((time.perf_counter() - t0) / 5 * 1e3), r[:3].tolist())
PY
cp /tmp/estoi_t.py gpurun_out/estoi_t.py
for w in 1 0; do echo "TOBW=$w"; NELE_ESTOI_TOBW=$w bash tools/prof_one.sh gpurun_out/estoi_t.py $GRAFT_REPO_ROOT 2>&1 | grep -E "estoi"; done
