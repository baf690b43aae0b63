#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_epoch_gpu.py tests/test_train_gpu.py tests/test_dataio_gpu.py tests/test_varlen_gpu.py -x -q 2>&1 | tail -3
for k in 1 2; do
timeout 900 python tools/files_time.py 0 256 2>&1 | grep epoch_from_files | python -c "
import sys, json
d = json.loads(sys.stdin.read().split(' ', 1)[1]); print('from files %.0f utt/s (%.1f ms), resident %.0f (%.1f ms), d_steps %s' % (d['value'], d['ms_per_epoch'], d['resident_batches']['value'], d['resident_batches']['ms_per_epoch'], d['d_steps']))"
done
