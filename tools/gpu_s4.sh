#!/bin/bash
# round 4, session 4: scratch pool + generic pass by look-back: parity (all encode tests), then encode times per content class
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-s4}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
SEL="encode or sweep or selectable or mixed or flat_frames or set_sizes or one_context or three_channel or 4k_frame or batch_1080p or many_small or recheck or 16k"
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SEL" > $OUT/pytest_enc.log 2>&1; echo "rc=$?" >> $OUT/pytest_enc.log; tail -5 $OUT/pytest_enc.log
for k in photo noise uiflat constant; do
  KIND=$k python tools/dev/enc_time.py - 256 2>&1 | grep -v amdgpu.ids | sed "s/^/[$k] 256 frames: /"
done | tee $OUT/enc_time.txt
python tools/dev/enc_time.py - 1024 2>&1 | grep -v amdgpu.ids | sed "s/^/[photo] 1024 frames: /" | tee -a $OUT/enc_time.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/enc_time.txt
import torch
from qoi_amd import api, synth
c = api.Context(0); w, h, F = 3840, 2160, 1024
ps = w*h*4; ss = (api.encode_bound(w, h, 4) + 255)//256*256
px = torch.empty(F*ps, dtype=torch.uint8, device='cuda'); st = torch.empty(F*ss, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
c.synth_frames(synth.KIND_ID['photo'], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s)
c.encode_batch(px.data_ptr(), ps, api.QoiDesc(w, h, 4, 0), F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
print('workspace bytes after a 1024-frame 4K encode:', c.workspace_bytes())
PY
