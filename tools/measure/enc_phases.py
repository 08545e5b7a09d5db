"""Where a wavefront of enc_sets spends its life: python tools/measure/enc_phases.py <lib built with -DQOIMI_ENC_PHASES> [frames]
(build: see the comment at g_enc_phase in qoi_amd/csrc/qoi_encode.hip; the library is a diagnostic build, never the product)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qoi_amd import api, synth
api.LIB_PATH = os.path.abspath(sys.argv[1])
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib = api.load_library()
w, h = 3840, 2160
c = api.Context(0)
npx = w * h; ps = npx * 4; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); st = torch.empty(F * ss, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
c.synth_frames(synth.KIND_ID[os.environ.get('KIND', 'photo')], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s)
desc = api.QoiDesc(w, h, 4, 0)
buf = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
lib.qoimi_debug_enc_phases(buf, 1)
for _ in range(3):
    c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
lib.qoimi_debug_enc_phases(buf, 0)
v = list(buf); sets = v[5]; tot = sum(v[0:5])
names = ['entry state', 'groups inside the image', 'general-form groups', 'look-back', 'copy-out']
print(f"R={os.environ.get('QOIMI_ENC_SET_SLABS', '-')} sets {sets}: ticks per set {tot / sets:.0f}; " + ', '.join(f"{n} {v[i] / sets:.0f} ({100.0 * v[i] / tot:.1f} %)" for i, n in enumerate(names)) + f"; sets that re-polled {100.0 * v[6] / max(sets, 1):.1f} %, re-polls per set {v[7] / max(sets, 1):.2f}")
