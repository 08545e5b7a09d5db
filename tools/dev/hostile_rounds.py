import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from qoi_amd import api
import test_gpu_parity as t
c = api.Context(0)
w, h = 1024, 600
s = t._hostile_stream(500000, w, h)
buf = torch.from_numpy(np.frombuffer(s + b"\0" * 8, dtype=np.uint8).copy()).cuda()
out = torch.full((w * h * 4 + 8,), 0xCD, dtype=torch.uint8, device="cuda")
import time
t0=time.time()
c.decode_batch(buf.data_ptr(), buf.numel(), [len(s)], [api.QoiDesc(w, h, 4, 0)], 4, out.data_ptr(), w * h * 4)
print("hostile", c.decode_stats(), round(time.time()-t0,3))
