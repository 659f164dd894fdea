import os, sys
sys.path.insert(0, "/root/repo/re2-modification_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.environ["MFA_VERBOSE"] = "1"
import numpy as np, torch, oracle_lib
from mfa_amd import capi, image
rng = np.random.default_rng(1)
strings = [bytes(rng.choice(list(b"ab"), size=600).tolist()) for _ in range(70000)]
data, off = oracle_lib.pack(strings)
d_b = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda"); d_b[:len(data)] = torch.from_numpy(data.copy())
d_o = torch.from_numpy(off.astype(np.int64)).cuda()
img = capi.Image(image.blob_from_dump(oracle_lib.load_dump("ex9_plain")))
os.environ["MFA_WALK"] = "table"
os.environ["MFA_MIXED_CUTS"] = "0.5"
mx = capi.Mixed([img])
for k in range(5):
    r = mx.match_tensors(d_b, d_o, [0, len(strings)]); torch.cuda.synchronize()
    print("call", k, int(r.sum()), flush=True)
