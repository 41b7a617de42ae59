#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP Canny path with the numpy restatement (GPU box): where do they part?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import canny_oracle as co
from goal_force_amd.canny import CannyDetector, ControlSignalDataset_CannyEdge as DS
from test_canny import _frames
shape = (2, 480, 832)
fr = _frames(sum(shape), *shape)
det = CannyDetector("cuda")
img = det.resize(fr).cpu().numpy()
ref = np.stack([co.resize_lanczos4_u8(f, 512, 896) for f in fr])
d = img.astype(int) - ref.astype(int)
print("lanczos: differing values", int((d != 0).sum()), "max abs", int(np.abs(d).max()), "where", np.argwhere(d != 0)[:6].tolist())
st = det.canny(torch.from_numpy(ref)).cpu().numpy()
e_ref = np.stack([co.canny_u8(f) for f in ref])
de = ((st == 2) * 255).astype(int) - e_ref.astype(int)
print("canny on the oracle's resized image: differing pixels", int((de != 0).sum()), "where", np.argwhere(de != 0)[:10].tolist())
if (de != 0).any():
    t, y, x = np.argwhere(de != 0)[0]
    print("  neighbourhood hip state:\n", st[t, max(0, y - 2):y + 3, max(0, x - 2):x + 3], "\n  oracle edges:\n", e_ref[t, max(0, y - 2):y + 3, max(0, x - 2):x + 3])
ds = DS(device="cuda")
# area resize alone: feed the oracle's edge state
import goal_force_amd.canny as cn
from goal_force_amd import _lib
state = torch.from_numpy(np.where(e_ref == 255, 2, 1).astype(np.uint8)).cuda()
tabs = [cn._dev(a, state.device) for a in (*cn._area_or_identity(896, 832), *cn._area_or_identity(512, 480))]
out = torch.empty((2, 480, 832, 3), dtype=torch.bfloat16, device="cuda")
_lib.check(_lib.load().gf_resize_area_u8(state.data_ptr(), out.data_ptr(), *(t.data_ptr() for t in tabs), 2, 512, 896, 480, 832, 1,
                                         torch.cuda.current_stream().cuda_stream), "area")
want = torch.stack([torch.from_numpy(co.resize_area_u8(np.stack([e, e, e], 2), 480, 832)) for e in e_ref]).float() / 127.5 - 1.0
dd = (out.cpu().float() - want.to(torch.bfloat16).float())
print("area resize on the oracle's edges: differing values", int((dd != 0).sum()), "where", (dd != 0).nonzero()[:6].tolist())
if (dd != 0).any():
    i = (dd != 0).nonzero()[0].tolist()
    print("  hip", float(out.cpu()[tuple(i)]), "oracle", float(want.to(torch.bfloat16)[tuple(i)]), "oracle f32", float(want[tuple(i)]))
