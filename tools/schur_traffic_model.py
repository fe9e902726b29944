#!/usr/bin/env python3
"""tools/schur_traffic_model.py -- how many bytes of E slots would a TILE-GROUPED Schur complement request?  

k_schur_pairs reads two 256-byte slots per term, pair by pair (L1723 shape: 6.07 M terms -> 3.11 GB requested, 1.94 GB of them
from beyond L2 by rocprofv3's FETCH_SIZE).  A workgroup per pair of camera TILES (14 cameras = 128 rows of S: the S tile fits the
LDS) that stages every slot once per tile pair needs, per landmark, (observations) x (distinct tiles of its cameras) slot loads.
This script counts that on the seeded bench shapes, with the cameras in generator order and in reverse-Cuthill-McKee order
(scipy's; the library's own RCM gives the same band).  No GPU, no library: numpy + scipy only."""
import sys, numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_amd import datasets as D
def model(name, gen, tile_cams):
    cams, pts, oc, op, oz = gen()
    nC = cams.shape[0]
    order = np.argsort(op, kind="stable"); oc_s = oc[order].astype(np.int64); op_s = op[order]
    cnt = np.bincount(op_s, minlength=pts.shape[0]); ptr = np.concatenate([[0], np.cumsum(cnt)])
    # camera graph + scipy RCM
    rows=[]; cols=[]
    for k in np.unique(cnt):
        if k < 2: continue
        ls = np.where(cnt == k)[0]
        obs = oc_s[ptr[ls][:, None] + np.arange(k)[None, :]]
        a, b = np.tril_indices(k, -1)
        rows.append(obs[:, a].reshape(-1)); cols.append(obs[:, b].reshape(-1))
    r = np.concatenate(rows); c = np.concatenate(cols)
    A = sp.coo_matrix((np.ones(r.size), (r, c)), shape=(nC, nC)).tocsr(); A = ((A + A.T) > 0).astype(np.int8).tocsr()
    perm = reverse_cuthill_mckee(A, symmetric_mode=True)
    pos = np.empty(nC, np.int64); pos[perm] = np.arange(nC)
    res = {}
    for label, p in (("camera index", np.arange(nC)), ("RCM position", pos)):
        tile = p[oc_s] // tile_cams
        loads = 0
        for k in np.unique(cnt):
            if k == 0: continue
            ls = np.where(cnt == k)[0]
            T = np.sort(tile[ptr[ls][:, None] + np.arange(k)[None, :]], axis=1)
            d = 1 + (np.diff(T, axis=1) != 0).sum(1)
            loads += int((k * d).sum())          # every slot is staged once per tile pair its tile takes part in: (distinct tiles) times
        res[label] = loads * 256
    terms = int((cnt.astype(np.int64) * (cnt + 1) // 2).sum())
    return terms, 2 * terms * 256, res
for name, gen in (("ladybug1723", D.ladybug_1723), ("venice1778", D.venice_1778)):
    for tc in (14, 28):
        terms, pm, res = model(name, gen, tc)
        print(f"{name} tile {tc} cams: pair-major {pm/1e9:.2f} GB requested; staged per tile pair: " + ", ".join(f"{k} {v/1e9:.2f} GB (x{pm/v:.1f})" for k, v in res.items()))
