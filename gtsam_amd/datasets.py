"""Synthetic stand-ins for the datasets the reference's configs name but this image does not ship
(no network): BAL Ladybug-1723 / Venice-1778 / Dubrovnik-16 shapes and small seeded graphs for tests.

The BAL generator follows SURVEY.md section 8(d): cameras on a closed multi-loop street path so that
distant cameras share points (non-banded reduced camera system), long-tailed track lengths
(2 + Exp(mean) capped, >= 1 % of points seen by > 50 cameras to emulate loop closures),
f ~ U(400, 900), small radial distortion, 0.5 px pixel noise, perturbed initial values.
Everything is seeded and generated with numpy on the host; nothing here is on the hot path.
"""
from __future__ import annotations

import numpy as np

from .problem import (NOISE_DIAGONAL, NOISE_GAUSSIAN, NOISE_ISOTROPIC, NOISE_UNIT, Problem, bal_problem,
                      pose_graph_problem)


def _rodrigues(w):
    """exp map for host-side data generation only (not the device formula)."""
    w = np.asarray(w, np.float64).reshape(-1, 3)
    th = np.linalg.norm(w, axis=1)
    k = w / np.where(th > 1e-12, th, 1.0)[:, None]
    K = np.zeros((w.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -k[:, 2], k[:, 1], k[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 0], -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, None, None], np.cos(th)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def synthetic_bal(n_cams, n_points, mean_track=4.34, seed=42, long_frac=0.012, pixel_noise=0.5,
                  n_loops=3):
    """Returns (cams17 initial, pts3 initial, obs_cam, obs_pt, obs_z) with observations sorted by point
    then camera, the order SfmData::FromBalFile produces (sfm/SfmData.cpp:189-246)."""
    rng = np.random.default_rng(seed)
    # camera path: n_loops laps around a rounded block, slightly different radii -> revisits
    s = np.linspace(0, n_loops * 2 * np.pi, n_cams, endpoint=False)
    radius = 60.0 + 4.0 * np.sin(3 * s) + 1.5 * (s / (2 * np.pi))
    centers = np.stack([radius * np.cos(s), radius * np.sin(s), 1.5 + 0.2 * np.sin(5 * s)], 1)
    heading = s + np.pi / 2 + rng.normal(0, 0.05, n_cams)        # drive direction
    # Ladybug is omnidirectional: each "camera" looks sideways-ish with a random yaw offset
    yaw = heading + rng.uniform(-np.pi, np.pi, n_cams)
    # camera frame: z forward (optical axis), x right, y down
    fwd = np.stack([np.cos(yaw), np.sin(yaw), np.zeros(n_cams)], 1)
    down = np.tile(np.array([0, 0, -1.0]), (n_cams, 1))
    right = np.cross(down, fwd)
    Rwc = np.stack([right, down, fwd], 2)                         # columns = camera axes in world
    tilt = _rodrigues(rng.normal(0, 0.03, (n_cams, 3)))
    Rwc = Rwc @ tilt
    f = rng.uniform(400, 900, n_cams)
    k1 = rng.normal(0, 1e-2, n_cams)
    k2 = rng.normal(0, 1e-4, n_cams)

    # points: scattered in an annulus around the path (building facades both sides)
    ang = rng.uniform(0, 2 * np.pi, n_points)
    rad = 60.0 + rng.choice([-1, 1], n_points) * rng.uniform(6, 25, n_points)
    pts = np.stack([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(0, 12, n_points)], 1)

    # track lengths: 2 + Exp, long tail
    k = 2 + np.floor(rng.exponential(mean_track - 1.72 - 50 * long_frac, n_points)).astype(np.int64)
    long_idx = rng.choice(n_points, max(1, int(long_frac * n_points)), replace=False)
    k[long_idx] = rng.integers(50, 120, long_idx.size)
    k = np.minimum(k, n_cams)

    # candidate cameras for a point: those whose centre is close in angle (any lap) and that see it
    cam_ang = np.mod(s, 2 * np.pi)
    order = np.argsort(cam_ang)
    sorted_ang = cam_ang[order]
    obs_cam, obs_pt = [], []
    for j in range(n_points):
        width = 0.07 + 0.004 * k[j] + 3.0 * (2 * np.pi * n_loops / n_cams)
        lo = np.searchsorted(sorted_ang, ang[j] - width)
        hi = np.searchsorted(sorted_ang, ang[j] + width)
        cand = order[lo:hi]
        if ang[j] - width < 0:
            cand = np.concatenate([cand, order[np.searchsorted(sorted_ang, ang[j] - width + 2 * np.pi):]])
        if ang[j] + width > 2 * np.pi:
            cand = np.concatenate([cand, order[:np.searchsorted(sorted_ang, ang[j] + width - 2 * np.pi)]])
        if cand.size == 0:
            cand = order[[lo % n_cams]]
        # keep those with the point in front and within a generous field of view
        d = pts[j] - centers[cand]
        zc = np.einsum("ni,ni->n", d, Rwc[cand][:, :, 2])
        xc = np.einsum("ni,ni->n", d, Rwc[cand][:, :, 0])
        ok = (zc > 1.0) & (np.abs(xc) < 1.2 * zc)
        good = cand[ok]
        if good.size < 2:
            # force visibility: re-aim nothing, just take nearest cameras in front
            good = cand[zc > 0.5]
        if good.size < 2:
            continue
        take = rng.choice(good, min(k[j], good.size), replace=False)
        take.sort()
        obs_cam.append(take); obs_pt.append(np.full(take.size, j))
    obs_cam = np.concatenate(obs_cam); obs_pt_raw = np.concatenate(obs_pt)
    # drop unobserved points, renumber
    used = np.unique(obs_pt_raw)
    remap = -np.ones(n_points, np.int64); remap[used] = np.arange(used.size)
    obs_pt = remap[obs_pt_raw]; pts = pts[used]
    # drop cameras that ended up without observations (tiny configs), renumber
    usedc = np.unique(obs_cam)
    if usedc.size != n_cams:
        remapc = -np.ones(n_cams, np.int64); remapc[usedc] = np.arange(usedc.size)
        obs_cam = remapc[obs_cam]
        centers, Rwc, f, k1, k2 = centers[usedc], Rwc[usedc], f[usedc], k1[usedc], k2[usedc]
        n_cams = usedc.size

    # ground-truth projection (host-side generation only)
    Rcw = np.swapaxes(Rwc, 1, 2)
    q = np.einsum("nij,nj->ni", Rcw[obs_cam], pts[obs_pt] - centers[obs_cam])
    x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = x * x + y * y
    g = 1 + (k1[obs_cam] + k2[obs_cam] * r2) * r2
    z = np.stack([f[obs_cam] * g * x, f[obs_cam] * g * y], 1) + rng.normal(0, pixel_noise, (obs_cam.size, 2))

    # initial values = truth perturbed
    dR = _rodrigues(rng.normal(0, 2e-3, (n_cams, 3)))
    cams = np.zeros((n_cams, 17))
    cams[:, :9] = (Rwc @ dR).reshape(-1, 9)
    cams[:, 9:12] = centers + rng.normal(0, 2e-2, (n_cams, 3))
    cams[:, 12] = f + rng.normal(0, 1.0, n_cams); cams[:, 13] = k1; cams[:, 14] = k2
    pts0 = pts + rng.normal(0, 5e-2, pts.shape)
    return cams, pts0, obs_cam.astype(np.int32), obs_pt.astype(np.int32), z


def synthetic_bal_streets(n_cams, n_points, mean_track=4.34, seed=42, grid=5, block=45.0, pixel_noise=0.5, long_frac=0.012):
    """A second BAL shape for the sensitivity of the ordering / tile design to the co-visibility structure (hard
    part 8): the cameras drive a RANDOM WALK through a grid x grid street network (no U-turns), so streets are revisited at random
    times and in both directions -- long-range loop closures all over the reduced camera system instead of the cyclic band of
    synthetic_bal's laps around one block.  Landmarks sit on the facades (6-25 m off a street's centre line); a landmark is seen
    by cameras within 35 m that have it in front and in their field of view, 2 + Exp(mean) of them (long tail as in
    synthetic_bal).  Same camera model, noise, perturbation and return convention as synthetic_bal."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    # ---- the drive: node to node on the grid, one camera every `step` metres
    n_edges_needed = 4 * grid * grid
    step = n_edges_needed * block / n_cams
    node = np.array([rng.integers(0, grid + 1), rng.integers(0, grid + 1)])
    prev_dir = None
    centers, headings = [], []
    carry = 0.0
    dirs = np.array([[1, 0], [-1, 0], [0, 1], [0, -1]])
    while len(centers) < n_cams:
        ok = [d for d in dirs if 0 <= node[0] + d[0] <= grid and 0 <= node[1] + d[1] <= grid and (prev_dir is None or not np.array_equal(d, -prev_dir))]
        d = ok[rng.integers(0, len(ok))]
        a, b = node * block, (node + d) * block
        t = carry
        while t < block and len(centers) < n_cams:
            c = a + (b - a) * (t / block)
            centers.append([c[0] + rng.normal(0, 0.3), c[1] + rng.normal(0, 0.3), 1.5 + 0.2 * rng.normal()])
            headings.append(np.arctan2(d[1], d[0]))
            t += step
        carry = t - block
        node = node + d; prev_dir = d
    centers = np.array(centers); heading = np.array(headings) + rng.normal(0, 0.05, n_cams)
    yaw = heading + rng.uniform(-np.pi, np.pi, n_cams)                   # omnidirectional rig: a random viewing direction per camera
    fwd = np.stack([np.cos(yaw), np.sin(yaw), np.zeros(n_cams)], 1)
    down = np.tile(np.array([0, 0, -1.0]), (n_cams, 1))
    right = np.cross(down, fwd)
    Rwc = np.stack([right, down, fwd], 2) @ _rodrigues(rng.normal(0, 0.03, (n_cams, 3)))
    f = rng.uniform(400, 900, n_cams); k1 = rng.normal(0, 1e-2, n_cams); k2 = rng.normal(0, 1e-4, n_cams)
    # ---- landmarks on the facades of the streets that were driven
    anchor = centers[rng.integers(0, n_cams, n_points)]
    side = rng.uniform(0, 2 * np.pi, n_points)
    off = rng.uniform(6, 25, n_points)
    pts = np.stack([anchor[:, 0] + off * np.cos(side), anchor[:, 1] + off * np.sin(side), rng.uniform(0, 12, n_points)], 1)
    k = 2 + np.floor(rng.exponential(mean_track - 1.72 - 50 * long_frac, n_points)).astype(np.int64)
    long_idx = rng.choice(n_points, max(1, int(long_frac * n_points)), replace=False)
    k[long_idx] = rng.integers(50, 120, long_idx.size)
    tree = cKDTree(centers[:, :2])
    cand_all = tree.query_ball_point(pts[:, :2], r=35.0)
    obs_cam, obs_pt = [], []
    for j in range(n_points):
        cand = np.asarray(cand_all[j], np.int64)
        if cand.size < 2:
            continue
        d = pts[j] - centers[cand]
        zc = np.einsum("ni,ni->n", d, Rwc[cand][:, :, 2]); xc = np.einsum("ni,ni->n", d, Rwc[cand][:, :, 0])
        good = cand[(zc > 1.0) & (np.abs(xc) < 1.2 * zc)]
        if good.size < 2:
            continue
        take = rng.choice(good, min(k[j], good.size), replace=False); take.sort()
        obs_cam.append(take); obs_pt.append(np.full(take.size, j))
    obs_cam = np.concatenate(obs_cam); obs_pt_raw = np.concatenate(obs_pt)
    used = np.unique(obs_pt_raw)
    remap = -np.ones(n_points, np.int64); remap[used] = np.arange(used.size)
    obs_pt = remap[obs_pt_raw]; pts = pts[used]
    usedc = np.unique(obs_cam)
    if usedc.size != n_cams:
        remapc = -np.ones(n_cams, np.int64); remapc[usedc] = np.arange(usedc.size)
        obs_cam = remapc[obs_cam]
        centers, Rwc, f, k1, k2 = centers[usedc], Rwc[usedc], f[usedc], k1[usedc], k2[usedc]
        n_cams = usedc.size
    Rcw = np.swapaxes(Rwc, 1, 2)
    q = np.einsum("nij,nj->ni", Rcw[obs_cam], pts[obs_pt] - centers[obs_cam])
    x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = x * x + y * y
    g = 1 + (k1[obs_cam] + k2[obs_cam] * r2) * r2
    z = np.stack([f[obs_cam] * g * x, f[obs_cam] * g * y], 1) + rng.normal(0, pixel_noise, (obs_cam.size, 2))
    dR = _rodrigues(rng.normal(0, 2e-3, (n_cams, 3)))
    cams = np.zeros((n_cams, 17))
    cams[:, :9] = (Rwc @ dR).reshape(-1, 9)
    cams[:, 9:12] = centers + rng.normal(0, 2e-2, (n_cams, 3))
    cams[:, 12] = f + rng.normal(0, 1.0, n_cams); cams[:, 13] = k1; cams[:, 14] = k2
    pts0 = pts + rng.normal(0, 5e-2, pts.shape)
    return cams, pts0, obs_cam.astype(np.int32), obs_pt.astype(np.int32), z


def streets_1723(seed=42):
    """The L1723 size (1 723 cameras, ~156 000 points, ~4.3 observations per point) on the street-network drive."""
    return synthetic_bal_streets(1723, 156502, mean_track=4.34, seed=seed)


def ladybug_1723(seed=42):
    """BAL Ladybug problem-1723-156502 shape (1723 cameras, 156 502 points, ~678 718 obs)."""
    return synthetic_bal(1723, 156502, mean_track=4.34, seed=seed)


def venice_1778(seed=42):
    return synthetic_bal(1778, 993923, mean_track=5.03, seed=seed)


def synthetic_bal_convergent(n_cams, n_points, mean_track, seed=42, pixel_noise=0.5):
    """A few cameras on a ring looking at one scene (the geometry of the small Dubrovnik problems): every camera sees
    every point, a point is observed by 2 + Exp cameras.  Same return convention as synthetic_bal."""
    rng = np.random.default_rng(seed)
    a = np.linspace(0, 2 * np.pi, n_cams, endpoint=False) + rng.normal(0, 0.05, n_cams)
    centers = np.stack([30 * np.cos(a), 30 * np.sin(a), rng.normal(0, 2.0, n_cams)], 1)
    target = rng.normal(0, 1.0, (n_cams, 3))
    fwd = target - centers; fwd /= np.linalg.norm(fwd, axis=1)[:, None]
    down = np.tile(np.array([0, 0, -1.0]), (n_cams, 1))
    right = np.cross(down, fwd); right /= np.linalg.norm(right, axis=1)[:, None]
    down = np.cross(fwd, right)
    Rwc = np.stack([right, down, fwd], 2)
    f = rng.uniform(400, 900, n_cams); k1 = rng.normal(0, 1e-2, n_cams); k2 = rng.normal(0, 1e-4, n_cams)
    pts = rng.normal(0, 4.0, (n_points, 3))
    k = np.minimum(2 + np.floor(rng.exponential(mean_track - 2.0, n_points)).astype(np.int64), n_cams)
    obs_cam = np.concatenate([np.sort(rng.choice(n_cams, kk, replace=False)) for kk in k])
    obs_pt = np.repeat(np.arange(n_points), k)
    Rcw = np.swapaxes(Rwc, 1, 2)
    q = np.einsum("nij,nj->ni", Rcw[obs_cam], pts[obs_pt] - centers[obs_cam])
    x, y = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = x * x + y * y
    g = 1 + (k1[obs_cam] + k2[obs_cam] * r2) * r2
    z = np.stack([f[obs_cam] * g * x, f[obs_cam] * g * y], 1) + rng.normal(0, pixel_noise, (obs_cam.size, 2))
    dR = _rodrigues(rng.normal(0, 2e-3, (n_cams, 3)))
    cams = np.zeros((n_cams, 17))
    cams[:, :9] = (Rwc @ dR).reshape(-1, 9)
    cams[:, 9:12] = centers + rng.normal(0, 2e-2, (n_cams, 3))
    cams[:, 12] = f + rng.normal(0, 1.0, n_cams); cams[:, 13] = k1; cams[:, 14] = k2
    return cams, pts + rng.normal(0, 5e-2, pts.shape), obs_cam.astype(np.int32), obs_pt.astype(np.int32), z


def dubrovnik_16(seed=42):
    """BAL Dubrovnik problem-16-22106 shape (16 cameras, 22 106 points, ~83 718 observations)."""
    return synthetic_bal_convergent(16, 22106, mean_track=4.29, seed=seed)


def random_pose_graph(n, n_closures, seed=0, noise="mixed", rot_scale=1.0, init_noise=0.2):
    """Small seeded Pose3 graph: chain + loop closures, prior on pose 0.  Returns (Problem, values0)."""
    rng = np.random.default_rng(seed)
    xi = rng.normal(size=(n, 6)); xi[:, :3] *= rot_scale
    R = _rodrigues(xi[:, :3]); t = xi[:, 3:] * 2.0
    poses = np.concatenate([R.reshape(n, 9), t], 1)
    edges = [(i, i + 1) for i in range(n - 1)]
    while len(edges) < n - 1 + n_closures:
        a, b = rng.integers(0, n, 2)
        if a != b:
            edges.append((int(a), int(b)))
    v1 = np.array([e[0] for e in edges]); v2 = np.array([e[1] for e in edges])
    Ra, ta, Rb, tb = R[v1], t[v1], R[v2], t[v2]
    hR = np.swapaxes(Ra, 1, 2) @ Rb
    ht = np.einsum("nji,nj->ni", Ra, tb - ta)
    nR = _rodrigues(rng.normal(0, 0.03, (len(edges), 3))); nt = rng.normal(0, 0.05, (len(edges), 3))
    zR = hR @ nR; zt = ht + np.einsum("nij,nj->ni", hR, nt)
    z = np.concatenate([zR.reshape(-1, 9), zt], 1)
    nk = np.zeros(len(edges), np.int32); nd = np.zeros((len(edges), 36))
    for k in range(len(edges)):
        mode = {"mixed": k % 3, "diagonal": 0, "gaussian": 1, "isotropic": 2}[noise]
        if mode == 0:
            nk[k] = NOISE_DIAGONAL; nd[k, :6] = [0.1, 0.1, 0.1, 0.3, 0.3, 0.2]
        elif mode == 1:
            A = rng.normal(size=(6, 6)); info = A @ A.T + 6 * np.eye(6)
            nk[k] = NOISE_GAUSSIAN; nd[k] = np.linalg.cholesky(info).T.reshape(-1)
        else:
            nk[k] = NOISE_ISOTROPIC; nd[k, 0] = 0.2
    p = pose_graph_problem(n, v1, v2, z, nk, nd)
    npri = p.add_noise(NOISE_DIAGONAL, 6, np.sqrt([1e-6] * 3 + [1e-4] * 3))  # Pose3SLAMExample_g2o.cpp:41-43
    p.add_prior(0, poses[0], npri)
    dR = _rodrigues(rng.normal(0, init_noise, (n, 3)))
    v0 = np.concatenate([(R @ dR).reshape(n, 9), t + rng.normal(0, init_noise, (n, 3))], 1)
    v0[0] = poses[0]
    return p, v0.reshape(-1)


def random_projection_graph(n_poses=6, n_points=40, seed=0, with_sensor=True, behind=True, distortion=None):
    """Small GenericProjectionFactor<Pose3,Point3,Cal3_S2> graph (+ priors).  Returns (Problem, values0).
    distortion: optional (2, 4) array k1, k2, p1, p2 for the two calibrations -- a non-zero row makes that calibration a Cal3DS2
    (GenericProjectionFactor<Pose3,Point3,Cal3DS2>); the measurements are generated through the same distortion."""
    from .problem import VAR_POINT3, VAR_POSE3
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 1.0, n_poses)
    centers = np.stack([4 * np.sin(ang), 0.3 * rng.normal(size=n_poses), -6 + 0.5 * np.cos(ang)], 1)
    R = _rodrigues(np.stack([0.05 * rng.normal(size=n_poses), -0.4 * ang + 0.2, 0.03 * rng.normal(size=n_poses)], 1))
    pts = np.stack([rng.uniform(-3, 3, n_points), rng.uniform(-2, 2, n_points), rng.uniform(2, 8, n_points)], 1)
    if behind:
        pts[-2:, 2] = -9.0  # behind every camera: cheirality branch
    K = np.array([[520.0, 515.0, 0.3, 320.0, 240.0], [400.0, 400.0, 0.0, 300.0, 200.0]])
    sensor = np.concatenate([_rodrigues(np.array([[0.02, -0.03, 0.01]])).reshape(-1), [0.1, -0.05, 0.2]])
    vt = np.concatenate([np.full(n_poses, VAR_POSE3, np.int32), np.full(n_points, VAR_POINT3, np.int32)])
    p = Problem(var_type=vt)
    n_iso = p.add_noise(NOISE_ISOTROPIC, 2, [1.5]); n_unit = p.add_noise(NOISE_UNIT, 2)
    pose_i, pt_i, zs, nz, ci, si = [], [], [], [], [], []
    for i in range(n_poses):
        for j in range(n_points):
            if rng.uniform() < 0.5:
                continue
            use_s = with_sensor and (i % 2 == 1)
            Rw, tw = R[i], centers[i]
            if use_s:
                Rs = sensor[:9].reshape(3, 3); ts = sensor[9:]
                Rw, tw = Rw @ Rs, tw + R[i] @ ts
            q = Rw.T @ (pts[j] - tw)
            k = K[i % 2]
            if q[2] > 0:
                u, v = q[0] / q[2], q[1] / q[2]
                if distortion is not None:
                    k1, k2, p1, p2 = np.asarray(distortion, float)[i % 2]
                    rr = u * u + v * v; gg = 1 + k1 * rr + k2 * rr * rr
                    u, v = gg * u + 2 * p1 * u * v + p2 * (rr + 2 * u * u), gg * v + 2 * p2 * u * v + p1 * (rr + 2 * v * v)
                z = np.array([k[0] * u + k[2] * v + k[3], k[1] * v + k[4]]) + rng.normal(0, 1.0, 2)
            else:
                z = rng.normal(0, 50, 2)
            pose_i.append(i); pt_i.append(n_poses + j); zs.append(z); nz.append(n_iso if j % 2 else n_unit)
            ci.append(i % 2); si.append(0 if use_s else -1)
    p.proj_pose = np.array(pose_i, np.int32); p.proj_point = np.array(pt_i, np.int32)
    p.proj_z = np.array(zs).reshape(-1); p.proj_noise = np.array(nz, np.int32)
    p.proj_calib = np.array(ci, np.int32); p.proj_sensor = np.array(si, np.int32)
    p.calib = K.reshape(-1); p.sensor = sensor
    if distortion is not None:
        p.calib_distortion = np.ascontiguousarray(np.asarray(distortion, np.float64).reshape(-1))
    poses = np.concatenate([R.reshape(n_poses, 9), centers], 1)
    n6 = p.add_noise(NOISE_DIAGONAL, 6, [0.01] * 3 + [0.05] * 3); n3 = p.add_noise(NOISE_ISOTROPIC, 3, [0.1])
    p.add_prior(0, poses[0], n6); p.add_prior(1, poses[1], n6); p.add_prior(n_poses, pts[0], n3)
    dR = _rodrigues(rng.normal(0, 0.01, (n_poses, 3)))
    v0 = np.concatenate([np.concatenate([(R @ dR).reshape(n_poses, 9), centers + rng.normal(0, 0.05, centers.shape)], 1).reshape(-1),
                         (pts + rng.normal(0, 0.05, pts.shape)).reshape(-1)])
    return p, v0


def synthetic_orbit_scene(n_cams=10, n_points=120, seed=0, see=0.7, pixel_noise=0.5, init_noise=(0.01, 0.05), arc=0.9, far_points=0,
                          far_distance=(1500.0, 4000.0), spread=1.0):
    """A small structure-from-motion scene with a sane field of view (cameras on an arc around a point cloud, every measurement
    within ~0.5 of the optical axis in intrinsic coordinates, so that Cal3Bundler::calibrate converges): the input of the smart
    factor tests.  `arc`: half the opening of the arc in radians (below ~0.55 every camera sees every other camera's viewing
    directions in front of it: what a smart factor's point at infinity needs).  `far_points`: that many of the points lie
    `far_distance` away behind the cloud (parallax of a few pixels: a landmark-distance threshold rejects them and the point at
    infinity is a good model of them).  `spread` widens the cloud across the optical axes (the distortion coefficients are
    barely observable from measurements near the image centre).  Returns the BAL-style tuple (cams17 -- perturbed --, pts3, obs_cam, obs_pt, obs_z) with the packing of
    bal_problem (pose R row-major + t, f, k1, k2, u0, v0); every camera sees at least two points, every point is seen twice."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(-arc, arc, n_cams)
    centers = np.stack([8 * np.sin(ang), 0.3 * rng.normal(size=n_cams), -8 * np.cos(ang)], 1)
    R = _rodrigues(np.stack([0.02 * rng.normal(size=n_cams), -ang, 0.02 * rng.normal(size=n_cams)], 1))   # looks at the origin (+z)
    pts = np.stack([rng.uniform(-2, 2, n_points) * spread, rng.uniform(-1.5, 1.5, n_points) * spread, rng.uniform(-2, 2, n_points)], 1)
    if far_points:                                     # (own generator: the stream of the near scene does not depend on this option)
        r2 = np.random.default_rng(seed + 1000)
        dist = r2.uniform(far_distance[0], far_distance[1], far_points)
        pts[-far_points:] = np.stack([dist * r2.uniform(-0.15, 0.15, far_points), dist * r2.uniform(-0.12, 0.12, far_points), dist], 1)
    f = rng.uniform(450, 800, n_cams); k1 = rng.normal(0, 2e-2, n_cams); k2 = rng.normal(0, 2e-3, n_cams)
    oc, op, oz = [], [], []
    seen = rng.uniform(size=(n_cams, n_points)) < see
    seen[:2, :] = True
    for j in range(n_points):
        for i in range(n_cams):
            if not seen[i, j]:
                continue
            q = R[i].T @ (pts[j] - centers[i])
            u, v = q[0] / q[2], q[1] / q[2]
            rr = u * u + v * v
            g = 1 + (k1[i] + k2[i] * rr) * rr
            oc.append(i); op.append(j); oz.append([f[i] * g * u + pixel_noise * rng.normal(), f[i] * g * v + pixel_noise * rng.normal()])
    dR = _rodrigues(rng.normal(0, init_noise[0], (n_cams, 3)))
    cams = np.zeros((n_cams, 17))
    cams[:, :9] = (R @ dR).reshape(n_cams, 9); cams[:, 9:12] = centers + rng.normal(0, init_noise[1], centers.shape)
    cams[:, 12] = f + rng.normal(0, 1.0, n_cams); cams[:, 13] = k1; cams[:, 14] = k2
    return cams, pts + rng.normal(0, 0.05, pts.shape), np.array(oc, np.int32), np.array(op, np.int32), np.array(oz)
