"""Dataset readers (gtsam_amd/io.py) against the reference's own loaders: the golden fixtures hold what
SfmData::FromBalFile / readG2o returned for the reference's shipped files; when /root/reference and oracle/_ref are
present (build container) the live loaders are compared as well."""
import os

import numpy as np
import pytest

from gtsam_amd import io
from gtsam_amd.problem import NOISE_GAUSSIAN, NOISE_ISOTROPIC, NOISE_UNIT
from tests.conftest import load_golden

DATA = "/root/reference/examples/Data/"


def test_bal_reader_matches_reference_loader():
    path = DATA + "dubrovnik-3-7-pre.txt"
    if not os.path.exists(path):
        pytest.skip("reference data not present on this machine")
    g = load_golden("dubrovnik_3_7")
    cams, pts, oc, op, oz = io.read_bal(path)
    assert np.array_equal(oc, g["obs_cam"]) and np.array_equal(op, g["obs_pt"])
    assert np.array_equal(oz, g["obs_z"]) and np.array_equal(pts, g["pts"])          # float32 temporaries: exact
    assert np.abs(cams - g["cams"]).max() <= 1e-14 * np.abs(g["cams"]).max()


def test_toro_reader_matches_reference_loader():
    path = DATA + "sphere2500.txt"
    if not os.path.exists(path):
        pytest.skip("reference data not present on this machine")
    g = load_golden("sphere2500")
    d = io.read_g2o3d(path)
    assert np.array_equal(d["v1"], g["v1"]) and np.array_equal(d["v2"], g["v2"])
    assert np.array_equal(d["noise_kind"], g["noise_kind"])
    assert np.abs(d["z"] - g["z"]).max() <= 1e-15
    assert np.abs(d["noise"] - g["noise"]).max() <= 1e-15
    assert d["vertex_keys"].size == 0        # load3D does not create missing vertices


@pytest.mark.parametrize("name", ["pose3example.txt", "pose3example-offdiagonal.txt", "pose3example-grid.txt"])
def test_g2o_quat_reader_matches_live_reference(name, live_ref):
    path = DATA + name
    if live_ref is None or not os.path.exists(path):
        pytest.skip("live reference not present")
    r = live_ref.load_g2o3d(path)
    d = io.read_g2o3d(path)
    for k in ("v1", "v2", "noise_kind", "vertex_keys"):
        assert np.array_equal(d[k], r[k]), k
    assert np.abs(d["z"] - r["z"]).max() <= 1e-14
    assert np.abs(d["vertex_poses"] - r["vertex_poses"]).max() <= 1e-14
    assert np.abs(d["noise"] - r["noise"]).max() <= 1e-12 * max(1.0, np.abs(r["noise"]).max())


def test_bal_reader_on_a_written_file(tmp_path):
    """Format round trip on a file written here (BAL text layout: header, observations, cameras, points)."""
    rng = np.random.default_rng(0)
    nc, npt = 3, 5
    obs = [(i, j, rng.normal() * 100, rng.normal() * 100) for j in range(npt) for i in range(nc) if (i + j) % 2 == 0]
    cam = rng.normal(size=(nc, 9)) * [0.1, 0.1, 0.1, 1, 1, 1, 0, 0, 0] + [0, 0, 0, 0, 0, 0, 500, 1e-3, 1e-6]
    pts = rng.normal(size=(npt, 3))
    p = tmp_path / "tiny-bal.txt"
    with open(p, "w") as f:
        f.write(f"{nc} {npt} {len(obs)}\n")
        for i, j, u, v in obs:
            f.write(f"{i} {j} {u:.6e} {v:.6e}\n")
        for row in cam:
            f.write("\n".join(f"{x:.16e}" for x in row) + "\n")
        for row in pts:
            f.write("\n".join(f"{x:.16e}" for x in row) + "\n")
    cams, P, oc, op, oz = io.read_bal(str(p))
    assert cams.shape == (nc, 17) and P.shape == (npt, 3) and oc.size == len(obs)
    assert np.all(np.diff(op) >= 0)                                   # grouped by track
    assert np.allclose(P, pts.astype(np.float32)) and np.allclose(cams[:, 12], cam[:, 6].astype(np.float32))
    R = cams[:, :9].reshape(nc, 3, 3)
    assert np.abs(R @ np.swapaxes(R, 1, 2) - np.eye(3)).max() < 1e-14
    u0 = np.float32(f"{obs[0][2]:.6e}"); v0 = np.float32(f"{obs[0][3]:.6e}")
    k = np.where((oc == obs[0][0]) & (op == obs[0][1]))[0][0]
    assert oz[k, 0] == float(u0) and oz[k, 1] == -float(v0)           # (u, -v)


@pytest.mark.parametrize("fn,name", [("w100.graph", "pose2_w100"), ("noisyToyGraph.txt", "pose2_toy"), ("w20000.txt", "pose2_w20000")])
def test_load2d_reader_matches_reference_loader(fn, name):
    """io.read_2d against what the reference's load2D returned for its shipped 2-D files (golden fixtures)."""
    path = DATA + fn
    if not os.path.exists(path):
        pytest.skip("reference data not present on this machine")
    g = load_golden(name)
    d = io.read_2d(path)
    assert np.array_equal(d["v1"], g["v1"]) and np.array_equal(d["v2"], g["v2"])
    assert np.array_equal(d["noise_kind"], g["noise_kind"])
    assert np.abs(d["z"] - g["z"]).max() <= 1e-15 and np.abs(d["noise"] - g["noise"]).max() <= 1e-15
    n = int(max(d["v1"].max(), d["v2"].max())) + 1
    v0 = np.zeros((n, 3)); v0[d["vertex_keys"]] = d["vertex_poses"]
    # w20000.txt has no VERTEX lines: load2D chains 20 060 measurements (dataset.cpp:541-546), values up to 700
    assert np.abs(v0.reshape(-1) - g["values0"]).max() <= 1e-12 * max(1.0, np.abs(g["values0"]).max())


def test_load2d_reader_g2o_order_on_a_written_file(tmp_path, live_ref):
    """EDGE_SE2 with a full covariance in COV order (v0 v1 v2 v3 v4 v5 = upper triangle) and a TORO-order line."""
    p = tmp_path / "tiny2d.g2o"
    p.write_text("VERTEX_SE2 0 0 0 0\nVERTEX_SE2 1 1 0.1 0.2\nVERTEX_SE2 2 2 0.3 -0.1\n"
                 "EDGE_SE2 0 1 1.0 0.1 0.2 4 0 0 5 0 6\nEDGE2 1 2 1.0 0.2 -0.3 2 0 3 7 0 0\n")
    d = io.read_2d(str(p))
    assert list(d["noise_kind"]) == [2, 2]
    assert np.allclose(d["noise"][0, :3], np.sqrt([4, 5, 6])) and np.allclose(d["noise"][1, :3], np.sqrt([2, 3, 7]))
    if live_ref is not None:
        r = live_ref.load_2d(str(p))
        for k in ("v1", "v2", "noise_kind", "vertex_keys"):
            assert np.array_equal(d[k], r[k]), k
        assert np.abs(d["noise"] - r["noise"]).max() <= 1e-15 and np.abs(d["z"] - r["z"]).max() <= 1e-15


# ---- native BAL parser / writers (gtsam_amd/csrc/io.cpp behind the C ABI; host-only, no GPU) -----------------------------
def _tokens(path):
    out = []
    for t in open(path).read().split():
        try:
            out.append(float(t))
        except ValueError:
            out.append(t)
    return out


def _same_tokens(a, b, rtol):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        if isinstance(x, str) or isinstance(y, str):
            assert x == y
        else:
            assert abs(x - y) <= rtol * max(abs(x), abs(y)) + 1e-300, (x, y)


def test_native_bal_reader_matches_restatement_and_reference():
    path = DATA + "dubrovnik-3-7-pre.txt"
    if not os.path.exists(path):
        pytest.skip("reference data not present on this machine")
    nat = io.read_bal(path); py = io.read_bal_py(path)
    for a, b in zip(nat, py):
        assert a.shape == b.shape and a.dtype == b.dtype
    assert np.array_equal(nat[2], py[2]) and np.array_equal(nat[3], py[3]) and np.array_equal(nat[4], py[4])
    assert np.array_equal(nat[1], py[1]) and np.abs(nat[0] - py[0]).max() <= 1e-15 * np.abs(py[0]).max()
    g = load_golden("dubrovnik_3_7")                      # what SfmData::FromBalFile returned
    assert np.array_equal(nat[2], g["obs_cam"]) and np.array_equal(nat[4], g["obs_z"]) and np.array_equal(nat[1], g["pts"])
    assert np.abs(nat[0] - g["cams"]).max() <= 1e-14 * np.abs(g["cams"]).max()


def test_native_bal_roundtrip_and_errors(tmp_path):
    from gtsam_amd import datasets as D
    cams, pts, oc, op, oz = D.synthetic_bal(7, 60, seed=5)[:5]
    order = np.argsort(op, kind="stable")
    oc, op, oz = np.asarray(oc)[order], np.asarray(op)[order], np.asarray(oz).reshape(-1, 2)[order]
    p = str(tmp_path / "rt.txt")
    io.write_bal(p, cams, pts, oc, op, oz)
    c2, p2, oc2, op2, oz2 = io.read_bal(p)
    # the reader parses through float32 like the reference; the writer prints 20 digits
    assert np.array_equal(oc2, oc) and np.array_equal(op2, op)
    assert np.allclose(oz2, oz, rtol=1e-6, atol=1e-4) and np.allclose(p2, np.asarray(pts).reshape(-1, 3), rtol=1e-6, atol=1e-6)
    assert np.abs(c2[:, :12] - np.asarray(cams).reshape(-1, 17)[:, :12]).max() <= 1e-5
    assert np.allclose(c2[:, 12:15], np.asarray(cams).reshape(-1, 17)[:, 12:15], rtol=1e-6)
    with pytest.raises(RuntimeError, match="can not find the file"):
        io.read_bal(str(tmp_path / "missing.txt"))
    (tmp_path / "short.txt").write_text("2 2 3\n0 0 1.0 2.0\n")
    with pytest.raises(RuntimeError, match="truncated"):
        io.read_bal(str(tmp_path / "short.txt"))
    with pytest.raises(RuntimeError, match="grouped by point"):
        io.write_bal(p, cams, pts, oc[::-1].copy(), op[::-1].copy(), oz[::-1].copy())


def test_bal_writer_matches_reference_writer(tmp_path, live_ref):
    path = DATA + "dubrovnik-3-7-pre.txt"
    if live_ref is None or not os.path.exists(path):
        pytest.skip("live reference not present")
    ref_out = str(tmp_path / "ref.txt"); my_out = str(tmp_path / "mine.txt")
    assert live_ref.rewrite_bal(path, ref_out) == 0
    io.write_bal(my_out, *io.read_bal(path))
    _same_tokens(_tokens(my_out), _tokens(ref_out), 1e-12)
    # and the reference's loader reads our file back to the same SfmData
    a = live_ref.load_bal(my_out); b = live_ref.load_bal(path)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.allclose(a[4], b[4], rtol=1e-6)
    assert np.abs(a[0] - b[0]).max() <= 1e-5 and np.allclose(a[1], b[1], rtol=1e-6)


@pytest.mark.parametrize("name,is3d", [("pose3example.txt", True), ("pose3example-offdiagonal.txt", True), ("sphere2500.txt", True),
                                       ("noisyToyGraph.txt", False), ("w100.graph", False)])
def test_g2o_writer_matches_reference_writer(tmp_path, live_ref, name, is3d):
    path = DATA + name
    if live_ref is None or not os.path.exists(path):
        pytest.skip("live reference not present")
    ref_out = str(tmp_path / "ref.g2o"); my_out = str(tmp_path / "mine.g2o")
    live_ref.rewrite_g2o(path, ref_out, is3d)
    d = io.read_g2o3d(path) if is3d else io.read_2d(path)
    io.write_g2o(my_out, d)
    _same_tokens(_tokens(my_out), _tokens(ref_out), 2e-5)     # 6 significant digits in the file
    # reading our own output back gives the graph we wrote (to the printed precision)
    d2 = io.read_g2o3d(my_out) if is3d else io.read_2d(my_out)
    assert np.array_equal(d2["v1"], d["v1"]) and np.array_equal(d2["v2"], d["v2"])
    assert np.abs(d2["z"] - d["z"]).max() <= 2e-5 * max(1.0, np.abs(d["z"]).max())


def test_sfmdata_tests_of_the_reference_restated(tmp_path):
    """gtsam/sfm/tests/testSfmData.cpp:66-160 on the native reader / writer: readBAL_Dubrovnik (3 cameras, 7 tracks, track 0
    has 3 measurements starting with camera 0, its point projects within 12 px of the measurement), writeBAL_Dubrovnik
    (what is written reads back to the same cameras, points and measurements)."""
    path = DATA + "dubrovnik-3-7-pre.txt"
    if not os.path.exists(path):
        pytest.skip("reference data not present on this machine")
    from oracle import gtsam_oracle as O
    cams, pts, oc, op, oz = io.read_bal(path)
    assert cams.shape == (3, 17) and pts.shape == (7, 3)
    assert int((op == 0).sum()) == 3 and oc[0] == 0
    pi, _, _, behind = O.sfm_project(cams[0], pts[0][None])
    assert not behind[0] and np.abs(pi[0] - oz[0]).max() <= 12
    out = str(tmp_path / "rewritten.txt")
    io.write_bal(out, cams, pts, oc, op, oz)
    c2, p2, oc2, op2, oz2 = io.read_bal(out)
    assert np.array_equal(oc2, oc) and np.array_equal(op2, op)
    # assert_equal's 1e-9 on what went through the file's float parsing: the reader rounds to binary32 like the reference
    assert np.abs(c2 - cams).max() <= 1e-5 * np.abs(cams).max() and np.allclose(p2, pts, rtol=1e-6) and np.allclose(oz2, oz, rtol=1e-6)
    # openGL2gtsam / gtsam2openGL are inverse to each other (testSfmData.cpp:84-112): a camera written and read back keeps
    # its pose to the file's precision, for a pose with all three rotation components
    R = O.so3_expmap(np.array([[0.2, 0.7, 1.1]]))[0]
    cam = np.concatenate([R.reshape(-1), [1.0, 20.0, 10.0], [500.0, 1e-3, 1e-6, 0, 0]])[None]
    io.write_bal(out, cam, np.array([[0.0, 0.0, 30.0]]), np.array([0], np.int32), np.array([0], np.int32), np.array([[1.0, 2.0]]))
    c3 = io.read_bal(out)[0]
    assert np.abs(c3[0, :12] - cam[0, :12]).max() <= 2e-6 * 20 and abs(c3[0, 12] - 500.0) <= 1e-4


def _quat_wxyz_to_R(w, x, y, z):
    return io._quat(x, y, z, w)


def test_dataset_tests_of_the_reference_restated():
    """gtsam/slam/tests/testDataset.cpp on the Python readers: load2D on w100.graph (:91-103: 300 factors, 100 poses, first
    factor BetweenFactor(1, 0, Pose2(-0.99879, 0.0417574, -0.00818381), Unit)); readG2o3D on pose3example (:147-218: six
    relative poses and five vertex poses as quaternion / translation literals, Isotropic precision 10000, 1e-5);
    readG2o3DNonDiagonalNoise (:221-256: information matrix with 10000 on the diagonal and i+1 off it, 1e-2)."""
    if not os.path.exists(DATA + "w100.graph"):
        pytest.skip("reference data not present on this machine")
    d = io.read_2d(DATA + "w100.graph")
    assert d["v1"].size == 300 and d["vertex_keys"].size == 100
    assert (int(d["v1"][0]), int(d["v2"][0])) == (1, 0) and int(d["noise_kind"][0]) == NOISE_UNIT
    assert np.abs(d["z"][0] - [-0.99879, 0.0417574, -0.00818381]).max() <= 1e-9
    rel_q = [((0.854230, 0.190253, 0.283162, -0.392318), (1.001367, 0.015390, 0.004948)),
             ((0.105373, 0.311512, 0.656877, -0.678505), (0.523923, 0.776654, 0.326659)),
             ((0.568551, 0.595795, -0.561677, 0.079353), (0.910927, 0.055169, -0.411761)),
             ((0.542221, -0.592077, 0.303380, -0.513226), (0.775288, 0.228798, -0.596923)),
             ((0.327419, -0.125250, -0.534379, 0.769122), (-0.577841, 0.628016, -0.543592)),
             ((0.083672, 0.104639, 0.627755, 0.766795), (-0.623267, 0.086928, 0.773222))]
    pose_q = [((1.0, 0.0, 0.0, 0.0), (0, 0, 0)), ((0.854230, 0.190253, 0.283162, -0.392318), (1.001367, 0.015390, 0.004948)),
              ((0.421446, -0.351729, -0.597838, 0.584174), (1.993500, 0.023275, 0.003793)),
              ((0.067024, 0.331798, -0.200659, 0.919323), (2.004291, 1.024305, 0.018047)),
              ((0.765488, -0.035697, -0.462490, 0.445933), (0.999908, 1.055073, 0.020212))]
    edges = [(0, 1), (1, 2), (2, 3), (3, 4), (1, 4), (3, 0)]
    g = io.read_g2o3d(DATA + "pose3example.txt")
    assert list(zip(g["v1"].tolist(), g["v2"].tolist())) == edges and list(g["vertex_keys"]) == [0, 1, 2, 3, 4]
    for k, (q, t) in enumerate(rel_q):
        assert np.abs(g["z"][k, :9].reshape(3, 3) - _quat_wxyz_to_R(*q)).max() <= 1e-5 and np.abs(g["z"][k, 9:] - t).max() <= 1e-5
        assert int(g["noise_kind"][k]) == NOISE_ISOTROPIC and abs(g["noise"][k, 0] - 1.0 / np.sqrt(10000.0)) <= 1e-12   # Isotropic::Precision(6, 10000)
    for j, (q, t) in enumerate(pose_q):
        assert np.abs(g["vertex_poses"][j, :9].reshape(3, 3) - _quat_wxyz_to_R(*q)).max() <= 1e-5
        assert np.abs(g["vertex_poses"][j, 9:] - t).max() <= 1e-5
    o = io.read_g2o3d(DATA + "pose3example-offdiagonal.txt")
    assert o["v1"].size == 1 and int(o["noise_kind"][0]) == NOISE_GAUSSIAN
    info = np.array([[10000.0 if i == j else min(i, j) + 1 for j in range(6)] for i in range(6)])
    R = o["noise"][0].reshape(6, 6)
    # the file's t,R block order is swapped into GTSAM's R,t order on reading (dataset.cpp:848-853): the literal of the test
    # is already in GTSAM order
    assert np.abs(R.T @ R - info).max() <= 1e-2 * 10000
    assert np.abs(o["z"][0, 9:] - [1.001367, 0.015390, 0.004948]).max() <= 1e-5


def test_read_g2o_2d_reads_information_matrices_and_round_trips(tmp_path):
    """readG2o on a 2-D file is load2D with NoiseFormatG2O (slam/dataset.cpp:621-633): the six numbers are the upper triangle of the
    INFORMATION matrix.  What write_g2o writes, read_g2o reads back to the same noise; the AUTO format of load2D treats the same
    numbers as a covariance (the reference's behaviour for .graph files) -- hence the explicit format.  Duplicate vertices raise."""
    from gtsam_amd import io as IO
    p = tmp_path / "t.g2o"
    p.write_text("VERTEX_SE2 0 0 0 0\nVERTEX_SE2 1 1 0 0.1\nEDGE_SE2 0 1 1.0 0.0 0.1 4.0 0.0 0.0 9.0 0.0 16.0\n")
    d = IO.read_g2o(str(p))
    assert d["noise_kind"][0] == 2 and np.allclose(d["noise"][0, :3], [0.5, 1.0 / 3.0, 0.25])     # sigmas = 1 / sqrt(information)
    d_auto = IO.read_2d(str(p))                                                                   # AUTO -> COV: the numbers are variances
    assert np.allclose(d_auto["noise"][0, :3], [2.0, 3.0, 4.0])
    out = tmp_path / "w.g2o"
    IO.write_g2o(str(out), d, d["vertex_keys"], d["vertex_poses"])
    d2 = IO.read_g2o(str(out))
    assert np.allclose(d2["noise"], d["noise"]) and np.allclose(d2["z"], d["z"])
    p.write_text("VERTEX_SE2 0 0 0 0\nVERTEX_SE2 0 1 0 0.1\n")
    with pytest.raises(ValueError):
        IO.read_2d(str(p))
    with pytest.raises(ValueError):
        IO.read_2d(str(out), noise_format="nonsense")


# ---- native g2o / TORO readers and writer (gtsam_amd/csrc/io_g2o.cpp behind the C ABI; host-only, no GPU) ---------------------------
# io.read_g2o3d / io.read_2d / io.write_g2o ARE the native ones (the tests above compare them with the reference's loaders and its
# writer); here: against the token-by-token restatements (*_py) on the reference's files and on files written here, and the errors.
@pytest.mark.parametrize("name,is3d", [("pose3example.txt", True), ("pose3example-offdiagonal.txt", True), ("pose3example-grid.txt", True),
                                       ("sphere2500.txt", True), ("noisyToyGraph.txt", False), ("w100.graph", False), ("w20000.txt", False)])
def test_native_g2o_reader_and_writer_match_the_restatement(tmp_path, name, is3d):
    path = DATA + name
    if not os.path.exists(path):
        pytest.skip("reference data not present on this machine")
    nat = io.read_g2o3d(path) if is3d else io.read_2d(path)
    py = io.read_g2o3d_py(path) if is3d else io.read_2d_py(path)
    for k in ("v1", "v2", "noise_kind", "vertex_keys"):
        assert np.array_equal(nat[k], py[k]), k
    for k in ("z", "noise", "vertex_poses"):
        assert nat[k].shape == py[k].shape and (nat[k].size == 0 or np.abs(nat[k] - py[k]).max() <= 1e-13 * max(1.0, np.abs(py[k]).max())), k
    a, b = str(tmp_path / "nat.g2o"), str(tmp_path / "py.g2o")
    io.write_g2o(a, nat); io.write_g2o_py(b, py)
    assert open(a).read() == open(b).read()          # `stream << double` = %g on both sides


def _random_graph_3d(rng, n=9):
    from oracle import gtsam_oracle as O
    R = O.so3_expmap(rng.normal(0, 1.0, (n, 3)))
    poses = np.concatenate([R.reshape(n, 9), rng.normal(0, 3, (n, 3))], 1)
    v1 = np.arange(n - 1); v2 = v1 + 1
    Rz = O.so3_expmap(rng.normal(0, 0.5, (n - 1, 3)))
    z = np.concatenate([Rz.reshape(n - 1, 9), rng.normal(0, 1, (n - 1, 3))], 1)
    kinds = np.array([k % 4 for k in range(n - 1)], np.int32)           # unit, isotropic, diagonal, full Gaussian in turn
    noise = np.zeros((n - 1, 36))
    for k in range(n - 1):
        if kinds[k] == 1:
            noise[k, 0] = 0.3 + 0.1 * k
        elif kinds[k] == 2:
            noise[k, :6] = rng.uniform(0.1, 2.0, 6)
        elif kinds[k] == 3:
            A = rng.normal(size=(6, 6)); noise[k] = np.linalg.cholesky(A @ A.T + 6 * np.eye(6)).T.reshape(-1)
    return dict(v1=v1, v2=v2, z=z, noise_kind=kinds, noise=noise, vertex_keys=np.arange(n), vertex_poses=poses)


def test_native_g2o_full_precision_round_trip_3d_and_2d(tmp_path):
    """What gtg_io_write_g2o writes with full_precision, gtg_io_read_g2o reads back to rounding (quaternion and information-matrix round
    trips), with every noise-model kind; the restatement reads the same file to the same arrays."""
    rng = np.random.default_rng(11)
    d = _random_graph_3d(rng)
    p = str(tmp_path / "g3.g2o")
    io.write_g2o(p, d, full_precision=True)
    for back in (io.read_g2o3d(p), io.read_g2o3d_py(p)):
        assert np.array_equal(back["v1"], d["v1"]) and np.array_equal(back["vertex_keys"], d["vertex_keys"])
        assert np.array_equal(back["noise_kind"], d["noise_kind"])
        assert np.abs(back["z"] - d["z"]).max() <= 1e-14 and np.abs(back["vertex_poses"] - d["vertex_poses"]).max() <= 1e-14 * 10
        assert np.abs(back["noise"] - d["noise"]).max() <= 1e-12
    # 2-D: a pure odometry file (no VERTEX lines) is chained by the reader; G2O order = information matrix
    p2 = tmp_path / "odo.g2o"
    p2.write_text("EDGE_SE2 0 1 1.0 0.0 0.5 4 0 0 9 0 16\nEDGE_SE2 1 2 1.0 0.1 -0.2 1 0 0 1 0 1\nEDGE_SE2 2 0 -1.7 0.4 -0.3 5 1 0 6 0.5 7\n")
    nat, py = io.read_g2o(str(p2)), io.read_2d_py(str(p2), noise_format=io.NOISE_FORMAT_G2O)
    assert list(nat["vertex_keys"]) == [0, 1, 2] and list(nat["noise_kind"]) == [2, 0, 3]
    for k in ("z", "noise", "vertex_poses"):
        assert np.abs(nat[k] - py[k]).max() <= 1e-14, k
    assert np.allclose(nat["vertex_poses"][1], [1.0, 0.0, 0.5]) and np.allclose(nat["noise"][0, :3], [0.5, 1 / 3.0, 0.25])
    out = str(tmp_path / "w2.g2o")
    io.write_g2o(out, nat, full_precision=True)
    back = io.read_g2o(out)
    assert np.abs(back["noise"] - nat["noise"]).max() <= 1e-13 and np.abs(back["z"] - nat["z"]).max() <= 1e-15


def test_native_g2o_errors(tmp_path):
    with pytest.raises(ValueError, match="can not find file"):
        io.read_g2o3d(str(tmp_path / "missing.g2o"))
    p = tmp_path / "bad.g2o"
    p.write_text("VERTEX_SE2 0 0 0 0\nVERTEX_SE2 0 1 0 0.1\n")
    with pytest.raises(ValueError, match="twice"):
        io.read_2d(str(p))
    p.write_text("EDGE_SE2 0 1 1.0 0.0 0.5 4 1 1 9 1 16\n")
    with pytest.raises(ValueError, match="unrecognized covariance matrix format"):
        io.read_2d(str(p))                                   # AUTO cannot guess this zero pattern (dataset.cpp:218-232)
    with pytest.raises(ValueError, match="not TORO matrix order"):
        p.write_text("EDGE2 0 1 1.0 0.0 0.5 4 0 0 0 0 0\n"); io.read_2d(str(p), noise_format=io.NOISE_FORMAT_TORO)
    p.write_text("EDGE_SE3:QUAT 0 1 0 0 0 0 0 0 1 1 0 0\n")
    with pytest.raises(ValueError, match="malformed edge"):
        io.read_g2o3d(str(p))
    with pytest.raises(ValueError, match="invalid noise format"):
        io.read_2d(str(p), noise_format="nonsense")
