"""Pins for the two-view geometric-verification oracle: the reference's own tolerance-level
tests restated with seeded inputs, and the count pin of akaze/tests/estimate_pose.rs:75.  CPU only."""
import itertools

import numpy as np


def _rot(v):
    """Rotation3::new(axis-angle vector) via Rodrigues."""
    th = np.linalg.norm(v)
    if th == 0:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _scene(rng, n=16):
    """eight-point/tests/random.rs:38-75 (ROT_MAGNITUDE 0.2, box 2, distance 3), seeded."""
    R = _rot(rng.random(3) * np.pi * 2 * 0.2)
    t = rng.random(3)
    pts = rng.random((n, 3)) * 2.0
    pts[:, 0] -= 1.0; pts[:, 1] -= 1.0; pts[:, 2] += 3.0
    pb = pts @ R.T + t
    a = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    b = pb / np.linalg.norm(pb, axis=1, keepdims=True)
    return R, t, a, b


def test_eight_point_residuals_on_exact_data(oracle):
    """eight-point/tests/random.rs:7-36: all 16 essential residuals < 1e-4 in > 95 % of 1000 rounds."""
    rng = np.random.default_rng(123)
    ok = 0
    for _ in range(1000):
        R, t, a, b = _scene(rng)
        E = oracle.eight_point(a[:8], b[:8])
        assert E is not None
        an = a / a[:, 2:3]; bn = b / b[:, 2:3]
        res = np.abs(np.einsum("ni,ij,nj->n", bn, E, an))   # EssentialMatrix::residual, essential.rs:266-275
        ok += bool((res < 1e-4).all())
    assert ok > 950, ok


def test_pose_recovery_and_residual(oracle):
    """cv-pinhole/src/essential.rs:197-216 doc-test: one of the four poses matches the true pose (rotation
    angle < 1e-4, translation direction < 1e-4), and CameraToCamera::residual of exact matches is ~0."""
    rng = np.random.default_rng(5)
    hits = 0
    for _ in range(100):
        R, t, a, b = _scene(rng)
        E = oracle.eight_point(a[:8], b[:8])
        P = oracle.essential_poses(E)
        assert P is not None
        good = False
        for p in P:
            Rp, tp = p[:, :3], p[:, 3]
            assert abs(np.linalg.det(Rp) - 1) < 1e-9 and np.allclose(Rp @ Rp.T, np.eye(3), atol=1e-9)
            ang = np.arccos(np.clip((np.trace(Rp.T @ R) - 1) / 2, -1, 1))
            tres = 1 - np.dot(tp / np.linalg.norm(tp), t / np.linalg.norm(t))
            if ang < 1e-4 and tres < 1e-4:
                good = True
                r = [oracle.pose_residual(p, a[i], b[i]) for i in range(16)]
                assert max(r) < 1e-7, max(r)      # the tutorial's inlier threshold (ch5 main.rs:48)
        hits += good
    assert hits >= 95, hits


def test_estimate_pose_inlier_count(oracle, kitti_golden):
    """akaze/tests/estimate_pose.rs:28-32,63-75: the 11 Lowe matches of the KITTI pair, calibrated with K_00,
    all come out as inliers of the best eight-point model at threshold 0.1 (the reference asserts
    inliers.len() == 11).  ARRSAC's sampler is un-vendored: every 8-subset of the 11 matches is scored."""
    g = kitti_golden
    m = g["sparse_lowe"]
    assert len(m) == 11
    a = oracle.calibrate(g["sparse_kp0"][m[:, 0]], 984.2439, 980.8141, 690.0, 233.1966)
    b = oracle.calibrate(g["sparse_kp14"][m[:, 1]], 984.2439, 980.8141, 690.0, 233.1966)
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0)
    samples = np.array(list(itertools.combinations(range(11), 8)), np.uint32)
    out = oracle.essential_batch(a, b, samples, 0.1)
    assert out is not None
    pose, best, inl, counts = out
    assert len(inl) == 11 and inl.tolist() == list(range(11))
    assert counts.max() == 11


def test_calibrate_k1(oracle):
    """cv-pinhole/src/lib.rs:191-202 vs :108-117: k1 = 0 reproduces the plain arm; the doc-test relation
    nkp == simple_nkp / (1 + k1*|simple_nkp|^2) holds."""
    from oracle.oracle import KP_DTYPE
    kps = np.zeros(3, KP_DTYPE)
    kps["x"] = [471.0, 10.0, 1300.5]; kps["y"] = [322.0, 400.25, 12.0]
    plain = oracle.calibrate(kps, 800.0, 900.0, 500.0, 600.0, skew=1.7)
    k0 = oracle.calibrate(kps, 800.0, 900.0, 500.0, 600.0, skew=1.7, k1=0.0)
    assert np.array_equal(plain, k0)
    k1 = -0.3728755
    d = oracle.calibrate(kps, 800.0, 900.0, 500.0, 600.0, skew=1.7, k1=k1)
    s = plain[:, :2] / plain[:, 2:3]
    n = d[:, :2] / d[:, 2:3]
    assert np.allclose(n, s / (1 + k1 * (s ** 2).sum(1, keepdims=True)), atol=1e-12)


# ---- PnP: Lambda Twist (lambda-twist/tests/consensus.rs) --------------------------------------------------
def _euler(r, p, y):
    """Rotation3::from_euler_angles(roll, pitch, yaw) = Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]]); Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _projective(points):
    """Projective::from_point: homogeneous [x, y, z, 1] with xyz normalised (cv-core/src/point.rs:20-25,44-46)."""
    h = np.concatenate([points, np.ones((len(points), 1))], 1)
    return h / np.linalg.norm(h[:, :3], axis=1, keepdims=True)


def _bearings(xy):
    b = np.concatenate([xy, np.ones((len(xy), 1))], 1)
    return b / np.linalg.norm(b, axis=1, keepdims=True)


def arrsac_manual_scene():
    """lambda-twist/tests/consensus.rs:20-57."""
    cam = np.array([[-0.228125, -0.061458334, 1.0], [0.41875, -0.58125, 2.0], [1.128125, 0.878125, 3.0],
                    [-0.528125, 0.178125, 2.5], [-0.923424, -0.235125, 2.8]])
    R = _euler(0.1, 0.2, 0.3); t = np.array([0.1, 0.2, 0.3])
    world = (cam - t) @ R                       # pose.inverse() * p
    return R, t, _bearings(cam[:, :2] / cam[:, 2:3]), _projective(world)


def endless_loop_scene():
    """lambda-twist/tests/consensus.rs:72-127."""
    a = (0.3070512144698557, 0.19317668016026052); b = (0.3208462966353674, 0.20741702947913013)
    xy = np.array([a, b, a, b, b, a, (0.26619553978146293, 0.15033756455213498),
                   (0.3494806979265859, 0.18264329458710366), (0.32132193890323213, 0.15408143785084824)])
    pts = np.array([[1, 1, 0], [1, 1.5, 0], [3, 1, 0], [1, 2, 0], [2, 2, 0], [3, 2, 0], [1, 3, 0], [2, 3, 0], [3, 3, 0]], float)
    return _bearings(xy), _projective(pts)


def test_p3p_arrsac_manual_pose(oracle):
    """lambda-twist/tests/consensus.rs:17-67: the pose is recovered within 1e-6 (threshold 0.01)."""
    R, t, b, w = arrsac_manual_scene()
    samples = np.array(list(itertools.permutations(range(5), 3)), np.uint32)
    pose, best, inl, counts = oracle.p3p_batch(b, w, samples, 0.01)
    assert len(inl) == 5
    assert np.abs(pose[:, :3] - R).max() < 1e-6 and np.abs(pose[:, 3] - t).max() < 1e-6
    # every minimal sample yields between 1 and 4 candidate poses, the true one among them
    for s in samples[:10]:
        P = oracle.p3p_poses(b[s], w[s])
        assert 1 <= len(P) <= 4
        assert min(np.abs(p[:, :3] - R).max() for p in P) < 1e-6
        for p in P:
            assert max(oracle.w2c_residual(p, b[i], w[i]) for i in s) < 1e-9   # each pose explains its own sample


def test_p3p_endless_loop_case_terminates(oracle):
    """lambda-twist/tests/consensus.rs:69-134: a degenerate 9-sample case must terminate and give a model."""
    b, w = endless_loop_scene()
    samples = np.array(list(itertools.combinations(range(9), 3)), np.uint32)
    out = oracle.p3p_batch(b, w, samples, 0.01)
    assert out is not None and len(out[2]) >= 3
