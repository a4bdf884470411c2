"""R1-R3 and R5 on the DEVICE against checkers that share no source with the kernels.

The oracle (oracle/ransac_oracle.c, p3p_oracle.c) and the kernels (cv_amd/csrc/rs_ransac.hip) both compile
include/akz_ransac_math.h and include/akz_p3p_math.h, so "HIP == oracle" shows gcc == hipcc on one source for those rows
(round-3 verdict, weak spot).  Everything in this file is restated here from the reference's definitions with numpy /
LAPACK only — a different eigen-solver, a different P3P algorithm — and held to the device at the reference's own
tolerance (lambda-twist/tests/consensus.rs: EPSILON_APPROX = 1e-6), plus inlier-set identity:

  * P3P (R5, lambda-twist/src/lib.rs:107-318): 1 200 random minimal samples and every triple of the reference's
    degenerate nine-point scene (tests/consensus.rs:69-134).  The checker solves P3P as the intersection of two conics
    in the depth ratios (a quartic assembled with numpy.polynomial, roots by numpy.roots, pose by an SVD Procrustes fit)
    — nothing of Lambda Twist's cubic / eigen-decomposition — and the two solution sets must be the same;
  * eight-point + the four poses (R1, R2; eight-point/src/lib.rs:11-58, cv-pinhole/src/essential.rs:114-231): 1 024
    random minimal samples, batched numpy SVD null space;
  * CameraToCamera::residual (R3, cv-core/src/pose.rs:249-295): LAPACK eigh on the 4 x 4 design matrix for every
    (pose, match) of those hypotheses — a million of them — and the device's inlier COUNT of every pose and the winner's
    inlier SET equal the checker's wherever no residual sits within rounding of the threshold.
"""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cv_amd import build
    build.build()
    return True


# ------------------------------------------------------------------------------------------------------------------
# an independent P3P: two conics in (u, v) = (s2 / s1, s3 / s1)
def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-15:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _procrustes(X, Y):
    """R, t with Y_i = R X_i + t for three point pairs (least squares, SVD; det R = +1)."""
    cx, cy = X.mean(0), Y.mean(0)
    H = (X - cx).T @ (Y - cy)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, cy - R @ cx


def p3p_conics(f, X):
    """All poses (R, t) with f_i parallel to R X_i + t and positive depths.  f [3,3] unit bearings, X [3,3] points."""
    from numpy.polynomial import Polynomial as Poly
    a2 = ((X[1] - X[2]) ** 2).sum(); b2 = ((X[0] - X[2]) ** 2).sum(); c2 = ((X[0] - X[1]) ** 2).sum()
    ca, cb, cg = f[1] @ f[2], f[0] @ f[2], f[0] @ f[1]
    v = Poly([0.0, 1.0])
    g = 1 + v * v - 2 * cb * v                               # |s3 f3 - s1 f1|^2 / s1^2 = b^2 / s1^2
    # C1: b2 (u^2 + v^2 - 2 u v ca) = a2 g      C2: c2 g = b2 (1 + u^2 - 2 u cg);  C1 - C2 is linear in u: u = N / D
    N = b2 * v * v - b2 + (c2 - a2) * g
    D = 2 * b2 * (ca * v - cg)
    quartic = b2 * N * N - 2 * b2 * cg * N * D + (b2 - c2 * g) * D * D
    out = []
    for r in np.roots(quartic.coef[::-1]):
        if abs(r.imag) > 1e-7 * max(1.0, abs(r.real)) or r.real <= 0:
            continue
        vv = r.real
        for _ in range(3):                                   # polish the real root (Newton on the quartic)
            d = quartic.deriv()(vv)
            if d != 0:
                vv -= quartic(vv) / d
        Dv = D(vv)
        if abs(Dv) < 1e-12:
            continue
        uu = N(vv) / Dv
        gv = g(vv)
        if uu <= 0 or gv <= 0:
            continue
        s1 = np.sqrt(b2 / gv)
        Y = np.stack([s1 * f[0], s1 * uu * f[1], s1 * vv * f[2]])
        R, t = _procrustes(X, Y)
        fit = np.abs(X @ R.T + t - Y).max()
        if fit < 1e-7 * max(1.0, np.abs(Y).max()):          # the three side lengths are all reproduced: a true solution
            out.append((R, t))
    return out


def _pose_gap(Ra, ta, Rb, tb):
    return max(np.abs(Ra - Rb).max(), np.abs(ta - tb).max())


def _w2c_residual(R, t, f, Xh):
    """WorldToCamera::residual (cv-core/src/pose.rs:194-201): 1 - f . bearing([R | t] Xh), Xh projective (w >= 0)."""
    q = Xh[:, :3] @ R.T + np.outer(Xh[:, 3], t)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    return 1.0 - (f * q).sum(1)


def _projective(points):
    h = np.concatenate([points, np.ones((len(points), 1))], 1)
    return h / np.linalg.norm(h[:, :3], axis=1, keepdims=True)


def test_lambda_twist_on_the_device_against_an_independent_p3p(gpu):
    from cv_amd.ransac import EssentialConsensus
    rng = np.random.default_rng(0xD3C0)
    H = 1200
    f = np.zeros((H, 3, 3)); X = np.zeros((H, 3, 3)); Rgt = np.zeros((H, 3, 3)); tgt = np.zeros((H, 3))
    for h in range(H):
        R = _rodrigues((rng.random(3) - 0.5) * 2.0)
        t = (rng.random(3) - 0.5) * 2.0
        Yc = np.stack([rng.uniform(-1.5, 1.5, 3), rng.uniform(-1.0, 1.0, 3), rng.uniform(2.0, 8.0, 3)], 1)   # camera frame
        X[h] = (Yc - t) @ R                                  # world = R^T (Yc - t)
        f[h] = Yc / np.linalg.norm(Yc, axis=1, keepdims=True)
        Rgt[h], tgt[h] = R, t
    bearings = f.reshape(-1, 3)
    world = _projective(X.reshape(-1, 3))
    samples = np.arange(3 * H, dtype=np.uint32).reshape(H, 3)
    cons = EssentialConsensus(3 * H, H)
    assert cons.p3p_model_inliers(bearings, world, samples, 1e-9) is not None
    P, ok = cons.poses(H)
    n_dev = n_ind = skipped = 0
    worst_constraint = worst_gt = 0.0
    for h in range(H):
        dev = [(P[h, p, :, :3], P[h, p, :, 3]) for p in range(4) if ok[h, p]]
        ind = p3p_conics(f[h], X[h])
        assert dev, h
        n_dev += len(dev); n_ind += len(ind)
        for R, t in dev:
            # a device pose is a rigid motion that puts the three points on their bearings
            assert abs(np.linalg.det(R) - 1.0) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9, h
            res = _w2c_residual(R, t, f[h], _projective(X[h]))
            worst_constraint = max(worst_constraint, res.max())
            assert res.max() < 1e-10, (h, res)
            assert ((X[h] @ R.T + t) * f[h]).sum(1).min() > 0, h          # in front of the camera
        # the pose the sample was made from is among them, at the reference's tolerance (consensus.rs: 1e-6)
        gap = min(_pose_gap(R, t, Rgt[h], tgt[h]) for R, t in dev)
        worst_gt = max(worst_gt, gap)
        assert gap < 1e-6, (h, gap)
        # ... and the two solvers found the same set of solutions.  Where two roots of the quartic nearly coincide the
        # root finder of the CHECKER loses digits (or a root pair turns complex); such samples are counted, not compared.
        same = bool(ind) and all(min(_pose_gap(*d, *i) for i in ind) < 1e-6 for d in dev) and \
            all(min(_pose_gap(*d, *i) for d in dev) < 1e-6 for i in ind)
        if not same:
            skipped += 1
    assert skipped < H // 50, skipped                         # near-double roots are rare among random samples
    assert n_dev >= H and abs(n_dev - n_ind) <= 4 * skipped, (n_dev, n_ind, skipped)
    print(f"P3P: {H} samples, {n_dev} device poses / {n_ind} checker poses, {skipped} near-double skipped, "
          f"worst constraint residual {worst_constraint:.2e}, worst distance to the generating pose {worst_gt:.2e}")


def test_lambda_twist_on_the_reference_degenerate_scene(gpu):
    """lambda-twist/tests/consensus.rs:69-134 (the nine co-planar points with repeated bearings): every triple goes
    through the device; whatever it returns satisfies the constraints it was asked to satisfy, and the consensus over all
    84 triples has exactly the inlier set numpy finds for the winning pose at the reference's threshold 0.01."""
    from cv_amd.ransac import EssentialConsensus
    a = (0.3070512144698557, 0.19317668016026052); b = (0.3208462966353674, 0.20741702947913013)
    xy = np.array([a, b, a, b, b, a, (0.26619553978146293, 0.15033756455213498),
                   (0.3494806979265859, 0.18264329458710366), (0.32132193890323213, 0.15408143785084824)])
    pts = np.array([[1, 1, 0], [1, 1.5, 0], [3, 1, 0], [1, 2, 0], [2, 2, 0], [3, 2, 0], [1, 3, 0], [2, 3, 0], [3, 3, 0]], float)
    fb = np.concatenate([xy, np.ones((9, 1))], 1)
    fb = fb / np.linalg.norm(fb, axis=1, keepdims=True)
    world = _projective(pts)
    samples = np.array(list(itertools.combinations(range(9), 3)), np.uint32)
    cons = EssentialConsensus(64, len(samples))
    got = cons.p3p_model_inliers(fb, world, samples, 0.01)
    assert got is not None                                     # the reference's assertion: a model comes out
    pose, inl, best = got
    P, ok = cons.poses(len(samples))
    # The scene repeats two bearings for different points, so most triples contradict themselves and have no exact pose:
    # Lambda Twist then returns what its algebra gives (the reference does not validate either; the consensus weeds it
    # out).  What can be held: every returned pose is a rigid motion, and where the independent solver finds an exact
    # solution of a triple the device has it too.
    checked = 0
    for h, tri in enumerate(samples):
        for p in range(4):
            if ok[h, p]:
                R = P[h, p, :, :3]
                assert abs(np.linalg.det(R) - 1.0) < 1e-8 and np.abs(R @ R.T - np.eye(3)).max() < 1e-8, (h, p)
        if len({tuple(fb[i]) for i in tri}) < 3:
            continue
        for Ri, ti in p3p_conics(fb[tri], pts[tri]):
            if _w2c_residual(Ri, ti, fb[tri], world[tri]).max() > 1e-12:
                continue
            dev = [(P[h, p, :, :3], P[h, p, :, 3]) for p in range(4) if ok[h, p]]
            assert dev and min(_pose_gap(Ri, ti, Rd, td) for Rd, td in dev) < 1e-6, (h, tri)
            checked += 1
    assert checked > 0
    res = _w2c_residual(pose[:, :3], pose[:, 3], fb, world)
    want = np.nonzero(res < 0.01)[0]
    assert np.array_equal(inl, want), (inl, want, res)
    # no pose of any triple has more inliers than the winner (consensus = arg max of the inlier count)
    best_cnt = 0
    for h in range(len(samples)):
        for p in range(4):
            if ok[h, p]:
                best_cnt = max(best_cnt, int((_w2c_residual(P[h, p, :, :3], P[h, p, :, 3], fb, world) < 0.01).sum()))
    assert best_cnt == len(want)


# ------------------------------------------------------------------------------------------------------------------
def _np_residuals(poses, a, b):
    """CameraToCamera::residual (cv-core/src/pose.rs:249-295) for every (pose, match): poses [P,3,4], a, b [n,3] ->
    [P, n].  Triangulation = the eigenvector of the smallest eigenvalue of the 4 x 4 design matrix (LAPACK eigh)."""
    Pn, n = len(poses), len(a)
    P0 = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    Ta = P0[None] - np.einsum("ni,nj,jk->nik", a, a, P0)                 # [n,3,4]
    Da = np.einsum("nij,nik->njk", Ta, Ta)                               # [n,4,4]
    out = np.zeros((Pn, n))
    for p in range(Pn):
        Tb = poses[p][None] - np.einsum("ni,nj,jk->nik", b, b, poses[p])
        D = Da + np.einsum("nij,nik->njk", Tb, Tb)
        w, V = np.linalg.eigh(D)                                         # ascending: column 0
        x = V[:, :, 0]
        neg = (x[:, 3] < 0) | ((x[:, 3] == 0) & np.signbit(x[:, 3]))
        x = np.where(neg[:, None], -x, x)
        x = x / np.linalg.norm(x[:, :3], axis=1, keepdims=True)
        q = x[:, :3] @ poses[p][:, :3].T + np.outer(x[:, 3], poses[p][:, 3])
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
        out[p] = 0.5 * ((1.0 - (a * x[:, :3]).sum(1)) + (1.0 - (b * q).sum(1)))
    return out


def test_eight_point_poses_and_inlier_sets_against_lapack(gpu):
    from cv_amd.ransac import EssentialConsensus
    rng = np.random.default_rng(0xE19)
    n, H, thr = 256, 1024, 1e-6
    # a rigid motion seen by two cameras, bearing noise 3e-4 rad, a third of the matches unrelated
    R = _rodrigues((rng.random(3) - 0.5) * 0.5)
    t = rng.random(3) - 0.5
    pts = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(3, 5, n)], 1)
    pb = pts @ R.T + t
    a = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    b = pb / np.linalg.norm(pb, axis=1, keepdims=True)
    b = b + rng.standard_normal((n, 3)) * 3e-4
    bad = rng.random(n) < 0.33
    rb = rng.standard_normal((n, 3)); rb[:, 2] = np.abs(rb[:, 2]) + 0.5
    b[bad] = rb[bad]
    b = b / np.linalg.norm(b, axis=1, keepdims=True)
    good = np.nonzero(~bad)[0]
    samples = np.stack([rng.choice(good if h % 2 == 0 else n, 8, replace=False) for h in range(H)]).astype(np.uint32)
    cons = EssentialConsensus(n, H)
    got = cons.model_inliers(a, b, samples, thr)
    assert got is not None
    pose, inl, best = got
    P, ok = cons.poses(H)
    counts = cons.counts(H)
    assert ok.all()
    # R1: the essential matrix behind the device's poses spans the null space LAPACK finds for the 8 x 9 system
    sa, sb = a[samples], b[samples]                                      # [H,8,3]
    A = np.einsum("hni,hnj->hnij", sa / sa[:, :, 2:3], sb / sa[:, :, 2:3]).reshape(H, 8, 9)   # kron(a / a.z, b / a.z): eight-point/src/lib.rs:11-24
    U, S, Vt = np.linalg.svd(A)
    E_raw = Vt[:, -1, :].reshape(H, 3, 3).transpose(0, 2, 1)            # Matrix3::from_iterator is column-major
    # possible_unscaled_poses reads only U and V of E's SVD (cv-pinhole/src/essential.rs:114-231): the poses describe the
    # essential matrix U diag(1, 1, 0) V^T nearest to the (noisy) null vector, not the null vector itself
    Ue, Se, Vte = np.linalg.svd(E_raw)
    E_ref = Ue @ np.diag([1.0, 1.0, 0.0]) @ Vte
    E_ref = E_ref / np.linalg.norm(E_ref.reshape(H, 9), axis=1)[:, None, None]
    tv = P[:, :, :, 3]
    Rm = P[:, :, :, :3]
    tx = np.zeros((H, 4, 3, 3))
    tx[..., 0, 1], tx[..., 0, 2] = -tv[..., 2], tv[..., 1]
    tx[..., 1, 0], tx[..., 1, 2] = tv[..., 2], -tv[..., 0]
    tx[..., 2, 0], tx[..., 2, 1] = -tv[..., 1], tv[..., 0]
    E = tx @ Rm
    E = E / np.linalg.norm(E.reshape(H, 4, 9), axis=2)[..., None, None]
    align = np.abs((E * E_ref[:, None]).sum((2, 3)))
    # (an 8 x 9 system whose two smallest singular values are close has no well-defined null vector: compare where it has)
    well = (S[:, 7] > 1e-6 * S[:, 0]) & (Se[:, 1] > 1e-3 * Se[:, 0])     # ... and E has a well-defined rank-2 part
    assert well.sum() > 0.9 * H
    assert np.abs(align[well] - 1.0).max() < 1e-8, np.abs(align[well] - 1.0).max()
    # R2: four proper poses: R in SO(3), |t| = 1, (t, R1), (t, R2), (-t, R1), (-t, R2)
    assert np.abs(np.linalg.det(Rm) - 1.0).max() < 1e-9
    assert np.abs(Rm @ Rm.transpose(0, 1, 3, 2) - np.eye(3)).max() < 1e-9
    assert np.abs(np.linalg.norm(tv, axis=2) - 1.0).max() < 1e-9
    assert np.array_equal(Rm[:, 0], Rm[:, 2]) and np.array_equal(Rm[:, 1], Rm[:, 3]) and np.allclose(tv[:, 0], -tv[:, 2], atol=0)
    # R3 + consensus: LAPACK residuals of all 4 H poses against all n matches; counts and the winner's inlier set
    res = _np_residuals(P.reshape(-1, 3, 4), a, b).reshape(H, 4, n)
    near = np.abs(res - thr) < 1e-9 * thr + 1e-15                        # a residual within rounding of the threshold
    clean = ~near.any(axis=2)
    want_counts = (res < thr).sum(2)
    assert clean.mean() > 0.99
    assert np.array_equal(counts[clean], want_counts[clean].astype(np.uint32)), np.nonzero(counts[clean] != want_counts[clean])
    bh, bp = divmod(int(best), 4)
    assert clean[bh, bp]
    assert np.array_equal(inl, np.nonzero(res[bh, bp] < thr)[0])
    # the winner is the arg max of (count, lowest id) — by the checker's counts
    flat = want_counts.reshape(-1)
    assert flat[best] == flat.max() and (flat[:best] < flat.max()).all()
    # ... and it is the scene's motion: the inliers are (nearly all of) the true matches, the rotation is R to the noise level
    assert len(inl) > 0.6 * len(good) and np.isin(inl, good).mean() > 0.98
    assert np.abs(pose[:, :3] - R).max() < 5e-3
    print(f"eight-point: {H} samples ({int(well.sum())} well-conditioned), {4 * H * n} residuals, {int((~clean).sum())} poses with a "
          f"residual at the threshold, winner {best} with {len(inl)} inliers")
