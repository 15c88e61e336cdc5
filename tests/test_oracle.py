"""-m "not gpu": the oracle against its committed golden vectors, against libm, and against independent
numpy restatements of the pieces where the reference delegates to a library."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from alego_amd import synth
from oracle import oracle_py as O
from util import assert_bit_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cfgA.npz")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_params_layout(params_a):
    assert C.sizeof(params_a) == synth.lib().alego_synth_params_sizeof()
    assert params_a.horizon_scan == 1800 and abs(params_a.ang_res_x - 0.2) < 1e-12
    ref = synth.default_params(16, 0)  # reference geometry utility.h:50-55
    assert ref.horizon_scan == 4000 and ref.ang_res_x == 0.09 and ref.ground_scan_id == 10


@pytest.mark.parametrize("name,n_scan,horizon,scans", [("oracle_cfgA.npz", 16, 1800, 12), ("oracle_geo_16x4000.npz", 16, 4000, 3), ("oracle_geo_64x2048.npz", 64, 2048, 3)])
def test_oracle_matches_golden(name, n_scan, horizon, scans):
    """the committed vectors (tests/golden/make_golden.py): a regression pin of the oracle — the reference holds no vectors of its own —
    at the bench geometry, the reference's own 16 x 4000 (utility.h:50-55) and config 5's 64 x 2048"""
    g = np.load(os.path.join(os.path.dirname(GOLD), name))
    p = synth.default_params(n_scan, horizon)
    o = O.Oracle(p)
    for k in range(scans):
        pts = synth.scan(p, k)
        assert digest(pts) == str(g[f"s{k}_in_digest"]), f"synthetic scan {k} changed"
        r = o.process_scan(pts)
        assert r == int(g[f"s{k}_ret"])
        for name_ in ("label_img", "seg_cloud", "seg_col", "seg_range", "seg_ground", "outlier", "less_sharp", "less_flat"):
            assert digest(o.get(name_)) == str(g[f"s{k}_{name_}_digest"]), f"scan {k} {name_}"
        for name_ in ("ring_start", "ring_end", "sharp_idx", "flat_idx", "lo_solve_info", "lm_info"):
            assert_bit_equal(o.get(name_), g[f"s{k}_{name_}"], f"scan {k} {name_}")
        for name_ in ("orientation", "lo_params", "odom_pose", "map_pose", "lm_params"):
            assert_bit_equal(o.get(name_), g[f"s{k}_{name_}"], f"scan {k} {name_}")


def test_atan2f_hypotf_equal_this_hosts_libm():
    """the fdlibm restatement must reproduce glibc's atan2f bit for bit."""
    rng = np.random.default_rng(11)
    n = 4_000_000
    L = O.lib()
    for scale in (1.0, 40.0):
        x = (rng.standard_normal(n) * scale).astype(np.float32)
        y = (rng.standard_normal(n) * scale).astype(np.float32)
        a, b = np.empty(n, np.float32), np.empty(n, np.float32)
        L.oracle_atan2f_array(y.ctypes.data, x.ctypes.data, a.ctypes.data, n)
        L.oracle_libm_atan2f_array(y.ctypes.data, x.ctypes.data, b.ctypes.data, n)
        assert_bit_equal(a, b, "atan2f")
        L.oracle_hypotf_array(x.ctypes.data, y.ctypes.data, a.ctypes.data, n)
        L.oracle_libm_hypotf_array(x.ctypes.data, y.ctypes.data, b.ctypes.data, n)
        assert_bit_equal(a, b, "hypotf")
    sp = np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-30, 1e30, 3e-39], np.float32)
    yy, xx = [np.ascontiguousarray(v.reshape(-1)) for v in np.meshgrid(sp, sp)]
    a, b = np.empty(yy.size, np.float32), np.empty(yy.size, np.float32)
    L.oracle_atan2f_array(yy.ctypes.data, xx.ctypes.data, a.ctypes.data, yy.size)
    L.oracle_libm_atan2f_array(yy.ctypes.data, xx.ctypes.data, b.ctypes.data, yy.size)
    assert_bit_equal(a, b, "atan2f special values")


def test_sinf_cosf_equal_this_hosts_libm():
    """transformPointCloud (laserMapping.h:166-173) builds its f32 matrix from std::sin / std::cos of FLOAT half angles:
    glibc's sinf / cosf, which are not the correctly rounded values.  The restatement shared by the oracle and the HIP
    path must reproduce this host's libm bit for bit over the range a key pose can reach (half angles in [-pi/2, pi/2];
    tested on [-3.2, 3.2] and around 0); (float)sin((double)x), which round 1 used on both sides, does not."""
    rng = np.random.default_rng(17)
    L = O.lib()
    n = 6_000_000
    x = np.concatenate([rng.uniform(-3.2, 3.2, n), rng.uniform(-1e-3, 1e-3, n // 10), [0.0, -0.0, np.pi / 4, -np.pi / 4, np.pi / 2, 1.5707964]]).astype(np.float32)
    a, b = np.empty(x.size, np.float32), np.empty(x.size, np.float32)
    for mode, name in ((0, "sinf"), (1, "cosf")):
        L.oracle_sincosf_array(x.ctypes.data, a.ctypes.data, x.size, mode)
        L.oracle_sincosf_array(x.ctypes.data, b.ctypes.data, x.size, mode + 2)
        assert_bit_equal(a, b, name)
    L.oracle_sincosf_array(x.ctypes.data, a.ctypes.data, x.size, 2)
    rounded = np.sin(x.astype(np.float64)).astype(np.float32)
    assert (a.view(np.uint32) != rounded.view(np.uint32)).mean() > 1e-3, "libm sinf is expected to differ from the correctly rounded sine"


def test_cell_centred_rays_land_in_their_cell(params_a):
    """the synthetic rays are cell-centred, so row/col are immune to libm ulp differences."""
    p = params_a
    o = O.Oracle(p)
    pts = synth.scan(p, 0)
    o.ip(pts)
    rimg = o.get("range_img").reshape(p.n_scan, p.horizon_scan)
    r = np.sqrt((pts[:, :3].astype(np.float32) ** 2).sum(1, dtype=np.float32))
    assert (rimg >= 0).sum() == len(pts)  # one cell per ray, no collisions
    assert np.allclose(np.sort(rimg[rimg >= 0]), np.sort(r), rtol=0, atol=1e-5)


def _labels_independent(rimg, ground, p):
    """Connected components by an independent union-find (numpy/python), numbering as imageProjection.cpp:147-156."""
    NS, H = rimg.shape
    active = (rimg >= 0) & (ground == 0)
    parent = np.arange(NS * H)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    def crit(r1, r2, alpha):
        d1, d2 = np.maximum(r1, r2).astype(np.float64), np.minimum(r1, r2).astype(np.float64)
        return np.arctan2(d2 * np.sin(alpha), d1 - d2 * np.cos(alpha)) > p.seg_theta

    right = active & np.roll(active, -1, axis=1) & crit(rimg, np.roll(rimg, -1, axis=1), p.seg_alpha_x)
    down = np.zeros_like(active)
    down[:-1] = active[:-1] & active[1:] & crit(rimg[:-1], rimg[1:], p.seg_alpha_y)
    for i, j in zip(*np.nonzero(right)):
        a, b = find(i * H + j), find(i * H + (j + 1) % H)
        if a != b:
            parent[max(a, b)] = min(a, b)
    for i, j in zip(*np.nonzero(down)):
        a, b = find(i * H + j), find((i + 1) * H + j)
        if a != b:
            parent[max(a, b)] = min(a, b)
    roots = np.array([find(v) if active.flat[v] else -1 for v in range(NS * H)])
    lab = np.full(NS * H, -1, np.int32)
    cnt = 0
    for r in np.unique(roots[roots >= 0]):  # ascending root == discovery order
        members = np.nonzero(roots == r)[0]
        rows = np.unique(members // H).size
        feasible = members.size >= p.seg_big_num or (members.size >= p.seg_valid_point_num and rows >= p.seg_valid_line_num)
        if feasible:
            cnt += 1
            lab[members] = cnt
        else:
            lab[members] = 999999
    return lab


def test_bfs_labels_equal_independent_union_find(params_a):
    p = params_a
    o = O.Oracle(p)
    o.ip(synth.scan(p, 2))
    rimg = o.get("range_img").reshape(p.n_scan, p.horizon_scan)
    ground = o.get("ground_img").reshape(p.n_scan, p.horizon_scan)
    want = _labels_independent(rimg, ground, p)
    assert_bit_equal(o.get("label_img"), want, "label image vs independent connected components")


def _voxel_numpy(pts, leaf):
    leaf = np.float32(leaf)
    inv = np.float32(1.0) / leaf
    mn, mx = pts[:, :3].min(0), pts[:, :3].max(0)
    minb = np.floor(mn * inv).astype(np.int64)
    div = np.floor(mx * inv).astype(np.int64) - minb + 1
    ijk = (np.floor(pts[:, :3] * inv) - minb.astype(np.float32)).astype(np.int64)
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    out = []
    i = 0
    while i < len(order):
        j = i
        acc = np.zeros(4, np.float32)
        while j < len(order) and idx[order[j]] == idx[order[i]]:
            acc = acc + pts[order[j]]
            j += 1
        out.append(acc / np.float32(j - i))
        i = j
    return np.array(out, np.float32)


def test_voxel_grid_equals_numpy_restatement(params_a):
    rng = np.random.default_rng(5)
    pts = (rng.standard_normal((3000, 4)) * [4, 4, 1, 1]).astype(np.float32)
    for leaf in (0.4, 0.8, 1.0):
        assert_bit_equal(O.voxel_grid(pts, leaf), _voxel_numpy(pts, leaf), f"voxel grid leaf {leaf}")
    assert O.voxel_grid(pts[:0], 0.4).shape[0] == 0
    one = O.voxel_grid(pts[:1], 0.4)
    assert_bit_equal(one, pts[:1], "single point")
    # dx*dy*dz exceeds INT_MAX: PCL warns "leaf size is too small" and returns the input unchanged
    assert_bit_equal(O.voxel_grid(pts, 0.001), pts, "leaf too small -> passthrough")


def test_kdtree_knn_equals_brute_force():
    rng = np.random.default_rng(9)
    cloud = (rng.standard_normal((5000, 4)) * 5).astype(np.float32)
    cloud[100] = cloud[7]  # exact duplicate: tie broken by the lowest index
    q = (rng.standard_normal((200, 4)) * 5).astype(np.float32)
    q[0] = cloud[7]
    idx, dist = O.knn(cloud, q, 5)
    for i in range(len(q)):
        d = np.zeros(len(cloud), np.float32)
        for a in range(3):
            df = cloud[:, a] - q[i, a]
            d = d + df * df
        order = np.lexsort((np.arange(len(cloud)), d))[:5]
        assert_bit_equal(idx[i], order.astype(np.int32), f"query {i} indices")
        assert_bit_equal(dist[i], d[order], f"query {i} distances")


def _plane_neighbourhoods(rng):
    """(name, 5 x 3 points) cases for the plane fit of laserMapping.cpp:425-452: what a voxel-filtered map hands the 5-NN search."""
    out = []
    for k in range(300):      # well-conditioned: five noisy points of a random plane 5 - 60 m from the origin
        n = rng.standard_normal(3); n /= np.linalg.norm(n)
        u = np.cross(n, rng.standard_normal(3)); u /= np.linalg.norm(u); v = np.cross(n, u)
        c = n * rng.uniform(5, 60)
        out.append(("plane", c + np.outer(rng.uniform(-0.5, 0.5, 5), u) + np.outer(rng.uniform(-0.5, 0.5, 5), v) + rng.normal(0, 0.01, (5, 3))))
    for k in range(300):      # near-collinear: a line with 1e-7 .. 1e-3 of lateral noise (a pole, the edge of a wall)
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        out.append(("near_collinear", rng.normal(0, 20, 3) + np.outer(np.linspace(-0.8, 0.8, 5), d) + rng.normal(0, 10.0 ** rng.uniform(-7, -3), (5, 3))))
    for k in range(100):      # exactly collinear along an axis: two columns are constant (axis-aligned, quantised surfaces)
        a = rng.integers(0, 3)
        p = np.tile(np.round(rng.normal(0, 20, 3) * 4) / 4, (5, 1)); p[:, a] += np.arange(5) * 0.4
        out.append(("axis_collinear", p))
    for k in range(100):      # exactly collinear, f32 coordinates, general direction
        d = rng.standard_normal(3)
        out.append(("collinear", (rng.normal(0, 20, 3) + np.outer(np.arange(5.0), d)).astype(np.float32).astype(np.float64)))
    for k in range(50):       # five coincident points / four coincident + one
        p = np.tile(rng.normal(0, 20, 3), (5, 1))
        if k % 2: p[4] += rng.normal(0, 0.3, 3)
        out.append(("duplicates", p))
    for k in range(100):      # a plane through the origin (n . p = 0: A x = -1 has no solution; the least-squares one is finite)
        n = rng.standard_normal(3); n /= np.linalg.norm(n)
        q = rng.normal(0, 5, (5, 3)); q -= np.outer(q @ n, n)
        out.append(("through_origin", q + rng.normal(0, 1e-3, (5, 3))))
    for a in range(3):        # ... exactly: one coordinate identically zero
        q = rng.normal(0, 5, (5, 3)); q[:, a] = 0.0
        out.append(("zero_column", q))
    out.append(("zero", np.zeros((5, 3))))
    return out


def test_plane_fit_is_eigens_column_pivoted_qr():
    """`matA0.colPivHouseholderQr().solve(matB0)` (laserMapping.cpp:435): the oracle's colpiv_qr_solve restates Eigen 3.3's ColPivHouseholderQR (pivoting by the largest
    remaining column norm, nonzeroPivots() by Eigen's tiny threshold, components beyond it ZERO) — checked against LAPACK's pivoted QR (scipy.linalg.qr(pivoting=True): the
    same pivot rule) used as the independent statement of "basic solution of rank r", and against numpy.linalg.lstsq where the system has full rank.  The rank itself is
    re-derived from LAPACK's R diagonal with Eigen's published threshold, cases within a factor 16 of it excepted (there the two roundings may legitimately differ).
    Round-5 review: the unpivoted Householder QR this replaces divided by a ~1e-17 pivot on collinear / coincident neighbourhoods; Eigen returns a finite normal there."""
    import scipy.linalg
    rng = np.random.default_rng(77)
    L = O.lib()
    eps = np.finfo(np.float64).eps
    seen = {}
    for name, pts in _plane_neighbourhoods(rng):
        A = np.asfortranarray(pts.astype(np.float64))
        b = -np.ones(5)
        Aw = A.copy(order="F")
        x = np.full(3, np.nan)
        r = L.oracle_colpiv_qr_solve(Aw.ctypes.data, b.ctypes.data, 5, 3, x.ctypes.data)
        if name == "zero":   # Eigen's test is a strict `<` against a threshold that is itself 0 here: all three pivots count, 0 / 0 — kept as Eigen has it (five map points
            assert r == 3    # at the origin cannot come out of a VoxelGrid: its output points lie in distinct voxels)
            continue
        assert np.all(np.isfinite(x)), (name, pts, x)
        Q, R, P = scipy.linalg.qr(A, pivoting=True)
        diag = np.abs(np.diag(R))
        maxn = np.sqrt((A * A).sum(axis=0)).max()
        helper = (maxn * eps) ** 2 / 5.0
        want_r, sure = 3, True
        for k in range(3):
            t = helper * (5 - k)
            if diag[k] ** 2 < t * 16 and diag[k] ** 2 > t / 16: sure = False
            if diag[k] ** 2 < t: want_r = k; break
        if sure: assert r == want_r, (name, pts, r, want_r, diag, helper)
        seen.setdefault(name, set()).add(r)
        # pivot order: LAPACK's choice wherever the competing column norms are clearly apart
        # basic solution of rank r from LAPACK's factorisation
        xr = np.zeros(3)
        if r > 0:
            c = (Q.T @ b)[:r]
            xr[P[:r]] = scipy.linalg.solve_triangular(R[:r, :r], c)
        # same pivots? (ties between nearly equal column norms may swap: then compare through the residual instead)
        cond = diag[0] / max(diag[r - 1], 1e-300) if r > 0 else 1.0
        scale = max(np.abs(xr).max(), 1e-300)
        if r == 3:
            np.testing.assert_allclose(x, xr, rtol=0, atol=1e-12 * cond * scale, err_msg=name)
            xl = np.linalg.lstsq(A, b, rcond=None)[0]
            if cond < 1e6: np.testing.assert_allclose(x, xl, rtol=0, atol=1e-10 * cond * np.abs(xl).max(), err_msg=name)
        else:
            # rank deficient: the zeroed components are exactly zero, the others solve the reduced problem; with a different (tied) pivot choice the
            # basic solution differs but the residual is the same least-squares residual
            assert np.count_nonzero(x) <= r, (name, x, r)
            res, res_r = np.linalg.norm(A @ x - b), np.linalg.norm(A @ xr - b)
            assert abs(res - res_r) <= 1e-9 * max(1.0, res_r), (name, res, res_r)
            if sure and np.all(np.abs(np.diff(np.sqrt((A * A).sum(axis=0)))) > 1e-6 * maxn):
                np.testing.assert_allclose(x, xr, rtol=0, atol=1e-9 * scale, err_msg=name)
    # every family reached the rank it was built for
    assert seen["plane"] == {3} and seen["near_collinear"] == {3} and seen["zero_column"] == {2}
    assert seen["axis_collinear"] == {2}, seen
    assert seen["duplicates"] <= {1, 2} and 1 in seen["duplicates"], seen
    assert 3 in seen["through_origin"]


def test_constructed_scene_reaches_rank_deficient_plane_fits():
    """The constructed LaserMapping scene of util.rank_deficient_plane_scene (the input of the -m gpu test of the same name): 25 of its 75 isolated five-point
    'rails' are exactly collinear along an axis, their queries' plane fits have nonzeroPivots() = 2 — and the oracle, following Eigen's column-pivoted QR, returns a
    finite unit normal for every one of them and accepts the plane (the five points lie on it), where the unpivoted QR of rounds 1 - 5 divided by a ~1e-17 pivot."""
    from util import rank_deficient_plane_scene
    mods, frames = rank_deficient_plane_scene()
    p = synth.default_params(16, 1800)
    for k, v in mods.items():
        setattr(p, k, v)
    o = O.Oracle(p)
    hist = []
    for c, s, ol, od in frames:
        o.lm_process(c, s, ol, od)
        hist.append(o.get("lm_plane_rank_hist").copy())
        b = o.get("lm_blocks14").reshape(-1, 14)
        assert np.isfinite(b).all()
        planes = b[b[:, 0] == 3]
        if len(planes):
            np.testing.assert_allclose(np.linalg.norm(planes[:, 4:7], axis=1), 1.0, rtol=0, atol=1e-12)
    assert hist[0].sum() == 0                      # frame 0: empty map, becomes key frame 0
    assert hist[1][2] == 25 and hist[2][2] == 25   # the axis-aligned rails: rank 2
    assert hist[1][3] > 2500 and hist[1][0] == hist[1][1] == 0
    info = o.get("lm_info")
    assert info[1] == 1 and info[4] > 2500         # the last frame optimised against > 2500 planes


def test_eig3_against_numpy_eigh():
    """The oracle's eig3 (cyclic Jacobi with a relative stopping test, standing in for Eigen::SelfAdjointEigenSolver<Matrix3d>, laserMapping.cpp:394)
    against numpy.linalg.eigh: scatter matrices of five near-collinear points (what lm_fit feeds it), generic ones, near-degenerate and zero ones.
    Eigenvalues ascending to 1e-13 of the trace, V orthonormal, A V = V diag(lam), the dominant eigenvector aligned where it is separated."""
    rng = np.random.default_rng(5)
    L = O.lib()
    mats = []
    for k in range(4000):
        if k % 4 == 0:      # five points along a line with small noise
            d = rng.standard_normal(3); d /= np.linalg.norm(d)
            pts = np.outer(np.linspace(-0.4, 0.4, 5) + rng.normal(0, 0.02, 5), d) + rng.normal(0, 10.0 ** rng.uniform(-6, -1), (5, 3)) + rng.normal(0, 30, 3)
        elif k % 4 == 1:    # generic
            pts = rng.standard_normal((5, 3)) * 10.0 ** rng.uniform(-3, 2)
        elif k % 4 == 2:    # two nearly equal eigenvalues
            pts = rng.standard_normal((5, 3)) * np.array([1.0, 1.0 + 10.0 ** rng.uniform(-12, -3), 0.1])
        else:               # rank one / zero
            pts = np.outer(rng.standard_normal(5), rng.standard_normal(3)) if k % 8 == 3 else np.zeros((5, 3))
        z = pts - pts.mean(axis=0)
        mats.append(z.T @ z)
    for A in mats:
        A = np.ascontiguousarray(0.5 * (A + A.T))
        lam, V = np.empty(3), np.empty(9)
        L.oracle_eig3(A.ctypes.data, lam.ctypes.data, V.ctypes.data)
        V = V.reshape(3, 3)
        w, U = np.linalg.eigh(A)
        tr = max(np.trace(A), 1e-300)
        assert np.all(np.diff(lam) >= 0)
        np.testing.assert_allclose(lam, w, rtol=0, atol=1e-13 * tr)
        np.testing.assert_allclose(V.T @ V, np.eye(3), rtol=0, atol=1e-14)
        np.testing.assert_allclose(A @ V, V * lam, rtol=0, atol=1e-13 * tr)
        if w[2] - w[1] > 1e-6 * tr:
            assert abs(V[:, 2] @ U[:, 2]) > 1 - 1e-12


def test_cost_functors_against_finite_differences():
    """Analytic Jacobians agree with central differences except where the reference deliberately does not:
    the LO functors zero most columns and mis-scale dz by 1/k (utility.h:226-231), the LM functors carry the
    dy_dp typo in the pitch column (utility.h:153)."""
    rng = np.random.default_rng(0)
    p = np.array([0.1, -0.05, 0.02, 0.01, -0.02, 0.03])
    for btype in range(4):
        g = np.zeros(13)
        g[0:3] = rng.normal(size=3) * 5
        g[3:6] = g[0:3] + rng.normal(size=3) * 0.3
        g[6:9] = g[3:6] + rng.normal(size=3)
        g[9:12] = g[3:6] + rng.normal(size=3)
        if btype == 3:
            n = rng.normal(size=3)
            g[3:6], g[12] = n / np.linalg.norm(n), 0.3
        r, J = O.eval_block(btype, g, p)
        Jn = np.zeros(6)
        for k in range(6):
            e = np.zeros(6)
            e[k] = 1e-6
            Jn[k] = (O.eval_block(btype, g, p + e)[0] - O.eval_block(btype, g, p - e)[0]) / 2e-6
        if btype == 0:
            assert np.all(J[[0, 1, 3, 4, 5]] == 0)
        elif btype == 1:
            assert np.all(J[[2, 3, 4]] == 0) and np.allclose(J[[0, 1, 5]], Jn[[0, 1, 5]], atol=1e-5)
        else:
            assert np.allclose(J[[0, 1, 2, 3, 5]], Jn[[0, 1, 2, 3, 5]], atol=1e-5)
            assert abs(J[4] - Jn[4]) > 1e-4  # the typo is reproduced


def test_solver_recovers_a_known_pose():
    """The Ceres restatement converges on exact point-to-plane / point-to-line data."""
    rng = np.random.default_rng(2)
    true = np.array([0.3, -0.2, 0.1, 0.0, 0.0, 0.04])  # roll/pitch 0: the pitch column typo is harmless there

    def rot(p):
        cy, sy = np.cos(p[5]), np.sin(p[5])
        return np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])

    blocks = []
    for i in range(300):
        cp = rng.normal(size=3) * 8
        lp = rot(true) @ cp + true[:3]
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        if i % 3:
            blocks.append([3, *cp, *n, 0, 0, 0, 0, 0, 0, -float(n @ lp)])
        else:
            blocks.append([2, *cp, *(lp + 0.1 * n), *(lp - 0.1 * n), 0, 0, 0, 0])
    x, info = O.solve(np.array(blocks), np.zeros(6), 20)
    assert info["final_cost"] < 1e-12 * max(info["initial_cost"], 1)
    assert np.allclose(x[[0, 1, 2, 5]], true[[0, 1, 2, 5]], atol=1e-5)
    assert info["successful"] >= 3


def test_full_loop_tracks_ground_truth(params_a):
    """Known-answer check on the synthetic trajectory: the restated pipeline follows T0 (0.1 m/scan)."""
    p = params_a
    o = O.Oracle(p)
    gt0 = synth.pose(0)
    for k in range(40):
        o.process_scan(synth.scan(p, k))
    mp = o.get("map_pose")
    gt = synth.pose(39)
    assert np.linalg.norm(mp[:2] - (gt[:2] - gt0[:2])) < 0.5
    assert o.get("lm_info")[11] >= 3  # key frames were added


def test_reference_option_variants(params_a):
    """nodelet vs standalone twins only differ where the options say so."""
    pts = synth.scan(params_a, 4)
    a = O.Oracle(params_a)
    b_p = params_a.copy()
    b_p.sector_formula = 1
    b = O.Oracle(b_p)
    a.ip(pts), a.fe(), b.ip(pts), b.fe()
    assert_bit_equal(a.get("sharp_idx"), b.get("sharp_idx"), "sector formulas agree when end >= start")
    c_p = params_a.copy()
    c_p.sort_mode = 1  # libstdc++ std::sort tie order
    c = O.Oracle(c_p)
    c.ip(pts), c.fe()
    same = np.array_equal(a.get("less_sharp_idx"), c.get("less_sharp_idx"))
    print("std::sort vs stable tie order gives identical picks on this scan:", same)


def test_std_sort_phases_equal_std_sort():
    """oracle_std_sort_order(depth_limit >= 0) calls libstdc++'s __introsort_loop + __final_insertion_sort directly (to force the
    heap-sort branch in the device's tie-order test); with std::sort's own depth limit 2 floor(log2 n) it must be std::sort."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 17, 100, 300, 1000, 4096):
        for hi in (4, 1 << 30):
            keys = rng.integers(0, hi, n).astype(np.uint32)
            a = O.std_sort_order(keys)
            b = O.std_sort_order(keys, 2 * (int(n).bit_length() - 1))
            assert_bit_equal(a, b, f"n={n} hi={hi}")
            assert np.array_equal(keys[a], np.sort(keys)), "sorted"
            assert np.array_equal(np.sort(a), np.arange(n)), "permutation"


def test_motion_deskew_restatement(params_a):
    """adjustDistortion (laserOdometry.cpp:557-726, dead in the reference: deskew_mode = 1 switches it on).  A platform at rest
    leaves the cloud untouched bit for bit; a pure yaw rate turns every point by the yaw the IMU integrated between the first
    point's time and the point's own (`col * scan_period / Horizon_SCAN`), checked against a double-precision formula."""
    from util import imu_stream
    p = params_a.copy()
    p.deskew_mode = 1
    pts = synth.scan(p, 3)
    o = O.Oracle(p)
    o.push_imu(imu_stream(-0.1, 0.5, yaw_rate=0.0, acc=(0, 0, 0), tilt=(0, 0)))
    o.set_scan_time(0.0)
    o.ip(pts), o.fe()
    assert_bit_equal(o.get("undistorted"), o.get("seg_cloud"), "platform at rest")
    w = 0.5
    o = O.Oracle(p)
    o.push_imu(imu_stream(-0.1, 0.5, yaw_rate=w, acc=(0, 0, 0), tilt=(0, 0)))
    o.set_scan_time(0.0)
    o.ip(pts), o.fe()
    seg, und, col = o.get("seg_cloud").reshape(-1, 4), o.get("undistorted").reshape(-1, 4), o.get("seg_col")
    t = col * p.scan_period / p.horizon_scan
    t0 = t[0]
    a = w * (t - t0)                     # yaw gained since the first point; R_start^-1 R_cur = Rz(a)
    want = np.stack([np.cos(a) * seg[:, 0] - np.sin(a) * seg[:, 1], np.sin(a) * seg[:, 0] + np.cos(a) * seg[:, 1], seg[:, 2]], 1)
    err = np.abs(und[:, :3] - want).max()
    assert err < 2e-3, err                 # the IMU's yaw is linear between 100 Hz samples; f32 rotation of points up to 50 m away
    assert np.abs(und[:, :3] - seg[:, :3]).max() > 0.05, "the cloud did move"
    assert_bit_equal(und[0], seg[0], "the first point defines the start pose and is not touched (:641-647)")


def test_quick_projection_polynomial_bound(params_a):
    """The error budget of ip_fused's quick projection (csrc/ip_common.h: ip_point_quick): the f32 evaluation of its odd degree-11 polynomial is
    within 1.8e-6 rad of atan on [0, 1], and the whole angle estimate — reduction to [0, 1], quadrant fix-ups, f32 roundings; 1 ulp is charged for
    the hardware's rcp / sqrt — stays inside the 4e-6 rad the margins are built from (twice over).  With that budget a point whose estimated
    column / row lies outside the margins lands in the reference's cell: checked here against the oracle's projection on random directions."""
    c = [np.float32(v) for v in (0.99997726, -0.33262347, 0.19354346, -0.11643287, 0.05265332, -0.01172120)]
    def poly(u):
        u2 = u * u
        return u * (c[0] + u2 * (c[1] + u2 * (c[2] + u2 * (c[3] + u2 * (c[4] + u2 * c[5])))))
    u = np.linspace(0.0, 1.0, 2_000_001).astype(np.float32)
    assert np.abs(poly(u).astype(np.float64) - np.arctan(u.astype(np.float64))).max() < 1.8e-6
    rng = np.random.default_rng(5)
    n = 1_000_000
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    pts = (v * rng.uniform(1.0, 100.0, n)[:, None]).astype(np.float32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    one = np.float32(1.0)
    ax, ay = np.abs(x), np.abs(y)
    mn, mx = np.minimum(ax, ay), np.maximum(ax, ay)
    b = poly(mn * (one / mx))
    b = np.where(ay > ax, np.float32(1.57079633) - b, b)
    b = np.where(x < 0, np.float32(3.14159265) - b, b)
    b = np.where(y < 0, -b, b)
    az = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(b.astype(np.float64) - az).max() < 3.0e-6            # azimuth estimate (before the 1 ulp of rcp: 1.2e-7 relative)
    hf = np.sqrt(x * x + y * y)
    ok = np.abs(z) < np.float32(0.6) * hf
    t = (z * (one / hf))[ok]
    el = np.arctan2(z.astype(np.float64), np.hypot(x.astype(np.float64), y.astype(np.float64)))[ok]
    assert np.abs(poly(t).astype(np.float64) - el).max() < 2.5e-6      # elevation estimate where the quick path is taken (|t| < 0.6)
    # end to end against the oracle's cells: quick decision = floor of the estimated (row + 0.5, column) outside the margins
    p = params_a
    H, NS = p.horizon_scan, p.n_scan
    mr = max(0.005, 8e-6 * 57.29577951 / p.ang_res_y + 1e-5); mc = max(0.02, 8e-6 * 57.29577951 / p.ang_res_x + 8e-7 * H)
    r0 = (poly(z * (one / hf)) * np.float32(57.29577951) + np.float32(p.ang_bottom)) * np.float32(1.0 / p.ang_res_y) + np.float32(0.5)
    c0 = (np.float32(6.28318531) - b) * (np.float32(57.29577951) * np.float32(1.0 / p.ang_res_x))
    fr, fc = r0 - np.floor(r0), c0 - np.floor(c0)
    decided = ok & (fr >= mr) & (fr <= 1 - mr) & (fc >= mc) & (fc <= 1 - mc)
    rfl = np.nan_to_num(np.floor(r0), nan=-100.0, posinf=1e6, neginf=-1e6).astype(np.int64)
    row = np.where(rfl >= 0, rfl, np.where(rfl == -1, 0, -1))
    col = np.nan_to_num(np.floor(c0), nan=-100.0, posinf=1e6, neginf=-1e6).astype(np.int64); col = np.where(col >= H, col - H, col)
    quick = np.where((row >= 0) & (row < NS) & (col >= 0) & (col < H), col + row * H, -1)
    # the reference expressions (imageProjection.cpp:79-97) in double on the f32-rounded libm results, as the oracle evaluates them
    va = np.arctan2(z, np.sqrt(x.astype(np.float64) ** 2 + y.astype(np.float64) ** 2).astype(np.float32)).astype(np.float32).astype(np.float64) * 180.0
    rr = ((va / np.pi + p.ang_bottom) / p.ang_res_y + 0.5); rrow = np.trunc(rr).astype(np.int64)
    ha = (-(np.arctan2(y, x).astype(np.float32)).astype(np.float64) + 2 * np.pi) * 180.0
    rcol = np.trunc((ha / np.pi) / p.ang_res_x).astype(np.int64); rcol = np.where(rcol >= H, rcol - H, rcol)
    ref = np.where((rrow >= 0) & (rrow < NS) & (rr > -1) & (rcol >= 0) & (rcol < H), rcol + rrow * H, -1)
    assert decided.mean() > 0.3
    assert np.array_equal(quick[decided], ref[decided])


def test_solver_agrees_with_scipy_least_squares_on_noisy_data_with_outliers():
    """An independent trust-region solver (scipy.optimize.least_squares, method 'trf', loss 'huber' with f_scale = the Huber delta: the same
    robust cost as ceres::HuberLoss, 0.5 * sum rho(r^2)) on the SAME residual / Jacobian functions (the oracle's cost functors, typos included)
    must end where the oracle's restatement of ceres::Solve ends, up to Ceres' default function tolerance: noisy point-to-plane / point-to-line data with 10 %
    gross outliers, so that the loss function and its corrector matter."""
    from scipy.optimize import least_squares
    rng = np.random.default_rng(12)
    true = np.array([0.25, -0.15, 0.08, 0.01, -0.015, 0.035])

    def rot(p):
        cr, sr, cp, sp, cy, sy = np.cos(p[3]), np.sin(p[3]), np.cos(p[4]), np.sin(p[4]), np.cos(p[5]), np.sin(p[5])
        rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]); ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]]); rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        return rz @ ry @ rx

    blocks = []
    for i in range(400):
        cp = rng.normal(size=3) * 8
        lp = rot(true) @ cp + true[:3] + rng.normal(size=3) * 0.02
        if i % 10 == 0:
            lp += rng.normal(size=3) * 1.5          # gross outlier: far beyond the Huber delta of 0.1
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        if i % 3:
            blocks.append([3, *cp, *n, 0, 0, 0, 0, 0, 0, -float(n @ lp)])
        else:
            blocks.append([2, *cp, *(lp + 0.1 * n), *(lp - 0.1 * n), 0, 0, 0, 0])
    blocks = np.array(blocks)
    huber = 0.1

    def geom(b):
        g = np.zeros(13); g[0:12] = b[1:13]; g[12] = b[13]
        return g

    def res(x):
        return np.array([O.eval_block(int(b[0]), geom(b), x)[0] for b in blocks])

    def jac(x):
        return np.array([O.eval_block(int(b[0]), geom(b), x)[1] for b in blocks])

    x0 = np.zeros(6)
    x, info = O.solve(blocks, x0, 50, huber)
    for _ in range(3):                              # (ceres::Solve is restarted from its own result, as scan2MapOptimization does twice)
        x, info = O.solve(blocks, x, 50, huber)
    ref = least_squares(res, x0, jac=jac, method="trf", loss="huber", f_scale=huber, xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=500)
    assert ref.success
    # ceres::Solve stops at its default function_tolerance (relative cost change <= 1e-6), scipy was asked for 1e-15: the two end points
    # differ by what that tolerance leaves (measured 7e-5 in the pose, 1e-7 relative in the cost), not by more
    assert np.abs(x - ref.x).max() < 3e-4, (x, ref.x)
    assert abs(info["final_cost"] - ref.cost) < 5e-6 * ref.cost, (info["final_cost"], ref.cost)
    assert np.abs(x - true)[[0, 1, 2, 5]].max() < 0.02   # and both are near the truth despite the outliers
