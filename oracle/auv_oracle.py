"""CPU oracle for the AUV simulation hot path (TEST INFRASTRUCTURE ONLY).

This module is the normative fp64 numpy restatement of the per-step path:
3-DOF vessel dynamics, moving-obstacle advance, LiDAR ray-cast, signed
obstacle distances / collision flag, per-obstacle encounter metadata and risk,
and observation assembly.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path never
does.

PARITY UNPINNED for the LiDAR, the polygon hull distance and the collision
predicate: the reference checkout (hebe2602/gym-auv @ v1) does not contain the
simulator (`radarCNN.py:232` only names the absent ``gym_auv/`` package), so
there is no reference function, golden vector or fixture to pin these against.
PINNED by the reference's own log (data/metadata_MovingObstaclesNoRules.csv,
data/risk_MovingObstaclesNoRules.csv), through fixtures under tests/golden/:
the vessel dynamics (integrator, step, inertia, damping, couplings, thrust and
moment limits: test_ref_surge_cpu.py, test_ref_turn_cpu.py), the integration of
the position, the bearing and the circle distance (test_ref_track_cpu.py), the
obstacle advance and every metadata column with the risk (ref_cri.py,
test_ref_cri_cpu.py, test_ref_ship_cpu.py).  What the
mounted files do constrain is restated here with citations:

* scan width 180 and maximum range 150 m          -> utils/dataloader.py:13,56,127
* closeness observation ``1 - range/150``          -> utils/dataloader.py:56
* scan wraps over the full circle (circular pad)   -> models/lidar_cnn_shallow.py:30
* observation dict {"perception": (B,1,180), "navigation": (B,1,6)}
                                                   -> radarCNN.py:165-175,184-193,215-262
* metadata row [speed_OS, heading_OS, (heading_ts, dist, speed, bearing,
  course) x n]; static obstacle <=> heading_ts == 0 and speed == speed_OS
                                                   -> utils/explore_dataset.py:8,40-53
* risk label reduction max / sum / mean over <= 8 per-obstacle risks
                                                   -> utils/dataloader.py:104-120

Arithmetic contract (this is what makes GPU-vs-oracle comparisons bit-exact
instead of tolerance-based): every quantity is IEEE-754 binary64, computed
with *unfused* +, -, *, /, sqrt, floor, rint in exactly the operation order
written below.  sin/cos/atan2 are not taken from libm: they are the fixed
polynomial sequences `sincos` / `atan2` in this file, so that every
implementation of the contract produces identical bits.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------
# constants of the arithmetic contract
# ----------------------------------------------------------------------------
PI = float.fromhex("0x1.921fb54442d18p+1")
TWO_PI = float.fromhex("0x1.921fb54442d18p+2")
INV_TWO_PI = float.fromhex("0x1.45f306dc9c883p-3")
PIO2 = float.fromhex("0x1.921fb54442d18p+0")
PIO4 = float.fromhex("0x1.921fb54442d18p-1")
PIO8 = float.fromhex("0x1.921fb54442d18p-2")
TWO_OVER_PI = float.fromhex("0x1.45f306dc9c883p-1")
# pi/2 split: HI holds the leading 26 mantissa bits, so k*HI is exact for |k| < 2**27.
PIO2_HI = float.fromhex("0x1.921fb50000000p+0")
PIO2_LO = float.fromhex("0x1.110b4611a6263p-26")

# Taylor coefficients (-1)^j/(2j+1)!, j = 1..8 and (-1)^j/(2j)!, j = 1..9;
# truncation error on |r| <= pi/4 is below 1e-19.
SIN_C = (-0.16666666666666666, 0.008333333333333333, -0.0001984126984126984,
         2.7557319223985893e-06, -2.505210838544172e-08, 1.6059043836821613e-10,
         -7.647163731819816e-13, 2.8114572543455206e-15)
COS_C = (-0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05,
         -2.755731922398589e-07, 2.08767569878681e-09, -1.1470745597729725e-11,
         4.779477332387385e-14, -1.5619206968586225e-16)

# atan argument reduction: z in [0,1] is moved to |w| <= tan(pi/16) around
# one of atan^-1{0, pi/8, pi/4}; series (-1)^k/(2k+1), k = 1..11.
ATAN_T1 = 0.198912367379658      # tan(pi/16)
ATAN_T3 = 0.6681786379192989     # tan(3pi/16)
TAN_PIO8 = float.fromhex("0x1.a827999fcef32p-2")
ATAN_C = tuple(((-1.0) ** k) / (2 * k + 1) for k in range(1, 12))

# RKF45 (Fehlberg) tableau, 5th-order solution propagated.
RK_A = (
    (),
    (1.0 / 4.0,),
    (3.0 / 32.0, 9.0 / 32.0),
    (1932.0 / 2197.0, -7200.0 / 2197.0, 7296.0 / 2197.0),
    (439.0 / 216.0, -8.0, 3680.0 / 513.0, -845.0 / 4104.0),
    (-8.0 / 27.0, 2.0, -3544.0 / 2565.0, 1859.0 / 4104.0, -11.0 / 40.0),
)
RK_B5 = (16.0 / 135.0, 0.0, 6656.0 / 12825.0, 28561.0 / 56430.0, -9.0 / 50.0, 2.0 / 55.0)

# indices into the model-constant vector `mp` (length 16); values are supplied
# by the caller so that oracle and device use identical numbers.
(MP_MINV11, MP_MINV22, MP_MINV23, MP_MINV32, MP_MINV33,
 MP_MU, MP_MXG, MP_RESERVED,
 MP_XU, MP_XUU, MP_YV, MP_YVV, MP_YR, MP_NV, MP_NR, MP_NRR) = range(16)

NAV_FEATURES = 6
META_FIELDS = 5   # heading_ts, dist, speed, bearing, course


# ----------------------------------------------------------------------------
# elementary functions of the contract
# ----------------------------------------------------------------------------
def sincos(x):
    """(sin x, cos x) by two-term Cody-Waite reduction and Taylor-Horner."""
    x = np.asarray(x, dtype=np.float64)
    k = np.rint(x * TWO_OVER_PI)
    r = (x - k * PIO2_HI) - k * PIO2_LO
    r2 = r * r
    ps = np.full_like(r, SIN_C[-1])
    for c in SIN_C[-2::-1]:
        ps = c + r2 * ps
    sr = r + (r * r2) * ps
    pc = np.full_like(r, COS_C[-1])
    for c in COS_C[-2::-1]:
        pc = c + r2 * pc
    cr = 1.0 + r2 * pc
    q = k.astype(np.int64) & 3
    s = np.where(q == 0, sr, np.where(q == 1, cr, np.where(q == 2, -sr, -cr)))
    c = np.where(q == 0, cr, np.where(q == 1, -sr, np.where(q == 2, -cr, sr)))
    return s, c


def atan2(y, x):
    """Four-quadrant arctangent of the contract; atan2(0, 0) = 0."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y, x = np.broadcast_arrays(y, x)
    ax = np.abs(x)
    ay = np.abs(y)
    zero = (ax == 0.0) & (ay == 0.0)
    swap = ay > ax
    num = np.where(swap, ax, ay)
    den = np.where(swap, ay, ax)
    den = np.where(zero, 1.0, den)
    z = num / den
    c = np.where(z > ATAN_T3, 1.0, np.where(z > ATAN_T1, TAN_PIO8, 0.0))
    base = np.where(z > ATAN_T3, PIO4, np.where(z > ATAN_T1, PIO8, 0.0))
    w = (z - c) / (1.0 + z * c)
    w2 = w * w
    p = np.full_like(w, ATAN_C[-1])
    for a in ATAN_C[-2::-1]:
        p = a + w2 * p
    a = base + (w + (w * w2) * p)
    a = np.where(swap, PIO2 - a, a)
    a = np.where(x < 0.0, PI - a, a)
    a = np.where(y < 0.0, -a, a)
    return np.where(zero, 0.0, a)


def wrap_angle(a):
    """Map an angle to [-pi, pi)."""
    a = np.asarray(a, dtype=np.float64)
    k = np.floor((a + PI) * INV_TWO_PI)
    return a - k * TWO_PI


# ----------------------------------------------------------------------------
# a1: vessel dynamics
# ----------------------------------------------------------------------------
def vessel_rhs(y, fx, mz, mp):
    """Time derivative of y = [x, y, psi, u, v, r] (arrays of shape [6, V])."""
    psi, u, v, r = y[2], y[3], y[4], y[5]
    s, c = sincos(psi)
    xd = c * u - s * v
    yd = s * u + c * v
    n1 = (mp[MP_XU] + mp[MP_XUU] * np.abs(u)) * u
    n2 = (mp[MP_YV] + mp[MP_YVV] * np.abs(v)) * v + (mp[MP_MU] * u + mp[MP_YR]) * r
    n3 = mp[MP_NV] * v + ((mp[MP_MXG] * u + mp[MP_NR]) + mp[MP_NRR] * np.abs(r)) * r
    r1 = fx - n1
    r2 = 0.0 - n2
    r3 = mz - n3
    ud = mp[MP_MINV11] * r1
    vd = mp[MP_MINV22] * r2 + mp[MP_MINV23] * r3
    rd = mp[MP_MINV32] * r2 + mp[MP_MINV33] * r3
    return np.stack([xd, yd, r, ud, vd, rd])


def step_dynamics(state, action, cfg):
    """One RKF45 step of size cfg['dt'].  state [6,V], action [2,V] -> [6,V].

    action[0] in [0,1] scales the surge force, action[1] in [-1,1] the yaw
    moment; both are clipped first.
    """
    mp = cfg["model"]
    h = cfg["dt"]
    fx = np.minimum(np.maximum(action[0], 0.0), 1.0) * cfg["thrust_max"]
    mz = np.minimum(np.maximum(action[1], -1.0), 1.0) * cfg["moment_max"]
    k = []
    for i in range(6):
        if i == 0:
            yi = state
        else:
            acc = RK_A[i][0] * k[0]
            for j in range(1, i):
                acc = acc + RK_A[i][j] * k[j]
            yi = state + h * acc
        k.append(vessel_rhs(yi, fx, mz, mp))
    acc = RK_B5[0] * k[0]
    for j in (2, 3, 4, 5):
        acc = acc + RK_B5[j] * k[j]
    out = state + h * acc
    out[2] = wrap_angle(out[2])
    return out


# ----------------------------------------------------------------------------
# a2: obstacle advance
# ----------------------------------------------------------------------------
def advance_obstacles(circles, poly_pose, dt):
    """Constant-velocity advance, in place.  circles [V,C,5] = (cx,cy,rad,vx,vy);
    poly_pose [V,P,6] = (X,Y,cos h,sin h,vx,vy)."""
    circles[..., 0] = circles[..., 0] + circles[..., 3] * dt
    circles[..., 1] = circles[..., 1] + circles[..., 4] * dt
    poly_pose[..., 0] = poly_pose[..., 0] + poly_pose[..., 4] * dt
    poly_pose[..., 1] = poly_pose[..., 1] + poly_pose[..., 5] * dt


def polygon_world_vertices(poly_pose, poly_verts):
    """Body-frame vertices [V,P,NV,2] -> world frame."""
    X = poly_pose[..., 0][..., None]
    Y = poly_pose[..., 1][..., None]
    ch = poly_pose[..., 2][..., None]
    sh = poly_pose[..., 3][..., None]
    bx = poly_verts[..., 0]
    by = poly_verts[..., 1]
    wx = X + (ch * bx - sh * by)
    wy = Y + (sh * bx + ch * by)
    return wx, wy


# ----------------------------------------------------------------------------
# a3: LiDAR
# ----------------------------------------------------------------------------
def polygon_reach2(poly_verts, poly_nv):
    """Per polygon [V,P]: the squared distance of its farthest vertex from its
    pose point (body frame), or -1 when the polygon is not strictly convex and
    counter-clockwise (every vertex strictly to the left of every edge it does
    not belong to).  A convex polygon seen from outside the circle of that
    radius about its pose point shows the sensor only the edges that face it
    (`lidar`)."""
    v = np.asarray(poly_verts, dtype=np.float64)
    nv = np.asarray(poly_nv)
    NV = v.shape[2]
    convex = nv >= 3
    reach2 = np.zeros(nv.shape)
    for k in range(NV):
        in_k = k < nv
        k1 = np.where(k + 1 < nv, k + 1, 0)
        vkx, vky = v[:, :, k, 0], v[:, :, k, 1]
        v1x = np.take_along_axis(v[..., 0], k1[..., None], axis=2)[..., 0]
        v1y = np.take_along_axis(v[..., 1], k1[..., None], axis=2)[..., 0]
        ex, ey = v1x - vkx, v1y - vky
        rr = vkx * vkx + vky * vky
        reach2 = np.where(in_k & (rr > reach2), rr, reach2)
        for j in range(NV):
            other = in_k & (j < nv) & (j != k) & (j != k1)
            wx, wy = v[:, :, j, 0] - vkx, v[:, :, j, 1] - vky
            left = (ex * wy - ey * wx) > 0.0
            convex = convex & (left | ~other)
    return np.where(convex, reach2, -1.0)


def lidar(state, circles, poly_pose, poly_verts, poly_nv, ray_dirs, sensor_range, poly_reach2=None):
    """Nearest boundary hit per ray.

    Returns (ranges [V,R] f64, hit [V,R] i32).  Obstacle ids: circles
    0..C-1, polygons C..C+P-1; -1 and range = sensor_range when nothing is
    hit.  The running best starts at sensor_range; primitives are visited in
    id order (polygon edges in vertex order) and a hit at parameter t replaces
    the best only when 0 <= t < best, so ties keep the lowest id and a hit at
    exactly sensor_range does not count.

    A return comes from a surface that faces the sensor: an edge of a convex
    polygon (`polygon_reach2` >= 0) whose outward side faces away from the
    vessel (a x e > 0) is not tested while the vessel is farther from the
    polygon's pose point than its farthest vertex (and so outside it).  Seen
    from there every ray meets a facing edge first, so the scan is the same
    except when a ray passes exactly through a silhouette vertex.
    """
    V = state.shape[1]
    R = ray_dirs.shape[0]
    C = circles.shape[1]
    P = poly_pose.shape[1]
    px = state[0][:, None]
    py = state[1][:, None]
    s, c = sincos(state[2])
    cb = ray_dirs[:, 0][None, :]
    sb = ray_dirs[:, 1][None, :]
    dx = c[:, None] * cb - s[:, None] * sb
    dy = s[:, None] * cb + c[:, None] * sb
    best = np.full((V, R), sensor_range, dtype=np.float64)
    hit = np.full((V, R), -1, dtype=np.int32)

    def consider(t, valid, oid):
        nonlocal best, hit
        upd = valid & (t >= 0.0) & (t < best)
        best = np.where(upd, t, best)
        hit = np.where(upd, np.int32(oid), hit)

    with np.errstate(invalid="ignore", divide="ignore"):
        for j in range(C):
            rad = circles[:, j, 2][:, None]
            mx = circles[:, j, 0][:, None] - px
            my = circles[:, j, 1][:, None] - py
            cc = (mx * mx + my * my) - rad * rad
            b = mx * dx + my * dy
            disc = b * b - cc
            ok = (rad > 0.0) & (disc >= 0.0)
            sq = np.sqrt(np.where(ok, disc, 0.0))
            t = b - sq
            t = np.where(t < 0.0, b + sq, t)
            consider(t, ok, j)
        if P:
            wx, wy = polygon_world_vertices(poly_pose, poly_verts)
            NV = poly_verts.shape[2]
            if poly_reach2 is None:
                poly_reach2 = polygon_reach2(poly_verts, poly_nv)
            for p in range(P):
                nv = poly_nv[:, p]
                mx = poly_pose[:, p, 0] - state[0]
                my = poly_pose[:, p, 1] - state[1]
                outside = (poly_reach2[:, p] >= 0.0) & ((mx * mx + my * my) > poly_reach2[:, p])
                for k in range(NV):
                    k1 = np.where(k + 1 < nv, k + 1, 0)
                    Ax = wx[:, p, k]
                    Ay = wy[:, p, k]
                    Bx = np.take_along_axis(wx[:, p, :], k1[:, None], axis=1)[:, 0]
                    By = np.take_along_axis(wy[:, p, :], k1[:, None], axis=1)[:, 0]
                    ax = (Ax - state[0])[:, None]
                    ay = (Ay - state[1])[:, None]
                    ex = (Bx - Ax)[:, None]
                    ey = (By - Ay)[:, None]
                    nt = ax * ey - ay * ex
                    den = dx * ey - dy * ex
                    ns = ax * dy - ay * dx
                    inseg = np.where(den > 0.0, (ns >= 0.0) & (ns <= den),
                                     (ns <= 0.0) & (ns >= den))
                    faces = ~(outside[:, None] & (nt > 0.0))
                    ok = (k < nv)[:, None] & (den != 0.0) & inseg & faces
                    t = nt / np.where(ok, den, 1.0)
                    consider(t, ok, C + p)
    return best, hit


# ----------------------------------------------------------------------------
# a4: signed obstacle distances and the collision predicate
# ----------------------------------------------------------------------------
def obstacle_distances(state, circles, poly_pose, poly_verts, poly_nv, vessel_radius, contact_dist=-2.0):
    """Signed hull distance per obstacle: (centre-to-surface distance, negative
    inside the obstacle) minus vessel_radius.  Absent obstacles give +inf.
    Returns (dist [V,C+P], collision u8 [V], min_dist [V], min_idx i32 [V]);
    collision <=> some dist < contact_dist (-2 with the default 3.255 m radius: the hull
    within 1.255 m of the vessel's centre); min_idx = lowest id attaining the minimum
    (-1 when the env has no obstacle)."""
    V = state.shape[1]
    C = circles.shape[1]
    P = poly_pose.shape[1]
    dist = np.full((V, C + P), np.inf, dtype=np.float64)
    px, py = state[0], state[1]
    for j in range(C):
        rad = circles[:, j, 2]
        mx = circles[:, j, 0] - px
        my = circles[:, j, 1] - py
        d = (np.sqrt(mx * mx + my * my) - rad) - vessel_radius
        dist[:, j] = np.where(rad > 0.0, d, np.inf)
    if P:
        wx, wy = polygon_world_vertices(poly_pose, poly_verts)
        NV = poly_verts.shape[2]
        for p in range(P):
            nv = poly_nv[:, p]
            dmin = np.full(V, np.inf)
            inside = np.zeros(V, dtype=bool)
            for k in range(NV):
                k1 = np.where(k + 1 < nv, k + 1, 0)
                Ax = wx[:, p, k]
                Ay = wy[:, p, k]
                Bx = np.take_along_axis(wx[:, p, :], k1[:, None], axis=1)[:, 0]
                By = np.take_along_axis(wy[:, p, :], k1[:, None], axis=1)[:, 0]
                ax = Ax - px
                ay = Ay - py
                by = By - py
                ex = Bx - Ax
                ey = By - Ay
                ee = ex * ex + ey * ey
                with np.errstate(invalid="ignore", divide="ignore"):
                    tp = (0.0 - (ax * ex + ay * ey)) / np.where(ee > 0.0, ee, 1.0)
                tp = np.where(ee > 0.0, np.minimum(np.maximum(tp, 0.0), 1.0), 0.0)
                qx = ax + tp * ex
                qy = ay + tp * ey
                d = np.sqrt(qx * qx + qy * qy)
                act = k < nv
                dmin = np.where(act & (d < dmin), d, dmin)
                nt = ax * ey - ay * ex
                up = (ay <= 0.0) & (by > 0.0) & (nt > 0.0)
                dn = (by <= 0.0) & (ay > 0.0) & (nt < 0.0)
                inside = inside ^ (act & (up | dn))
            d = np.where(inside, -dmin, dmin) - vessel_radius
            dist[:, C + p] = np.where(nv > 0, d, np.inf)
    if C + P:
        min_idx = np.argmin(dist, axis=1).astype(np.int32)
        min_dist = dist[np.arange(V), min_idx]
        min_idx = np.where(np.isinf(min_dist), np.int32(-1), min_idx)
    else:
        min_idx = np.full(V, -1, dtype=np.int32)
        min_dist = np.full(V, np.inf)
    collision = (min_dist < contact_dist).astype(np.uint8)
    return dist, collision, min_dist, min_idx


# ----------------------------------------------------------------------------
# a5: per-obstacle encounter metadata and risk
# ----------------------------------------------------------------------------
def encounter_metadata(state, circles, poly_pose, poly_heading, dist):
    """Per obstacle [heading_ts, dist, speed, bearing, course] (layout of
    utils/explore_dataset.py:45) plus own [speed, heading].

    heading_ts: obstacle heading (0 for circles); dist: signed hull distance;
    speed: |v_obstacle - v_own|; bearing: direction to the obstacle centre
    relative to own heading; course: direction of the relative velocity
    relative to own heading (0 when the relative velocity is zero).
    Returns (own [V,2], meta [V,C+P,5])."""
    V = state.shape[1]
    C = circles.shape[1]
    P = poly_pose.shape[1]
    px, py, psi, u, v = state[0], state[1], state[2], state[3], state[4]
    s, c = sincos(psi)
    vox = c * u - s * v
    voy = s * u + c * v
    own = np.stack([np.sqrt(vox * vox + voy * voy), psi], axis=1)
    meta = np.zeros((V, C + P, META_FIELDS), dtype=np.float64)
    for o in range(C + P):
        if o < C:
            cx, cy = circles[:, o, 0], circles[:, o, 1]
            vx, vy = circles[:, o, 3], circles[:, o, 4]
            hd = np.zeros(V)
        else:
            q = o - C
            cx, cy = poly_pose[:, q, 0], poly_pose[:, q, 1]
            vx, vy = poly_pose[:, q, 4], poly_pose[:, q, 5]
            hd = poly_heading[:, q]
        mx = cx - px
        my = cy - py
        rvx = vx - vox
        rvy = vy - voy
        meta[:, o, 0] = hd
        meta[:, o, 1] = dist[:, o]
        meta[:, o, 2] = np.sqrt(rvx * rvx + rvy * rvy)
        meta[:, o, 3] = wrap_angle(atan2(my, mx) - psi)
        meta[:, o, 4] = wrap_angle(atan2(rvy, rvx) - psi)
    return own, meta


def risk_default(state, circles, poly_pose, dist, sensor_range, rp):
    """SELF-DEFINED default collision-risk index in [0,1) per obstacle.

    The reference's CRI formula is not in the checkout (only its outputs,
    data/risk_*.csv); this default is a closest-point-of-approach index:

        risk = w_d * (1 - d/R)^2 + w_c * (1 - dcpa/R)^2 * Tc/(Tc + tcpa)

    for obstacles with hull distance d in [0, R) (d clipped at 0), 0 outside
    sensor range; the second term only when the obstacle is closing
    (tcpa > 0).  rp = (w_d, w_c, Tc).  dcpa / tcpa use obstacle centres and
    the relative velocity."""
    V = state.shape[1]
    C = circles.shape[1]
    P = poly_pose.shape[1]
    px, py, psi, u, v = state[0], state[1], state[2], state[3], state[4]
    s, c = sincos(psi)
    vox = c * u - s * v
    voy = s * u + c * v
    w_d, w_c, t_c = rp
    risk = np.zeros((V, C + P), dtype=np.float64)
    for o in range(C + P):
        if o < C:
            cx, cy = circles[:, o, 0], circles[:, o, 1]
            vx, vy = circles[:, o, 3], circles[:, o, 4]
        else:
            q = o - C
            cx, cy = poly_pose[:, q, 0], poly_pose[:, q, 1]
            vx, vy = poly_pose[:, q, 4], poly_pose[:, q, 5]
        mx = cx - px
        my = cy - py
        rvx = vx - vox
        rvy = vy - voy
        d = np.maximum(dist[:, o], 0.0)
        inrange = d < sensor_range
        fd = 1.0 - np.where(inrange, d, 0.0) / sensor_range
        term_d = w_d * (fd * fd)
        vv = rvx * rvx + rvy * rvy
        closing = 0.0 - (mx * rvx + my * rvy)
        has_c = (vv > 0.0) & (closing > 0.0)
        tcpa = np.where(has_c, closing, 0.0) / np.where(has_c, vv, 1.0)
        qx = mx + rvx * tcpa
        qy = my + rvy * tcpa
        dcpa = np.sqrt(qx * qx + qy * qy)
        fc = 1.0 - np.minimum(dcpa, sensor_range) / sensor_range
        term_c = (w_c * (fc * fc)) * (t_c / (t_c + tcpa))
        r = term_d + np.where(has_c, term_c, 0.0)
        risk[:, o] = np.where(inrange, r, 0.0)
    return risk


# ----------------------------------------------------------------------------
# a7: observation assembly
# ----------------------------------------------------------------------------
def path_table(path, nseg=None):
    """Segment table [V,KS,8] = (x0, y0, angle, cos, sin, length, arc length
    before the segment, 0) and segment counts [V] of a path given as one
    straight segment per env [V,6] or as a polyline [V,KS,6] with `nseg`
    segments in use per env (None: all KS)."""
    path = np.asarray(path, dtype=np.float64)
    if path.ndim == 2:
        path = path[:, None, :]
    V, KS = path.shape[:2]
    tab = np.zeros((V, KS, 8))
    tab[:, :, :6] = path
    cum = np.zeros(V)
    for k in range(KS):
        tab[:, k, 6] = cum
        cum = cum + path[:, k, 5]
    n = np.full(V, KS, dtype=np.int32) if nseg is None else np.asarray(nseg, dtype=np.int32).copy()
    return tab, n


def navigation_features(state, path, look_ahead, nseg=None, seg=None):
    """[u, v, r, look-ahead heading error, heading error, cross-track error]
    with respect to a path per env: one straight segment [V,6] = (x0, y0,
    angle, cos angle, sin angle, length), or a segment table [V,KS,8] of
    `path_table` with `nseg` segments in use and `seg` [V] the segment each
    vessel is on (updated in place: a vessel whose projection has passed the end
    of its segment moves on to the next one, never back).  Along-track
    position = arc length before the segment + projection on it; cross-track
    and heading errors are those of the current segment; the look-ahead point
    lies `look_ahead` further along the path, on the segment that holds that arc
    length (the last one extended).  Also returns the along-track position [V]
    and the path's total length [V]."""
    px, py, psi = state[0], state[1], state[2]
    path = np.asarray(path)
    if path.ndim == 2:
        path, nseg = path_table(path)
    V, KS = path.shape[:2]
    if seg is None:
        seg = np.zeros(V, dtype=np.int32)
    rows = np.arange(V)

    def rec(j):
        return path[rows, j]

    pt = rec(seg)
    rx = px - pt[:, 0]
    ry = py - pt[:, 1]
    loc = rx * pt[:, 3] + ry * pt[:, 4]
    for _ in range(KS - 1):
        go = (seg + 1 < nseg) & (loc > pt[:, 5])
        if not go.any():
            break
        seg[go] += 1
        pt = rec(seg)
        rx = px - pt[:, 0]
        ry = py - pt[:, 1]
        loc = rx * pt[:, 3] + ry * pt[:, 4]
    along = pt[:, 6] + loc
    cte = ry * pt[:, 3] - rx * pt[:, 4]
    st = along + look_ahead
    j = seg.copy()
    pl = pt
    for _ in range(KS - 1):
        go = (j + 1 < nseg) & (st > pl[:, 6] + pl[:, 5])
        if not go.any():
            break
        j[go] += 1
        pl = rec(j)
    tx = pl[:, 0] + (st - pl[:, 6]) * pl[:, 3]
    ty = pl[:, 1] + (st - pl[:, 6]) * pl[:, 4]
    la = wrap_angle(atan2(ty - py, tx - px) - psi)
    he = wrap_angle(pt[:, 2] - psi)
    last = rec(nseg - 1)
    total = last[:, 6] + last[:, 5]
    nav = np.stack([state[3], state[4], state[5], la, he, cte], axis=1)
    return nav, along, total


def closeness(ranges, sensor_range):
    """Perception observation 1 - range/sensor_range (utils/dataloader.py:56),
    rounded once to fp32, shape [V,1,R]."""
    return (1.0 - ranges / sensor_range).astype(np.float32)[:, None, :]


# ----------------------------------------------------------------------------
# the whole step, in the order the device path runs it
# ----------------------------------------------------------------------------
class OracleEnv:
    """Vectorised CPU env with the same buffers as the device library."""

    def __init__(self, cfg, ray_dirs):
        self.cfg = cfg
        self.ray_dirs = np.ascontiguousarray(ray_dirs, dtype=np.float64)
        V = cfg["n_envs"]
        self.state = np.zeros((6, V))
        self.t_step = np.zeros(V, dtype=np.int32)
        straight = np.zeros((V, 6))
        straight[:, 3] = 1.0
        self.set_path(straight)
        # in-step restart: an env whose done flag is set is reset by the next
        # step() instead of being integrated (reward 0, action ignored)
        self.auto_reset = False
        # ... and, with `regenerate`, gets a new scene from the generator
        # (oracle/scene_oracle.py) for its next episode number
        self.regenerate = False
        self.episode = np.zeros(V, dtype=np.int32)
        self.generator = None   # (seed, params)

    def set_scene(self, circles, poly_pose, poly_verts, poly_nv, poly_heading=None):
        self.circles = np.array(circles, dtype=np.float64, copy=True)
        self.poly_pose = np.array(poly_pose, dtype=np.float64, copy=True)
        self.poly_verts = np.array(poly_verts, dtype=np.float64, copy=True)
        self.poly_nv = np.array(poly_nv, dtype=np.int32, copy=True)
        if poly_heading is None:
            poly_heading = np.zeros(self.poly_pose.shape[:2])
        self.poly_heading = np.array(poly_heading, dtype=np.float64, copy=True)
        self.poly_reach2 = polygon_reach2(self.poly_verts, self.poly_nv)
        self.circles0 = self.circles.copy()
        self.poly_pose0 = self.poly_pose.copy()

    def set_path(self, path, nseg=None):
        """One straight segment per env [V,6], or a polyline [V,KS,6] with `nseg` [V] segments in use."""
        self.path, self.path_nseg = path_table(path, nseg)
        self.path_seg = np.zeros(self.path.shape[0], dtype=np.int32)

    def generate_scene(self, seed, params=None, env_ids=None):
        """Scene, paths and initial states of every env from the generator, episode 0.
        env_ids: the generator's env numbers of this oracle's envs (default 0..V-1;
        a sample of a larger batch passes the sampled indices)."""
        from oracle import scene_oracle

        V = self.state.shape[1]
        C, P, NV = self.cfg["n_circles"], self.cfg["n_polys"], self.cfg["n_verts"]
        self.generator = (int(seed), params)
        self.env_ids = np.arange(V) if env_ids is None else np.asarray(env_ids, dtype=np.int64)
        self.episode[:] = 0
        sc = scene_oracle.generate(int(seed), self.env_ids, self.episode, C, P, NV, params)
        self.set_scene(sc["circles"], sc["poly_pose"], sc["poly_verts"], sc["poly_nv"], sc["poly_heading"])
        self.set_path(sc["path"])
        self.init0 = sc["init_state"]

    def reset(self, init_state, mask=None):
        # None: the initial states of the last reset call or of the generator
        init_state = self.init0 if init_state is None else np.asarray(init_state, dtype=np.float64)
        self.init0 = init_state.copy()
        # a reset also puts the env's obstacles back where they were uploaded
        if mask is None:
            self.state = init_state.copy()
            self.t_step[:] = 0
            self.path_seg[:] = 0
            self.circles = self.circles0.copy()
            self.poly_pose = self.poly_pose0.copy()
        else:
            m = np.asarray(mask).astype(bool)
            self.state[:, m] = init_state[:, m]
            self.t_step[m] = 0
            self.path_seg[m] = 0
            self.circles[m] = self.circles0[m]
            self.poly_pose[m] = self.poly_pose0[m]
        self.sense()

    def sense(self, is_step=False):
        cfg = self.cfg
        R = cfg["sensor_range"]
        self.poly_reach2 = polygon_reach2(self.poly_verts, self.poly_nv)   # the vertices may have been regenerated
        self.ranges, self.hit = lidar(self.state, self.circles, self.poly_pose,
                                      self.poly_verts, self.poly_nv, self.ray_dirs, R, self.poly_reach2)
        (self.obst_dist, self.collision, self.min_dist,
         self.min_idx) = obstacle_distances(self.state, self.circles, self.poly_pose,
                                            self.poly_verts, self.poly_nv,
                                            cfg["vessel_radius"], float(cfg.get("contact_dist", -2.0)))
        self.own, self.meta = encounter_metadata(self.state, self.circles, self.poly_pose,
                                                 self.poly_heading, self.obst_dist)
        if int(cfg.get("risk_model", 1)) == 1:
            from .ref_cri import reference_distances, reference_metadata, risk_reference
            dref = reference_distances(self.state, self.circles, self.poly_pose, self.poly_verts, self.poly_nv,
                                       self.obst_dist, cfg["vessel_radius"])
            self.meta = reference_metadata(self.state, self.circles, self.poly_pose, self.meta)
            self.meta[:, :, 1] = dref
            self.risk = risk_reference(self.state, self.circles, self.poly_pose, dref)
        else:
            self.risk = risk_default(self.state, self.circles, self.poly_pose, self.obst_dist,
                                     R, cfg["risk"])
        self.perception = closeness(self.ranges, R)
        prev_along = getattr(self, "along", None)
        nav, self.along, total = navigation_features(self.state, self.path, cfg["look_ahead"], self.path_nseg,
                                                     self.path_seg)
        self.navigation = nav.astype(np.float32)[:, None, :]
        self.reached_goal = ((total - self.along) <= cfg["goal_radius"]).astype(np.uint8)
        self.truncated = (self.t_step >= cfg["max_steps"]).astype(np.uint8)
        self.done = (self.collision | self.reached_goal | self.truncated).astype(np.uint8)
        if is_step:
            w_p, w_col, w_g, w_cte = cfg["reward"]
            rew = w_p * (self.along - prev_along)
            rew = rew - w_cte * np.abs(nav[:, 5])
            rew = rew - np.where(self.collision != 0, w_col, 0.0)
            rew = rew + np.where(self.reached_goal != 0, w_g, 0.0)
            self.reward = rew
        else:
            self.reward = np.zeros(self.state.shape[1])

    def step(self, action):
        cfg = self.cfg
        action = np.asarray(action, dtype=np.float64)
        restart = self.done.astype(bool) if self.auto_reset else np.zeros(self.state.shape[1], dtype=bool)
        advance_obstacles(self.circles, self.poly_pose, cfg["dt"])
        self.state = step_dynamics(self.state, action, cfg)
        self.t_step = self.t_step + 1
        if restart.any():
            self.episode[restart] += 1
            if self.regenerate:
                from oracle import scene_oracle

                C, P, NV = cfg["n_circles"], cfg["n_polys"], cfg["n_verts"]
                sc = scene_oracle.generate(self.generator[0], self.env_ids[restart], self.episode[restart], C, P, NV,
                                           self.generator[1])
                self.circles0[restart] = sc["circles"]
                self.poly_pose0[restart] = sc["poly_pose"]
                self.poly_verts[restart] = sc["poly_verts"]
                self.poly_nv[restart] = sc["poly_nv"]
                self.poly_heading[restart] = sc["poly_heading"]
                self.path[restart, 0, :6] = sc["path"]
                self.path[restart, 0, 6] = 0.0
                self.path_nseg[restart] = 1
                self.init0[:, restart] = sc["init_state"]
            self.state[:, restart] = self.init0[:, restart]
            self.t_step[restart] = 0
            self.path_seg[restart] = 0
            self.circles[restart] = self.circles0[restart]
            self.poly_pose[restart] = self.poly_pose0[restart]
        self.sense(is_step=True)
        self.reward[restart] = 0.0


# ----------------------------------------------------------------------------
# scalar (one vessel, pure Python float) forms used to cross-check the
# vectorised code on small cases
# ----------------------------------------------------------------------------
def lidar_scalar(p, psi, circles, polys_world, ray_dirs, sensor_range, poly_info=None):
    """p = (x, y); circles = [(cx, cy, rad)]; polys_world = list of vertex
    lists [(x, y), ...] (empty list = absent); poly_info = per polygon
    (pose x, pose y, reach2 of `polygon_reach2`), None: no polygon is taken as
    convex.  Python-float restatement of `lidar` for one vessel."""
    import math
    s, c = sincos(np.float64(psi))
    s = float(s)
    c = float(c)
    out_r = []
    out_i = []
    for i in range(len(ray_dirs)):
        cb = float(ray_dirs[i][0])
        sb = float(ray_dirs[i][1])
        dx = c * cb - s * sb
        dy = s * cb + c * sb
        best = sensor_range
        hit = -1
        for j, (cx, cy, rad) in enumerate(circles):
            if not rad > 0.0:
                continue
            mx = cx - p[0]
            my = cy - p[1]
            cc = (mx * mx + my * my) - rad * rad
            b = mx * dx + my * dy
            disc = b * b - cc
            if not disc >= 0.0:
                continue
            sq = math.sqrt(disc)
            t = b - sq
            if t < 0.0:
                t = b + sq
            if t >= 0.0 and t < best:
                best = t
                hit = j
        for q, verts in enumerate(polys_world):
            nv = len(verts)
            outside = False
            if poly_info is not None:
                X, Y, reach2 = (float(t) for t in poly_info[q])
                mx = X - p[0]
                my = Y - p[1]
                outside = reach2 >= 0.0 and (mx * mx + my * my) > reach2
            for k in range(nv):
                Ax, Ay = verts[k]
                Bx, By = verts[(k + 1) % nv]
                ax = Ax - p[0]
                ay = Ay - p[1]
                ex = Bx - Ax
                ey = By - Ay
                nt = ax * ey - ay * ex
                den = dx * ey - dy * ex
                ns = ax * dy - ay * dx
                if den > 0.0:
                    ok = 0.0 <= ns <= den
                elif den < 0.0:
                    ok = den <= ns <= 0.0
                else:
                    ok = False
                if not ok or (outside and nt > 0.0):
                    continue
                t = nt / den
                if t >= 0.0 and t < best:
                    best = t
                    hit = len(circles) + q
        out_r.append(best)
        out_i.append(hit)
    return np.array(out_r), np.array(out_i, dtype=np.int32)
