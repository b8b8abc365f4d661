/* CPU oracle of the AUV simulation step in plain C (TEST INFRASTRUCTURE ONLY).
 *
 * A second, scalar-per-env restatement of oracle/auv_oracle.py (which remains
 * the normative text): same arithmetic contract -- IEEE binary64, unfused
 * + - * / sqrt floor rint in the order written, sin/cos/atan2 as fixed
 * polynomial sequences -- so its results equal the numpy oracle's bit for bit
 * (tests/test_c_oracle_cpu.py).  Build with -ffp-contract=off and without
 * -ffast-math.  Used by the tests as a cross-check of the numpy code and by
 * bench.py as the multi-core CPU baseline; the product never links it.
 *
 * PARITY UNPINNED for the simulation itself: the reference checkout does not
 * contain the simulator (radarCNN.py:232 only names the absent gym_auv
 * package).  Interface facts that it does fix: 180 rays / 150 m /
 * closeness = 1 - range/150 (utils/dataloader.py:13,56), metadata record
 * [heading_ts, dist, speed, bearing, course] (utils/explore_dataset.py:45). */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define PI 0x1.921fb54442d18p+1
#define TWO_PI 0x1.921fb54442d18p+2
#define INV_TWO_PI 0x1.45f306dc9c883p-3
#define PIO2 0x1.921fb54442d18p+0
#define PIO4 0x1.921fb54442d18p-1
#define PIO8 0x1.921fb54442d18p-2
#define TWO_OVER_PI 0x1.45f306dc9c883p-1
#define PIO2_HI 0x1.921fb50000000p+0
#define PIO2_LO 0x1.110b4611a6263p-26
#define ATAN_T1 0.198912367379658
#define ATAN_T3 0.6681786379192989
#define TAN_PIO8 0x1.a827999fcef32p-2

static const double SIN_C[8] = {-0.16666666666666666, 0.008333333333333333, -0.0001984126984126984,
                                2.7557319223985893e-06, -2.505210838544172e-08, 1.6059043836821613e-10,
                                -7.647163731819816e-13, 2.8114572543455206e-15};
static const double COS_C[9] = {-0.5, 0.041666666666666664, -0.001388888888888889, 2.48015873015873e-05,
                                -2.755731922398589e-07, 2.08767569878681e-09, -1.1470745597729725e-11,
                                4.779477332387385e-14, -1.5619206968586225e-16};
static const double ATAN_C[11] = {-1.0 / 3.0, 1.0 / 5.0, -1.0 / 7.0, 1.0 / 9.0, -1.0 / 11.0, 1.0 / 13.0,
                                  -1.0 / 15.0, 1.0 / 17.0, -1.0 / 19.0, 1.0 / 21.0, -1.0 / 23.0};
static const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 4.0, 0, 0, 0, 0},
    {3.0 / 32.0, 9.0 / 32.0, 0, 0, 0},
    {1932.0 / 2197.0, -7200.0 / 2197.0, 7296.0 / 2197.0, 0, 0},
    {439.0 / 216.0, -8.0, 3680.0 / 513.0, -845.0 / 4104.0, 0},
    {-8.0 / 27.0, 2.0, -3544.0 / 2565.0, 1859.0 / 4104.0, -11.0 / 40.0},
};
static const double RK_B5[6] = {16.0 / 135.0, 0.0, 6656.0 / 12825.0, 28561.0 / 56430.0, -9.0 / 50.0, 2.0 / 55.0};

enum { MP_MINV11, MP_MINV22, MP_MINV23, MP_MINV32, MP_MINV33, MP_MU, MP_MXG, MP_RESERVED,
       MP_XU, MP_XUU, MP_YV, MP_YVV, MP_YR, MP_NV, MP_NR, MP_NRR };

typedef struct {
    int32_t V, R, C, P, NV, max_steps;
    int32_t risk_model, KS;     /* 0: closest-point-of-approach index, 1: oc_risk_reference; KS: segment slots per path */
    double dt, sensor_range, vessel_radius, thrust_max, moment_max, look_ahead, goal_radius;
    double risk[3], reward[4], model[16];
    double contact_dist;        /* collision <=> nearest obstacle distance below this */
} oc_config;

typedef struct {
    /* inputs / state (updated in place) */
    double *state;      /* [6,V] */
    int32_t *t_step;    /* [V] */
    double *circles;    /* [V,C,5] */
    double *poly_pose;  /* [V,P,6] */
    const double *poly_verts, *poly_heading, *path, *ray_dirs;
    const int32_t *poly_nv;
    double *along;      /* [V] in: previous along-track position, out: new */
    /* outputs */
    double *ranges; int32_t *hit; float *perception; double *obst_dist; uint8_t *collision;
    double *min_dist; int32_t *min_idx; double *own, *meta, *risk; float *navigation;
    uint8_t *reached_goal, *truncated, *done; double *reward;
    const int32_t *path_nseg;   /* [V] segments in use; path is [V,KS,8] = x0,y0,angle,cos,sin,length,arc length before,0 */
    int32_t *path_seg;          /* [V] the segment each vessel is on (in/out) */
} oc_buffers;

/* oracle/ref_cri.py: bearing_membership, cri_core, risk_reference -- the
 * reference's risk index as recovered from its dataset
 * files (data/metadata_MovingObstaclesNoRules.csv, data/risk_MovingObstaclesNoRules.csv). */
static double oc_cri_terms(double along, double cross, int moving, double d, double b, double aspect) {
    const double D1 = 32.0, D2 = 160.0, DA = 20.08, DB = 45.18;
    const double dc = fabs(cross), ln = fabs(along);
    const int near = dc <= D1;
    const double lin = (D2 - dc) * 0.0078125;
    const double u_dc = near ? 1.0 : (lin > 0.0 ? lin : 0.0);
    const double r2 = D2 * D2 - dc * dc;
    const double t2 = sqrt(r2 > 0.0 ? r2 : 0.0);
    const double r1 = D1 * D1 - dc * dc;
    const double t1 = near ? sqrt(r1 > 0.0 ? r1 : 0.0) : D1 - dc;
    const double den = t2 - t1;
    double q = (t2 - ln) / (den > 0.0 ? den : 1.0);
    q = q > 0.0 ? q : 0.0;
    q = q < 1.0 ? q : 1.0;
    const double u_ta = ln <= t1 ? 1.0 : q;
    const double o = 1.0 - ln * 0.03125;
    const double u_tr = o > 0.0 ? o : 0.0;
    const double u_t = along < 0.0 ? u_ta : u_tr;
    const double ct = moving ? (0.3 * u_dc) * u_t : 0.0;
    const double span = DB - d;
    const double qd = span > 0.0 ? span / (DB - DA) : 0.0;
    const double u_d = d <= DA ? 1.0 : qd * qd;
    const double bd = b * 57.29577951308232;
    const double q1 = fabs(90.0 - bd) / 80.0, q2 = fabs(bd + 270.0) / 225.0;
    const double u_b = bd < -45.0 ? q2 * q2 : (bd <= 10.0 ? 1.0 : (bd < 90.0 ? q1 * q1 : 0.0));
    const double total = (ct + 0.3 * u_d) + (0.2 * u_b + 0.2 * aspect);
    return total > 0.0 ? total : 0.0;
}

/* oracle/auv_oracle.py: sincos */
static void oc_sincos(double x, double *s, double *c) {
    const double k = rint(x * TWO_OVER_PI);
    const double r = (x - k * PIO2_HI) - k * PIO2_LO;
    const double r2 = r * r;
    double ps = SIN_C[7];
    for (int i = 6; i >= 0; --i) ps = SIN_C[i] + r2 * ps;
    const double sr = r + (r * r2) * ps;
    double pc = COS_C[8];
    for (int i = 7; i >= 0; --i) pc = COS_C[i] + r2 * pc;
    const double cr = 1.0 + r2 * pc;
    switch ((int64_t)k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

/* oracle/auv_oracle.py: atan2 */
static double oc_atan2(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    if (ax == 0.0 && ay == 0.0) return 0.0;
    const int swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    const double z = num / den;
    const double c = z > ATAN_T3 ? 1.0 : (z > ATAN_T1 ? TAN_PIO8 : 0.0);
    const double base = z > ATAN_T3 ? PIO4 : (z > ATAN_T1 ? PIO8 : 0.0);
    const double w = (z - c) / (1.0 + z * c);
    const double w2 = w * w;
    double p = ATAN_C[10];
    for (int i = 9; i >= 0; --i) p = ATAN_C[i] + w2 * p;
    double a = base + (w + (w * w2) * p);
    if (swap) a = PIO2 - a;
    if (x < 0.0) a = PI - a;
    if (y < 0.0) a = -a;
    return a;
}

static double oc_wrap(double a) { return a - floor((a + PI) * INV_TWO_PI) * TWO_PI; }

/* oracle/ref_cri.py: reference_motion */
typedef struct { int still; double ss, wx, wy, ww, rr; } oc_ref_motion;
static oc_ref_motion oc_reference_motion(double ovx, double ovy, double vox, double voy, double cpsi, double spsi,
                                         double psi) {
    oc_ref_motion m;
    m.ss = sqrt(vox * vox + voy * voy);
    const double sx = m.ss * cpsi, sy = m.ss * spsi;
    m.still = ovx == 0.0 && ovy == 0.0;
    const double oo = ovx * ovx + ovy * ovy;
    const double X = cpsi * ovy - spsi * ovx;
    const int mirror = (spsi < 0.0 && ovy > 0.0 && X < 0.0) || (spsi > 0.0 && ovy < 0.0 && X > 0.0);
    const double dot = sx * ovx + sy * ovy;
    const double f = (2.0 * dot) / (m.still ? 1.0 : oo);
    const double ex = mirror ? f * ovx - sx : sx;
    const double ey = mirror ? f * ovy - sy : sy;
    const double rx = ovx - ex, ry = ovy - ey;
    m.rr = rx * rx + ry * ry;
    m.wx = m.still ? (psi > 0.0 ? 1.0 : -1.0) : rx;
    m.wy = m.still ? 0.0 : ry;
    m.ww = m.still ? 1.0 : m.rr;
    return m;
}

/* oracle/ref_cri.py: risk_reference */
static double oc_risk_reference(double mx, double my, double ovx, double ovy, double vox, double voy, double dist,
                                double cpsi, double spsi, double psi, int seen) {
    const oc_ref_motion rm = oc_reference_motion(ovx, ovy, vox, voy, cpsi, spsi, psi);
    const double mm = mx * mx + my * my;
    const double den = sqrt(mm * rm.ww);
    const double dens = den > 0.0 ? den : 1.0;
    const int moving = rm.still ? rm.ss > 0.0 : rm.rr > 0.0;
    const double mo = mx * ovx + my * ovy;
    const double ad = mm * (ovx * ovx + ovy * ovy);
    const int has_ad = ad > 0.0;
    const double aspect = has_ad ? 0.0 - (mo * fabs(mo)) / ad : 0.0;
    const double b = oc_wrap(oc_atan2(my, mx) - psi);
    const double along = dist * ((mx * rm.wx + my * rm.wy) / dens);
    const double cross = dist * ((mx * rm.wy - my * rm.wx) / dens);
    const double r = oc_cri_terms(along, cross, moving, dist, b, aspect);
    return seen ? r : 0.0;   /* seen: the slot holds an obstacle */
}

/* oracle/auv_oracle.py: vessel_rhs */
static void oc_rhs(const double y[6], double fx, double mz, const double *mp, double out[6]) {
    const double psi = y[2], u = y[3], v = y[4], r = y[5];
    double s, c;
    oc_sincos(psi, &s, &c);
    const double n1 = (mp[MP_XU] + mp[MP_XUU] * fabs(u)) * u;
    const double n2 = (mp[MP_YV] + mp[MP_YVV] * fabs(v)) * v + (mp[MP_MU] * u + mp[MP_YR]) * r;
    const double n3 = mp[MP_NV] * v + ((mp[MP_MXG] * u + mp[MP_NR]) + mp[MP_NRR] * fabs(r)) * r;
    const double r1 = fx - n1, r2 = 0.0 - n2, r3 = mz - n3;
    out[0] = c * u - s * v;
    out[1] = s * u + c * v;
    out[2] = r;
    out[3] = mp[MP_MINV11] * r1;
    out[4] = mp[MP_MINV22] * r2 + mp[MP_MINV23] * r3;
    out[5] = mp[MP_MINV32] * r2 + mp[MP_MINV33] * r3;
}

/* oracle/auv_oracle.py: step_dynamics */
static void oc_dynamics(const oc_config *cf, double y[6], double a0, double a1) {
    const double h = cf->dt;
    const double fx = fmin(fmax(a0, 0.0), 1.0) * cf->thrust_max;
    const double mz = fmin(fmax(a1, -1.0), 1.0) * cf->moment_max;
    double k[6][6];
    for (int i = 0; i < 6; ++i) {
        double yi[6];
        if (i == 0) {
            memcpy(yi, y, sizeof yi);
        } else {
            for (int f = 0; f < 6; ++f) {
                double acc = RK_A[i][0] * k[0][f];
                for (int j = 1; j < i; ++j) acc = acc + RK_A[i][j] * k[j][f];
                yi[f] = y[f] + h * acc;
            }
        }
        oc_rhs(yi, fx, mz, cf->model, k[i]);
    }
    for (int f = 0; f < 6; ++f) {
        double acc = RK_B5[0] * k[0][f];
        for (int j = 2; j < 6; ++j) acc = acc + RK_B5[j] * k[j][f];
        y[f] = y[f] + h * acc;
    }
    y[2] = oc_wrap(y[2]);
}

static void world_vertex(const double *pose, const double *v, double *wx, double *wy) {
    *wx = pose[0] + (pose[2] * v[0] - pose[3] * v[1]);
    *wy = pose[1] + (pose[3] * v[0] + pose[2] * v[1]);
}

/* everything that follows the state update, for env e (oracle/auv_oracle.py: OracleEnv.sense) */
/* oracle/auv_oracle.py: polygon_reach2 -- squared distance of the farthest vertex from the pose point, or -1
 * unless every vertex lies strictly to the left of every edge it does not belong to (strictly convex,
 * counter-clockwise).  v: body-frame vertices [NV,2], nv of them in use. */
static double oc_polygon_reach2(const double *v, int nv) {
    if (nv < 3) return -1.0;
    double reach2 = 0.0;
    for (int k = 0; k < nv; ++k) {
        const int k1 = (k + 1 < nv) ? k + 1 : 0;
        const double ex = v[2 * k1] - v[2 * k], ey = v[2 * k1 + 1] - v[2 * k + 1];
        const double rr = v[2 * k] * v[2 * k] + v[2 * k + 1] * v[2 * k + 1];
        if (rr > reach2) reach2 = rr;
        for (int j = 0; j < nv; ++j) {
            if (j == k || j == k1) continue;
            const double wx = v[2 * j] - v[2 * k], wy = v[2 * j + 1] - v[2 * k + 1];
            if (!(ex * wy - ey * wx > 0.0)) return -1.0;
        }
    }
    return reach2;
}

static void oc_sense_env(const oc_config *cf, const oc_buffers *b, int e, int is_step) {
    const int V = cf->V, R = cf->R, C = cf->C, P = cf->P, NV = cf->NV, O = C + P;
    const double px = b->state[e], py = b->state[(size_t)V + e], psi = b->state[2 * (size_t)V + e];
    const double u = b->state[3 * (size_t)V + e], v = b->state[4 * (size_t)V + e], r = b->state[5 * (size_t)V + e];
    const double range = cf->sensor_range;
    const double *circ = b->circles + (size_t)e * C * 5;
    const double *pose = b->poly_pose + (size_t)e * P * 6;
    const double *verts = b->poly_verts + (size_t)e * P * NV * 2;
    const int32_t *nvs = b->poly_nv + (size_t)e * P;
    double s, c;
    oc_sincos(psi, &s, &c);

    /* lidar: per-primitive quantities once per env, then every ray against every primitive */
    enum { MAXPRIM = 4096 };
    double cmx[MAXPRIM], cmy[MAXPRIM], ccc[MAXPRIM];
    double sax[MAXPRIM], say[MAXPRIM], sex[MAXPRIM], sey[MAXPRIM], snt[MAXPRIM];
    int con[MAXPRIM], sid[MAXPRIM];
    int n_seg = 0;
    for (int j = 0; j < C; ++j) {
        const double rad = circ[j * 5 + 2];
        cmx[j] = circ[j * 5] - px;
        cmy[j] = circ[j * 5 + 1] - py;
        ccc[j] = (cmx[j] * cmx[j] + cmy[j] * cmy[j]) - rad * rad;
        con[j] = rad > 0.0;
    }
    for (int p = 0; p < P; ++p) {
        const int nv = nvs[p];
        /* a convex polygon seen from beyond its farthest vertex shows only the edges that face the vessel */
        const double reach2 = oc_polygon_reach2(verts + (size_t)p * NV * 2, nv);
        const double pmx = pose[p * 6] - px, pmy = pose[p * 6 + 1] - py;
        const int outside = reach2 >= 0.0 && (pmx * pmx + pmy * pmy) > reach2;
        for (int k = 0; k < nv; ++k) {   /* edges k >= nv never count */
            const int k1 = (k + 1 < nv) ? k + 1 : 0;
            double Ax, Ay, Bx, By;
            world_vertex(pose + p * 6, verts + ((size_t)p * NV + k) * 2, &Ax, &Ay);
            world_vertex(pose + p * 6, verts + ((size_t)p * NV + k1) * 2, &Bx, &By);
            sax[n_seg] = Ax - px;
            say[n_seg] = Ay - py;
            sex[n_seg] = Bx - Ax;
            sey[n_seg] = By - Ay;
            snt[n_seg] = sax[n_seg] * sey[n_seg] - say[n_seg] * sex[n_seg];
            sid[n_seg] = C + p;
            if (outside && snt[n_seg] > 0.0) continue;   /* faces away: not a LiDAR surface */
            ++n_seg;
        }
    }
    for (int i = 0; i < R; ++i) {
        const double cb = b->ray_dirs[2 * i], sb = b->ray_dirs[2 * i + 1];
        const double dx = c * cb - s * sb, dy = s * cb + c * sb;
        double best = range;
        int32_t hit = -1;
        for (int j = 0; j < C; ++j) {
            const double bb = cmx[j] * dx + cmy[j] * dy;
            const double disc = bb * bb - ccc[j];
            if (!(con[j] && disc >= 0.0)) continue;
            const double sq = sqrt(disc);
            double t = bb - sq;
            if (t < 0.0) t = bb + sq;
            if (t >= 0.0 && t < best) { best = t; hit = j; }
        }
        for (int q = 0; q < n_seg; ++q) {
            const double den = dx * sey[q] - dy * sex[q];
            const double ns = sax[q] * dy - say[q] * dx;
            const int inseg = den > 0.0 ? (ns >= 0.0 && ns <= den) : (ns <= 0.0 && ns >= den);
            if (!(den != 0.0 && inseg)) continue;
            const double t = snt[q] / den;
            if (t >= 0.0 && t < best) { best = t; hit = sid[q]; }
        }
        b->ranges[(size_t)e * R + i] = best;
        b->hit[(size_t)e * R + i] = hit;
        b->perception[(size_t)e * R + i] = (float)(1.0 - best / range);
    }

    /* signed hull distances, nearest obstacle, collision */
    double *dist = b->obst_dist + (size_t)e * O;
    for (int j = 0; j < C; ++j) {
        const double rad = circ[j * 5 + 2];
        const double mx = circ[j * 5] - px, my = circ[j * 5 + 1] - py;
        const double d = (sqrt(mx * mx + my * my) - rad) - cf->vessel_radius;
        dist[j] = rad > 0.0 ? d : INFINITY;
    }
    for (int p = 0; p < P; ++p) {
        const int nv = nvs[p];
        double dmin = INFINITY;
        int inside = 0;
        for (int k = 0; k < NV; ++k) {
            const int k1 = (k + 1 < nv) ? k + 1 : 0;
            double Ax, Ay, Bx, By;
            world_vertex(pose + p * 6, verts + ((size_t)p * NV + k) * 2, &Ax, &Ay);
            world_vertex(pose + p * 6, verts + ((size_t)p * NV + k1) * 2, &Bx, &By);
            const double ax = Ax - px, ay = Ay - py, by = By - py, ex = Bx - Ax, ey = By - Ay;
            const double ee = ex * ex + ey * ey;
            double tp = 0.0;
            if (ee > 0.0) {
                tp = (0.0 - (ax * ex + ay * ey)) / ee;
                tp = fmin(fmax(tp, 0.0), 1.0);
            }
            const double qx = ax + tp * ex, qy = ay + tp * ey;
            const double d = sqrt(qx * qx + qy * qy);
            const int act = k < nv;
            if (act && d < dmin) dmin = d;
            const double nt = ax * ey - ay * ex;
            const int up = ay <= 0.0 && by > 0.0 && nt > 0.0;
            const int dn = by <= 0.0 && ay > 0.0 && nt < 0.0;
            inside ^= act && (up || dn);
        }
        dist[C + p] = nv > 0 ? (inside ? -dmin : dmin) - cf->vessel_radius : INFINITY;
    }
    double md = INFINITY;
    int32_t mi = -1;
    for (int o = 0; o < O; ++o)
        if (dist[o] < md) { md = dist[o]; mi = o; }
    b->min_dist[e] = md;
    b->min_idx[e] = mi;
    const uint8_t coll = md < cf->contact_dist;
    b->collision[e] = coll;

    /* metadata and risk */
    const double vox = c * u - s * v, voy = s * u + c * v;
    b->own[2 * (size_t)e] = sqrt(vox * vox + voy * voy);
    b->own[2 * (size_t)e + 1] = psi;
    const double w_d = cf->risk[0], w_c = cf->risk[1], t_c = cf->risk[2];
    for (int o = 0; o < O; ++o) {
        double cx, cy, vx, vy, hd;
        if (o < C) {
            cx = circ[o * 5]; cy = circ[o * 5 + 1]; vx = circ[o * 5 + 3]; vy = circ[o * 5 + 4]; hd = 0.0;
        } else {
            const double *q = pose + (o - C) * 6;
            cx = q[0]; cy = q[1]; vx = q[4]; vy = q[5]; hd = b->poly_heading[(size_t)e * P + (o - C)];
        }
        const double mx = cx - px, my = cy - py, rvx = vx - vox, rvy = vy - voy;
        double *m = b->meta + ((size_t)e * O + o) * 5;
        m[0] = hd;
        m[1] = dist[o];
        m[2] = sqrt(rvx * rvx + rvy * rvy);
        m[3] = oc_wrap(oc_atan2(my, mx) - psi);
        m[4] = oc_wrap(oc_atan2(rvy, rvx) - psi);
        double dref = dist[o];
        if (cf->risk_model == 1) { /* oracle/ref_cri.py: reference_distances, reference_metadata */
            if (o >= C && nvs[o - C] > 0) {
                double hi = -INFINITY, lo = INFINITY;
                for (int k = 0; k < nvs[o - C]; ++k) {
                    const double y = verts[((size_t)(o - C) * NV + k) * 2 + 1];
                    if (y > hi) hi = y;
                    if (y < lo) lo = y;
                }
                dref = (sqrt(mx * mx + my * my) - (hi - lo)) - cf->vessel_radius;
            }
            m[1] = dref;
            const oc_ref_motion rm = oc_reference_motion(vx, vy, vox, voy, c, s, psi);
            m[0] = rm.still ? 0.0 : oc_atan2(vy, vx);
            m[2] = rm.still ? rm.ss : sqrt(rm.rr);
            m[4] = rm.still ? (psi > 0.0 ? 0.0 : 0.0 - PI) : oc_atan2(rm.wy, rm.wx);
        }
        const double d = fmax(dist[o], 0.0);
        const int inrange = d < range;
        const double fd = 1.0 - (inrange ? d : 0.0) / range;
        const double term_d = w_d * (fd * fd);
        const double vv = rvx * rvx + rvy * rvy;
        const double closing = 0.0 - (mx * rvx + my * rvy);
        const int has_c = vv > 0.0 && closing > 0.0;
        const double tcpa = (has_c ? closing : 0.0) / (has_c ? vv : 1.0);
        const double qx = mx + rvx * tcpa, qy = my + rvy * tcpa;
        const double dcpa = sqrt(qx * qx + qy * qy);
        const double fc = 1.0 - fmin(dcpa, range) / range;
        const double term_c = (w_c * (fc * fc)) * (t_c / (t_c + tcpa));
        const double rr = term_d + (has_c ? term_c : 0.0);
        b->risk[(size_t)e * O + o] = cf->risk_model == 1
            ? oc_risk_reference(mx, my, vx, vy, vox, voy, dref, c, s, psi, dref < INFINITY) : (inrange ? rr : 0.0);
    }

    /* navigation, flags, reward (oracle/auv_oracle.py: navigation_features) */
    const int KS = cf->KS > 0 ? cf->KS : 1;
    const double *tab = b->path + (size_t)e * KS * 8;
    const int nseg = b->path_nseg[e];
    int seg = b->path_seg[e];
    const double *pt = tab + (size_t)seg * 8;
    double rx = px - pt[0], ry = py - pt[1];
    double loc = rx * pt[3] + ry * pt[4];
    while (seg + 1 < nseg && loc > pt[5]) {   /* past the end of its segment: on to the next one, never back */
        ++seg;
        pt += 8;
        rx = px - pt[0];
        ry = py - pt[1];
        loc = rx * pt[3] + ry * pt[4];
    }
    b->path_seg[e] = seg;
    const double along = pt[6] + loc;
    const double cte = ry * pt[3] - rx * pt[4];
    const double st = along + cf->look_ahead;
    const double *pl = pt;
    for (int j = seg; j + 1 < nseg && st > pl[6] + pl[5]; ++j) pl += 8;
    const double tx = pl[0] + (st - pl[6]) * pl[3];
    const double ty = pl[1] + (st - pl[6]) * pl[4];
    const double la = oc_wrap(oc_atan2(ty - py, tx - px) - psi);
    const double he = oc_wrap(pt[2] - psi);
    const double total = tab[(size_t)(nseg - 1) * 8 + 6] + tab[(size_t)(nseg - 1) * 8 + 5];
    float *nav = b->navigation + (size_t)e * 6;
    nav[0] = (float)u; nav[1] = (float)v; nav[2] = (float)r;
    nav[3] = (float)la; nav[4] = (float)he; nav[5] = (float)cte;
    const double prev_along = b->along[e];
    b->along[e] = along;
    const uint8_t goal = (total - along) <= cf->goal_radius;
    const uint8_t trunc = b->t_step[e] >= cf->max_steps;
    b->reached_goal[e] = goal;
    b->truncated[e] = trunc;
    b->done[e] = coll | goal | trunc;
    double rew = 0.0;
    if (is_step) {
        rew = cf->reward[0] * (along - prev_along);
        rew = rew - cf->reward[3] * fabs(cte);
        rew = rew - (coll ? cf->reward[1] : 0.0);
        rew = rew + (goal ? cf->reward[2] : 0.0);
    }
    b->reward[e] = rew;
}

/* Envs [begin, end): when action != NULL advance obstacles, integrate the vessel
 * and count the step (OracleEnv.step), then sense; action == NULL: sense only
 * (after a reset).  Envs are independent, so disjoint ranges may run on
 * different threads. */
void oc_run(const oc_config *cf, const oc_buffers *b, const double *action, int32_t begin, int32_t end) {
    const int V = cf->V, C = cf->C, P = cf->P;
    if (C > 4096 || P * cf->NV > 4096) return;  /* beyond the per-env scratch of oc_sense_env */
    for (int e = begin; e < end; ++e) {
        if (action) {
            double *circ = b->circles + (size_t)e * C * 5;
            for (int j = 0; j < C; ++j) {
                circ[j * 5] = circ[j * 5] + circ[j * 5 + 3] * cf->dt;
                circ[j * 5 + 1] = circ[j * 5 + 1] + circ[j * 5 + 4] * cf->dt;
            }
            double *pose = b->poly_pose + (size_t)e * P * 6;
            for (int p = 0; p < P; ++p) {
                pose[p * 6] = pose[p * 6] + pose[p * 6 + 4] * cf->dt;
                pose[p * 6 + 1] = pose[p * 6 + 1] + pose[p * 6 + 5] * cf->dt;
            }
            double y[6];
            for (int f = 0; f < 6; ++f) y[f] = b->state[(size_t)f * V + e];
            oc_dynamics(cf, y, action[e], action[(size_t)V + e]);
            for (int f = 0; f < 6; ++f) b->state[(size_t)f * V + e] = y[f];
            b->t_step[e] = b->t_step[e] + 1;
        }
        oc_sense_env(cf, b, e, action != NULL);
    }
}
