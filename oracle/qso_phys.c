/* qso_phys.c -- ORACLE (test infrastructure, not product code).
 *
 * Restatement of what the reference obtains from pybullet.stepSimulation()
 * (quadruped_spring/env/quadruped_gym_env.py:218-225, world set-up :299-321, robot set-up
 * quadruped_spring/env/quadruped.py:454-519, 663-683).  pybullet==3.2.5 is an un-vendored third-party
 * dependency (setup.py:10) and absent here, so this file restates the published algorithms Bullet's
 * btMultiBodyDynamicsWorld is built from (SURVEY.md App. D lists the assumed semantics, "parity unpinned"):
 *   1. forward dynamics: Featherstone articulated-body algorithm, floating base, link coordinates
 *      (R. Featherstone, Rigid Body Dynamics Algorithms, 2008, Table 9.4), gravity (0,0,-g)
 *   2. v* = v + dt*a, every generalized velocity clamped to +-vel_cap (maxJointVelocity, quadruped.py:678-683)
 *   3. collision: foot sphere r=0.02 against the plane z=0; the other link primitives (trunk box, hip cylinders, thigh and calf boxes,
 *      payload block) flag invalid contacts and, under cfg.body_contacts, push back at up to two support points per leg; link-link
 *      contacts that involve a calf (self-collision rule) are counted
 *   4. constraint rows: per contact point 1 normal + 2 friction rows -- implicit cone (cfg.friction_cone, PyBullet's default: both friction
 *      rows from the same velocities, projected onto the disc of radius mu*f_n) or pyramid (|f_t| <= mu*f_n per direction) --, violated
 *      joint limits as unilateral rows, the six rows of the payload block's fixed constraint (cfg.payload_soft); projected Gauss-Seidel in
 *      Bullet's row order, at most `solver_iters` sweeps (gym_env.py:113,302) with PyBullet's residual early exit, velocity space
 *   5. semi-implicit Euler on positions (quaternion by exponential map)
 */
#include "qso_internal.h"

/* ------------------------------------------------------------------ spatial algebra */
typedef struct { real E[3][3]; real r[3]; } xform; /* parent->child coordinates: E = R_child_in_parent^T */

static void xm(const xform* X, const real* v, real* o) { /* motion vector parent -> child */
    real t[3], w[3];
    v3cross(X->r, v, t);                 /* r x w */
    real lin[3] = {v[3] - t[0], v[4] - t[1], v[5] - t[2]};
    m3v(X->E, v, w);
    real l2[3]; m3v(X->E, lin, l2);
    o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; o[3] = l2[0]; o[4] = l2[1]; o[5] = l2[2];
}
static void xft(const xform* X, const real* f, real* o) { /* force vector child -> parent (X^T) */
    real n[3], l[3], t[3];
    m3tv(X->E, f, n); m3tv(X->E, f + 3, l);
    v3cross(X->r, l, t);
    o[0] = n[0] + t[0]; o[1] = n[1] + t[1]; o[2] = n[2] + t[2]; o[3] = l[0]; o[4] = l[1]; o[5] = l[2];
}
static void x6(const xform* X, real M[6][6]) {
    real rx[3][3] = {{0, -X->r[2], X->r[1]}, {X->r[2], 0, -X->r[0]}, {-X->r[1], X->r[0], 0}};
    memset(M, 0, 36 * sizeof(real));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            M[i][j] = X->E[i][j];
            M[i + 3][j + 3] = X->E[i][j];
            real s = 0;
            for (int k = 0; k < 3; k++) s += X->E[i][k] * rx[k][j];
            M[i + 3][j] = -s;
        }
}
static void crm(const real* v, const real* u, real* o) { /* v x u (motion) */
    real a[3], b[3], c[3];
    v3cross(v, u, a); v3cross(v, u + 3, b); v3cross(v + 3, u, c);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(const real* v, const real* f, real* o) { /* v x* f (force) */
    real a[3], b[3], c[3];
    v3cross(v, f, a); v3cross(v + 3, f + 3, b); v3cross(v, f + 3, c);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
static void m6v(const real M[6][6], const real* v, real* o) {
    real t[6];
    for (int i = 0; i < 6; i++) { real s = 0; for (int j = 0; j < 6; j++) s += M[i][j] * v[j]; t[i] = s; }
    memcpy(o, t, sizeof(t));
}
static real dot6(const real* a, const real* b) { real s = 0; for (int i = 0; i < 6; i++) s += a[i] * b[i]; return s; }

static int solve6(const real A[6][6], const real* b, real* x) { /* Gaussian elimination, partial pivoting */
    real M[6][7];
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) M[i][j] = A[i][j]; M[i][6] = b[i]; }
    for (int c = 0; c < 6; c++) {
        int p = c; real best = fabs(M[c][c]);
        for (int r = c + 1; r < 6; r++) if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); p = r; }
        if (best == 0) return -1;
        if (p != c) for (int j = 0; j < 7; j++) { real t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
        for (int r = c + 1; r < 6; r++) {
            real f = M[r][c] / M[c][c];
            for (int j = c; j < 7; j++) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 5; i >= 0; i--) {
        real s = M[i][6];
        for (int j = i + 1; j < 6; j++) s -= M[i][j] * x[j];
        x[i] = s / M[i][i];
    }
    return 0;
}

void qso_quat_to_mat(const real* q, real R[3][3]) { /* q = (x,y,z,w); row-major like getMatrixFromQuaternion */
    real x = q[0], y = q[1], z = q[2], w = q[3];
    real d = x * x + y * y + z * z + w * w;
    real s = 2 / d;
    real xs = x * s, ys = y * s, zs = z * s;
    real wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0][0] = 1 - (yy + zz); R[0][1] = xy - wz; R[0][2] = xz + wy;
    R[1][0] = xy + wz; R[1][1] = 1 - (xx + zz); R[1][2] = yz - wx;
    R[2][0] = xz - wy; R[2][1] = yz + wx; R[2][2] = 1 - (xx + yy);
}

static void joint_xform(const qso_model* M, int i, real q, xform* X) {
    real c = cos(q), s = sin(q);
    /* R = rot(axis, q) (child axes in parent); E = R^T */
    if (M->jaxis[i] == 0) {
        real E[3][3] = {{1, 0, 0}, {0, c, s}, {0, -s, c}};
        memcpy(X->E, E, sizeof(E));
    } else {
        real E[3][3] = {{c, 0, -s}, {0, 1, 0}, {s, 0, c}};
        memcpy(X->E, E, sizeof(E));
    }
    memcpy(X->r, M->jpos[i], 3 * sizeof(real));
}

/* ------------------------------------------------------------------ kinematics + ABA with cached factors */
typedef struct {
    xform X[NB];
    real v[NB][6];          /* spatial velocity, body coordinates */
    real IA[NB][6][6];
    real U[NB][6], D[NB];
    real Rw[NB][3][3], ow[NB][3]; /* world pose of each link frame */
    real R0[3][3];
} phys_cache;

static void kinematics(const qso_model* M, const qso_dyn* s, phys_cache* C) {
    qso_quat_to_mat(s->quat, C->R0);
    memcpy(C->Rw[0], C->R0, sizeof(C->R0));
    memcpy(C->ow[0], s->pos, 3 * sizeof(real));
    m3tv(C->R0, s->vang, C->v[0]);
    m3tv(C->R0, s->vlin, C->v[0] + 3);
    for (int i = 1; i < NB; i++) {
        int p = M->parent[i];
        joint_xform(M, i, s->q[i - 1], &C->X[i]);
        xm(&C->X[i], C->v[p], C->v[i]);
        C->v[i][M->jaxis[i]] += s->qd[i - 1];
        /* world pose: R_i = R_p * E^T ; o_i = o_p + R_p * jpos */
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                real t = 0;
                for (int k = 0; k < 3; k++) t += C->Rw[p][a][k] * C->X[i].E[b][k];
                C->Rw[i][a][b] = t;
            }
        real t3[3]; m3v(C->Rw[p], M->jpos[i], t3);
        for (int a = 0; a < 3; a++) C->ow[i][a] = C->ow[p][a] + t3[a];
    }
}

/* forward dynamics; acc = [alpha_b a_b qdd] (base part: spatial acceleration in base coordinates) */
static int aba(const qso_model* M, const qso_dyn* s, const real* tau, real g, phys_cache* C, real* acc) {
    real c[NB][6], pA[NB][6], u[NB], a[NB][6];
    for (int i = 0; i < NB; i++) {
        memcpy(C->IA[i], M->I6[i], sizeof(M->I6[i]));
        real Iv[6]; m6v(M->I6[i], C->v[i], Iv);
        crf(C->v[i], Iv, pA[i]);
        if (i > 0) {
            real vJ[6] = {0, 0, 0, 0, 0, 0};
            vJ[M->jaxis[i]] = s->qd[i - 1];
            crm(C->v[i], vJ, c[i]);
        }
    }
    for (int i = NB - 1; i >= 1; i--) {
        int p = M->parent[i], ax = M->jaxis[i];
        for (int k = 0; k < 6; k++) C->U[i][k] = C->IA[i][k][ax];
        C->D[i] = C->U[i][ax];
        u[i] = tau[i - 1] - pA[i][ax];
        real Ia[6][6], pa[6], Iac[6];
        for (int r = 0; r < 6; r++)
            for (int q = 0; q < 6; q++) Ia[r][q] = C->IA[i][r][q] - C->U[i][r] * C->U[i][q] / C->D[i];
        m6v(Ia, c[i], Iac);
        for (int k = 0; k < 6; k++) pa[k] = pA[i][k] + Iac[k] + C->U[i][k] * u[i] / C->D[i];
        real X6[6][6], T[6][6];
        x6(&C->X[i], X6);
        for (int r = 0; r < 6; r++)
            for (int q = 0; q < 6; q++) { real t = 0; for (int k = 0; k < 6; k++) t += Ia[r][k] * X6[k][q]; T[r][q] = t; }
        for (int r = 0; r < 6; r++)
            for (int q = 0; q < 6; q++) { real t = 0; for (int k = 0; k < 6; k++) t += X6[k][r] * T[k][q]; C->IA[p][r][q] += t; }
        real pp[6]; xft(&C->X[i], pa, pp);
        for (int k = 0; k < 6; k++) pA[p][k] += pp[k];
    }
    real rhs[6];
    for (int k = 0; k < 6; k++) rhs[k] = -pA[0][k];
    if (solve6(C->IA[0], rhs, a[0])) return -1;
    for (int i = 1; i < NB; i++) {
        int p = M->parent[i], ax = M->jaxis[i];
        real ap[6]; xm(&C->X[i], a[p], ap);
        for (int k = 0; k < 6; k++) ap[k] += c[i][k];
        real qdd = (u[i] - dot6(C->U[i], ap)) / C->D[i];
        memcpy(a[i], ap, sizeof(ap));
        a[i][ax] += qdd;
        acc[6 + i - 1] = qdd;
    }
    real gw[3] = {0, 0, -g}, gb[3];
    m3tv(C->R0, gw, gb);
    for (int k = 0; k < 3; k++) { acc[k] = a[0][k]; acc[3 + k] = a[0][3 + k] + gb[k]; }
    return 0;
}

/* y = H^-1 x using the factors left in the cache by aba() (Bullet: calcAccelerationDeltasMultiDof) */
static void minv_apply(const qso_model* M, const phys_cache* C, const real* x, real* y) {
    real pA[NB][6], u[NB], a[NB][6];
    memset(pA, 0, sizeof(pA));
    for (int i = NB - 1; i >= 1; i--) {
        int p = M->parent[i], ax = M->jaxis[i];
        u[i] = x[6 + i - 1] - pA[i][ax];
        real pa[6], pp[6];
        for (int k = 0; k < 6; k++) pa[k] = pA[i][k] + C->U[i][k] * u[i] / C->D[i];
        xft(&C->X[i], pa, pp);
        for (int k = 0; k < 6; k++) pA[p][k] += pp[k];
    }
    real rhs[6];
    for (int k = 0; k < 6; k++) rhs[k] = x[k] - pA[0][k];
    solve6(C->IA[0], rhs, a[0]);
    for (int k = 0; k < 6; k++) y[k] = a[0][k];
    for (int i = 1; i < NB; i++) {
        int p = M->parent[i], ax = M->jaxis[i];
        real ap[6]; xm(&C->X[i], a[p], ap);
        real qdd = (u[i] - dot6(C->U[i], ap)) / C->D[i];
        memcpy(a[i], ap, sizeof(ap));
        a[i][ax] += qdd;
        y[6 + i - 1] = qdd;
    }
}

/* ------------------------------------------------------------------ collision geometry */
/* foot: sphere r=0.02 at calf-frame (0,0,-0.213)  go1.urdf:218-236 */
#define FOOT_R ((real)0.02)
static const real FOOT_OFF[3] = {0, 0, -0.213};
/* Contact breaking thresholds = 0.02 * angular-motion-disc of each link's compound shape
 * (Bullet CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD; hypothesis, DESIGN.md "contact model"). */
#define THR_FOOT ((real)0.000727)
#define THR_TRUNK ((real)0.00407)
#define THR_HIP ((real)0.00139)
#define THR_THIGH ((real)0.00433)
#define THR_CALF ((real)0.00429)

static real box_min_z(const real R[3][3], const real* o, const real* centre, const real* half) {
    real mn = 1e30;
    for (int sx = -1; sx <= 1; sx += 2)
        for (int sy = -1; sy <= 1; sy += 2)
            for (int sz = -1; sz <= 1; sz += 2) {
                real p[3] = {centre[0] + sx * half[0], centre[1] + sy * half[1], centre[2] + sz * half[2]};
                real z = o[2] + R[2][0] * p[0] + R[2][1] * p[1] + R[2][2] * p[2];
                if (z < mn) mn = z;
            }
    return mn;
}

static const real TRUNK_HALF[3] = {0.1881, 0.04675, 0.057};          /* go1.urdf:74-79 */
static const real LINK_BOX_C[3] = {0, 0, -0.1065};                     /* thigh :180-183, calf :207-210 */
static const real THIGH_HALF[3] = {0.017, 0.01225, 0.1065}, CALF_HALF[3] = {0.008, 0.008, 0.1065};
#define HIP_CYL_R ((real)0.046)
#define HIP_CYL_H ((real)0.02)
#define HIP_SELF_R ((real)0.046)   /* link-link tests treat the hip housing as a sphere of its cylinder's radius */
#define PAYLOAD_HALF ((real)0.05)  /* quadruped.py:793 */
#define THR_PAYLOAD ((real)0.00173)

static void push_contact(qso_env* e, int body_a, int body_b, int link_a, int link_b, real dist) {
    if (e->n_contacts >= QSO_MAX_CONTACTS) return;
    e->contacts[e->n_contacts].body_a = body_a; e->contacts[e->n_contacts].body_b = body_b;
    e->contacts[e->n_contacts].link_a = link_a; e->contacts[e->n_contacts].link_b = link_b;
    e->contacts[e->n_contacts].dist = dist; e->contacts[e->n_contacts].force = 0;
    e->n_contacts++;
}

/* ---- link-link contacts (URDF_USE_SELF_COLLISION, quadruped.py:533-539).  Box / box pairs: Bullet's btBoxBoxDetector reports a contact
 * iff the boxes overlap; decided here by clipping the twelve edges of each box against the other (two convex polyhedra meet iff an
 * edge of one meets the other, or one lies inside the other -- then its edges do).  The product uses the separating-axis test. */
typedef struct { real c[3]; real R[3][3]; real h[3]; } obox;   /* centre, axes as COLUMNS of R (world), half extents */
static void link_box(const phys_cache* C, int body, const real* centre_local, const real* half, obox* B) {
    real t[3]; m3v(C->Rw[body], centre_local, t);
    for (int k = 0; k < 3; k++) { B->c[k] = C->ow[body][k] + t[k]; B->h[k] = half[k]; }
    memcpy(B->R, C->Rw[body], sizeof(B->R));
}
static int segment_hits_box(const real* a, const real* d, const real* h) { /* segment a + t d, t in [0,1], box |x_k| <= h_k (local frame) */
    real t0 = 0, t1 = 1;
    for (int k = 0; k < 3; k++) {
        if (fabs(d[k]) < (real)1e-30) { if (fabs(a[k]) > h[k]) return 0; continue; }
        real ta = (-h[k] - a[k]) / d[k], tb = (h[k] - a[k]) / d[k];
        if (ta > tb) { real t = ta; ta = tb; tb = t; }
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
        if (t0 > t1) return 0;
    }
    return 1;
}
static int edges_hit(const obox* A, const obox* B) { /* does an edge of A meet B? */
    for (int ax = 0; ax < 3; ax++)
        for (int s1 = -1; s1 <= 1; s1 += 2)
            for (int s2 = -1; s2 <= 1; s2 += 2) {
                int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
                real pl[3]; pl[ax] = -A->h[ax]; pl[a1] = s1 * A->h[a1]; pl[a2] = s2 * A->h[a2];
                real dl[3] = {0, 0, 0}; dl[ax] = 2 * A->h[ax];
                real pw[3], dw[3]; m3v(A->R, pl, pw); m3v(A->R, dl, dw);
                for (int k = 0; k < 3; k++) pw[k] += A->c[k] - B->c[k];
                real pb[3], db[3]; m3tv(B->R, pw, pb); m3tv(B->R, dw, db);
                if (segment_hits_box(pb, db, B->h)) return 1;
            }
    return 0;
}
static int boxes_overlap(const obox* A, const obox* B) { return edges_hit(A, B) || edges_hit(B, A); }
static real sphere_box_dist(const real* c, real r, const obox* B) {
    real d[3] = {c[0] - B->c[0], c[1] - B->c[1], c[2] - B->c[2]}, l[3], e2 = 0;
    m3tv(B->R, d, l);
    for (int k = 0; k < 3; k++) { real q = l[k] > B->h[k] ? B->h[k] : (l[k] < -B->h[k] ? -B->h[k] : l[k]); e2 += (l[k] - q) * (l[k] - q); }
    return sqrt(e2) - r;
}

/* Contacts that the reference counts as invalid (quadruped.py:237-249): non-foot links and the payload block on the ground, and
 * link-link contacts that involve a calf.  Fills the contact list; returns the count. */
static int collect_invalid(const qso_config* cfg, qso_env* e, const phys_cache* C, const qso_dyn* s) {
    int n = 0;
    static const real zc[3] = {0, 0, 0};
    real d;
    if ((d = box_min_z(C->Rw[0], C->ow[0], zc, TRUNK_HALF)) < THR_TRUNK) { n++; push_contact(e, 1, 0, 0, -1, d); }
    for (int L = 0; L < 4; L++) {
        int ih = 1 + 3 * L;
        real az = C->Rw[ih][2][1]; /* z component of the cylinder axis (link y) */
        real s2 = 1 - az * az; if (s2 < 0) s2 = 0;
        real zmin = C->ow[ih][2] - HIP_CYL_H * fabs(az) - HIP_CYL_R * sqrt(s2);
        if (zmin < THR_HIP) { n++; push_contact(e, 1, 0, 2 + 4 * L, -1, zmin); }
        if ((d = box_min_z(C->Rw[ih + 1], C->ow[ih + 1], LINK_BOX_C, THIGH_HALF)) < THR_THIGH) { n++; push_contact(e, 1, 0, 3 + 4 * L, -1, d); }
        if ((d = box_min_z(C->Rw[ih + 2], C->ow[ih + 2], LINK_BOX_C, CALF_HALF)) < THR_CALF) { n++; push_contact(e, 1, 0, 4 + 4 * L, -1, d); }
    }
    if (e->m_pay > 0) { /* the block is a second body bolted to the base (quadruped.py:778-819); its box is aligned with the base frame */
        static const real ph[3] = {PAYLOAD_HALF, PAYLOAD_HALF, PAYLOAD_HALF};
        if (cfg->payload_soft) {
            real Rb[3][3]; qso_quat_to_mat(e->blk.quat, Rb);
            d = box_min_z(Rb, e->blk.pos, zc, ph);
        } else d = box_min_z(C->Rw[0], C->ow[0], e->r_pay, ph);
        if (d < THR_PAYLOAD) { n++; push_contact(e, 2, 0, -1, -1, d); }
    }
    if (cfg->self_collision) {
        obox trunk, thigh[4], calf[4];
        link_box(C, 0, zc, TRUNK_HALF, &trunk);
        for (int L = 0; L < 4; L++) { link_box(C, 2 + 3 * L, LINK_BOX_C, THIGH_HALF, &thigh[L]); link_box(C, 3 + 3 * L, LINK_BOX_C, CALF_HALF, &calf[L]); }
        for (int i = 0; i < 4; i++) {
            if (boxes_overlap(&calf[i], &trunk)) { n++; push_contact(e, 1, 1, 4 + 4 * i, 0, 0); }
            for (int j = 0; j < 4; j++) {
                if (j == i) continue;
                if (boxes_overlap(&calf[i], &thigh[j])) { n++; push_contact(e, 1, 1, 4 + 4 * i, 3 + 4 * j, 0); }
                if (j > i && boxes_overlap(&calf[i], &calf[j])) { n++; push_contact(e, 1, 1, 4 + 4 * i, 4 + 4 * j, 0); }
                if ((d = sphere_box_dist(C->ow[1 + 3 * j], HIP_SELF_R, &calf[i])) < THR_HIP) { n++; push_contact(e, 1, 1, 4 + 4 * i, 2 + 4 * j, d); }
                real fc[3]; m3v(C->Rw[3 + 3 * j], FOOT_OFF, fc);
                for (int k = 0; k < 3; k++) fc[k] += C->ow[3 + 3 * j][k];
                if ((d = sphere_box_dist(fc, FOOT_R, &calf[i])) < THR_FOOT) { n++; push_contact(e, 1, 1, 4 + 4 * i, 5 + 4 * j, d); }
            }
        }
    }
    (void)s;
    return n;
}

/* ------------------------------------------------------------------ payload block as a body of its own (cfg->payload_soft; ORACLE ONLY)
 * quadruped.py:778-819: createMultiBody(mass, box of half extent 0.05) at base + delta, createConstraint(robot base, block, JOINT_FIXED,
 * parentFramePosition 0, childFramePosition -delta): a btMultiBodyFixedConstraint -- three rows that hold the block's point -delta on the
 * base origin, three that hold the two frames parallel -- in the same PGS as the contacts, ERP m_erp = 0.2, impulse bound 500 N x dt. */
void qso_block_place(qso_env* e) {
    real R0[3][3]; qso_quat_to_mat(e->s.quat, R0);
    real d[3]; m3v(R0, e->r_pay, d);
    for (int k = 0; k < 3; k++) { e->blk.pos[k] = e->s.pos[k] + d[k]; e->blk.w[k] = e->s.vang[k]; }
    memcpy(e->blk.quat, e->s.quat, sizeof(e->blk.quat));
    real wxd[3]; v3cross(e->s.vang, d, wxd);
    for (int k = 0; k < 3; k++) e->blk.v[k] = e->s.vlin[k] + wxd[k];
    memset(e->blk.lam, 0, sizeof(e->blk.lam)); e->blk.gap = 0;
}

/* ------------------------------------------------------------------ one stepSimulation() */
typedef struct {
    real J[NVX], W[NVX]; /* Jacobian row and H^-1 J^T; entries 18..23: the payload block as its own body (w, v in world coordinates) */
    real dinv, rhs, lo, hi, lam;
    int fric_of;         /* index of the normal row bounding this friction row, -1 otherwise */
    real mu;
} row;

static void clamp_vel(real* v, real cap) { if (*v > cap) *v = cap; if (*v < -cap) *v = -cap; }

/* rows (normal, t1, t2) of a contact at world point p of body `body` (0 = base, 1 + 3 L + j = link j of leg L), `dist` above the plane.
 * Directions: normal +z; friction t1 = (0,-1,0), t2 = (1,0,0) (btPlaneSpace1 of +z). */
static void contact_rows(const qso_config* cfg, const qso_model* M, const phys_cache* C, const qso_dyn* s, const real* v, real mu, int body,
                         const real* p, real dist, real warm, row* nr, row* f1, row* f2, int nor_index) {
    static const real dirs[3][3] = {{0, 0, 1}, {0, -1, 0}, {1, 0, 0}};
    real dt = cfg->dt;
    for (int rr = 0; rr < 3; rr++) {
        row* r = rr == 0 ? nr : (rr == 1 ? f1 : f2);
        memset(r, 0, sizeof(*r));
        const real* d = dirs[rr];
        real rb[3], pw[3] = {p[0] - s->pos[0], p[1] - s->pos[1], p[2] - s->pos[2]};
        m3tv(C->R0, pw, rb);
        real db[3]; m3tv(C->R0, d, db);
        real ang[3]; v3cross(rb, db, ang);
        for (int k = 0; k < 3; k++) { r->J[k] = ang[k]; r->J[3 + k] = db[k]; }
        if (body > 0) {
            int L = (body - 1) / 3, depth = (body - 1) % 3;
            for (int j = 0; j <= depth; j++) {
                int b = 1 + 3 * L + j;
                real ax[3] = {C->Rw[b][0][M->jaxis[b]], C->Rw[b][1][M->jaxis[b]], C->Rw[b][2][M->jaxis[b]]};
                real rp[3] = {p[0] - C->ow[b][0], p[1] - C->ow[b][1], p[2] - C->ow[b][2]};
                real t[3]; v3cross(ax, rp, t);
                r->J[6 + 3 * L + j] = v3dot(d, t);
            }
        }
        minv_apply(M, C, r->J, r->W);
        real dd = 0; for (int k = 0; k < NV; k++) dd += r->J[k] * r->W[k];
        r->dinv = 1 / dd;
        real rel = 0; for (int k = 0; k < NV; k++) rel += r->J[k] * v[k];
        if (rr == 0) {
            /* btMultiBodyConstraintSolver::setupMultiBodyContactConstraint: penetration = distance + m_linearSlop */
            real pen = dist + (real)cfg->contact_slop, pos_err = 0, vel_err = -rel;
            if (pen > 0) vel_err -= pen / dt; else pos_err = -pen * cfg->contact_erp / dt;
            r->rhs = (pos_err + vel_err) * r->dinv;
            r->lo = 0; r->hi = 1e10; r->fric_of = -1;
            r->lam = warm * cfg->warmstart;
        } else {
            r->rhs = -rel * r->dinv;
            r->fric_of = nor_index; r->mu = mu;
        }
    }
}

void qso_physics_substep(const qso_config* cfg, qso_env* e, const real* tau, real g) {
    const qso_model* M = &e->model;
    qso_dyn* s = &e->s;
    phys_cache C;
    real dt = cfg->dt, cap = cfg->vel_cap;
    kinematics(M, s, &C);
    real acc[NV];
    aba(M, s, tau, g, &C, acc);

    /* generalized velocity in base coordinates, predicted */
    real v[NV];
    for (int k = 0; k < 6; k++) v[k] = C.v[0][k];
    for (int j = 0; j < NJ; j++) v[6 + j] = s->qd[j];
    {
        /* world-frame semi-implicit update: classical acceleration of the base origin = a_lin + w x v */
        real wxv[3]; v3cross(C.v[0], C.v[0] + 3, wxv);
        real al[3] = {acc[3] + wxv[0], acc[4] + wxv[1], acc[5] + wxv[2]};
        real aw[3], lw[3];
        m3v(C.R0, acc, aw); m3v(C.R0, al, lw);
        for (int k = 0; k < 3; k++) { s->vang[k] += dt * aw[k]; s->vlin[k] += dt * lw[k]; }
        for (int j = 0; j < NJ; j++) { s->qd[j] += dt * acc[6 + j]; clamp_vel(&s->qd[j], cap); }
        for (int k = 0; k < 3; k++) { clamp_vel(&s->vang[k], cap); clamp_vel(&s->vlin[k], cap); }
        m3tv(C.R0, s->vang, v); m3tv(C.R0, s->vlin, v + 3);
        for (int j = 0; j < NJ; j++) v[6 + j] = s->qd[j];
    }

    /* ---- constraint rows ---- */
    /* (capacity: mode 0 has at most 4 feet + 8 support points; the four-points-per-primitive experiment 4 + 4 (trunk) + 4 x (2 + 4 + 4)) */
    enum { MAXN = 48 };
    static _Thread_local row rows[24 + MAXN + 2 * MAXN]; int nlim = 0, nn = 0, nf = 0;
    row* lim = rows; row* nor = rows + 24; row* fr = rows + 24 + MAXN;
    int nor_foot[MAXN];    /* foot index of a normal row, -1 for the other links' support points */
    int nor_contact[MAXN]; /* contact-list entry that receives the row's force */
    int sup_of[MAXN];      /* mode 0 / 2: 8 x leg + candidate of a support point's normal row */
    int sel[4][5]; memset(sel, 0, sizeof(sel)); memset(sup_of, 0xff, sizeof(sup_of));
    e->n_contacts = 0;
    /* joint limits: a row exists only while the limit is violated (btMultiBodyJointLimitConstraint) */
    for (int j = 0; j < NJ; j++) {
        real lo = QSO_JOINT_LO[j % 3], hi = QSO_JOINT_HI[j % 3];
        for (int side = 0; side < 2; side++) {
            real pen = side == 0 ? s->q[j] - lo : hi - s->q[j];
            if (pen > 0) continue;
            row* r = &lim[nlim++];
            memset(r, 0, sizeof(*r));
            r->J[6 + j] = side == 0 ? 1 : -1;
            minv_apply(M, &C, r->J, r->W);
            real d = 0; for (int k = 0; k < NV; k++) d += r->J[k] * r->W[k];
            r->dinv = 1 / d;
            real rel = 0; for (int k = 0; k < NV; k++) rel += r->J[k] * v[k];
            r->rhs = (-pen * cfg->joint_erp / dt - rel) * r->dinv;
            r->lo = 0; r->hi = 1e10; r->fric_of = -1;
        }
    }
    int fixed0 = -1;   /* first of the six rows of the payload constraint in lim[] */
    real vblk[6] = {0, 0, 0, 0, 0, 0};
    if (cfg->payload_soft && e->m_pay > 0) {
        /* the block's own step: gravity (isotropic inertia: no gyroscopic term) */
        e->blk.v[2] -= dt * g;
        for (int k = 0; k < 3; k++) { vblk[k] = e->blk.w[k]; vblk[3 + k] = e->blk.v[k]; }
        real Rb[3][3]; qso_quat_to_mat(e->blk.quat, Rb);
        real nd[3] = {-e->r_pay[0], -e->r_pay[1], -e->r_pay[2]}, rB[3];
        m3v(Rb, nd, rB);                                             /* block centre -> its pivot, world */
        real perr[3]; for (int k = 0; k < 3; k++) perr[k] = s->pos[k] - (e->blk.pos[k] + rB[k]);    /* pivot A (base origin) - pivot B */
        e->blk.gap = sqrt(v3dot(perr, perr));
        /* orientation error: rotation vector of q_A q_B^-1 (world), small-angle */
        const real* qa = s->quat; const real* qb = e->blk.quat;
        real qe[4] = {qa[3] * -qb[0] + qa[0] * qb[3] + qa[1] * -qb[2] - qa[2] * -qb[1],
                      qa[3] * -qb[1] - qa[0] * -qb[2] + qa[1] * qb[3] + qa[2] * -qb[0],
                      qa[3] * -qb[2] + qa[0] * -qb[1] - qa[1] * -qb[0] + qa[2] * qb[3],
                      qa[3] * qb[3] - qa[0] * -qb[0] - qa[1] * -qb[1] - qa[2] * -qb[2]};
        real sg = qe[3] < 0 ? -2 : 2, aerr[3] = {sg * qe[0], sg * qe[1], sg * qe[2]};
        real mI = 1 / (e->m_pay * (real)(0.1 * 0.1 / 6.0)), mM = 1 / e->m_pay, bound = (real)500.0 * dt;
        fixed0 = nlim;
        for (int k = 0; k < 6; k++) {
            row* r = &lim[nlim++];
            memset(r, 0, sizeof(*r));
            real ax[3] = {k % 3 == 0, k % 3 == 1, k % 3 == 2}, ab[3];
            m3tv(C.R0, ax, ab);
            if (k < 3) {   /* linear row along world axis k at the pivots: robot side = base origin (no lever), block side lever rB */
                for (int i = 0; i < 3; i++) r->J[3 + i] = ab[i];
                real t[3]; v3cross(rB, ax, t);
                for (int i = 0; i < 3; i++) { r->J[18 + i] = -t[i]; r->J[21 + i] = -ax[i]; }
            } else {       /* angular row */
                for (int i = 0; i < 3; i++) { r->J[i] = ab[i]; r->J[18 + i] = -ax[i]; }
            }
            minv_apply(M, &C, r->J, r->W);
            for (int i = 0; i < 3; i++) { r->W[18 + i] = r->J[18 + i] * mI; r->W[21 + i] = r->J[21 + i] * mM; }
            real d = 0; for (int i = 0; i < NVX; i++) d += r->J[i] * r->W[i];
            r->dinv = 1 / d;
            real rel = 0; for (int i = 0; i < NV; i++) rel += r->J[i] * v[i];
            for (int i = 0; i < 6; i++) rel += r->J[18 + i] * vblk[i];
            real err = k < 3 ? perr[k] : aerr[k - 3];         /* d(err)/dt = J v for both kinds of rows */
            r->rhs = (-err * cfg->joint_erp / dt - rel) * r->dinv;
            r->lo = -bound; r->hi = bound; r->fric_of = -1;
        }
    }
    /* contacts, leg by leg: the foot, then (cfg->body_contacts) up to two more support points of the leg -- the lowest of
       {the trunk corner on the leg's side, hip housing, the two ends of the thigh box, knee end of the calf box} within contact range */
    for (int L = 0; L < 4; L++) {
        int ih = 1 + 3 * L, it = ih + 1, ic = ih + 2;
        real c3[3]; m3v(C.Rw[ic], FOOT_OFF, c3);
        real centre[3] = {C.ow[ic][0] + c3[0], C.ow[ic][1] + c3[1], C.ow[ic][2] + c3[2]};
        real dist = centre[2] - FOOT_R;
        e->foot_contact[L] = 0; e->foot_force[L] = 0;
        if (!(dist < THR_FOOT)) e->warm[L] = 0;
        else {
            e->foot_contact[L] = 1;
            real p[3] = {centre[0], centre[1], centre[2] - FOOT_R}; /* contact point on the sphere */
            contact_rows(cfg, M, &C, s, v, e->mu, ic, p, dist, e->warm[L], &nor[nn], &fr[nf], &fr[nf + 1], nn);
            nor_foot[nn] = L; nor_contact[nn] = e->n_contacts;
            push_contact(e, 1, 0, 5 + 4 * L, -1, dist);
            nn++; nf += 2;
        }
        if (!cfg->body_contacts) continue;
        if (e->manifold_mode == 1) {
            /* experiment: up to four points per primitive -- the vertices (for the hip cylinder: the lowest point of each end's rim) inside the
               primitive's contact range, lowest first */
            for (int prim = (L == 0 ? 0 : 1); prim < 4; prim++) {
                real pts[8][3]; int np = 0, body, link; real thr;
                if (prim == 0) {   /* trunk box (once) */
                    body = 0; link = 0; thr = THR_TRUNK;
                    for (int a = 0; a < 8; a++) {
                        real pb[3] = {(a & 1 ? 1 : -1) * TRUNK_HALF[0], (a & 2 ? 1 : -1) * TRUNK_HALF[1], (a & 4 ? 1 : -1) * TRUNK_HALF[2]}, pw[3];
                        m3v(C.R0, pb, pw);
                        for (int k = 0; k < 3; k++) pts[np][k] = s->pos[k] + pw[k];
                        np++;
                    }
                } else if (prim == 1) {   /* hip cylinder: both ends */
                    body = ih; link = 2 + 4 * L; thr = THR_HIP;
                    real a[3] = {C.Rw[ih][0][1], C.Rw[ih][1][1], C.Rw[ih][2][1]};
                    real az = a[2], s2 = 1 - az * az; if (s2 < (real)1e-12) s2 = (real)1e-12;
                    real inv = HIP_CYL_R / sqrt(s2);
                    for (int end = -1; end <= 1; end += 2) {
                        for (int k = 0; k < 3; k++) pts[np][k] = C.ow[ih][k] + end * HIP_CYL_H * a[k] - ((k == 2 ? 1 : 0) - az * a[k]) * inv;
                        np++;
                    }
                } else {   /* thigh / calf box */
                    int b = prim == 2 ? it : ic;
                    const real* half = prim == 2 ? THIGH_HALF : CALF_HALF;
                    body = b; link = (prim == 2 ? 3 : 4) + 4 * L; thr = prim == 2 ? THR_THIGH : THR_CALF;
                    for (int a = 0; a < 8; a++) {
                        real pl[3] = {LINK_BOX_C[0] + (a & 1 ? 1 : -1) * half[0], LINK_BOX_C[1] + (a & 2 ? 1 : -1) * half[1], LINK_BOX_C[2] + (a & 4 ? 1 : -1) * half[2]}, pw[3];
                        if (prim == 3 && (a & 4) == 0) continue;   /* the calf box's foot end belongs to the foot sphere, as in mode 0 */
                        m3v(C.Rw[b], pl, pw);
                        for (int k = 0; k < 3; k++) pts[np][k] = C.ow[b][k] + pw[k];
                        np++;
                    }
                }
                int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int slot = 0; slot < 4 && nn < MAXN; slot++) {
                    int bi = -1;
                    for (int i = 0; i < np; i++) if (!used[i] && pts[i][2] < thr && (bi < 0 || pts[i][2] < pts[bi][2])) bi = i;
                    if (bi < 0) break;
                    used[bi] = 1;
                    contact_rows(cfg, M, &C, s, v, e->mu, body, pts[bi], pts[bi][2], 0, &nor[nn], &fr[nf], &fr[nf + 1], nn);
                    nor_foot[nn] = -1; nor_contact[nn] = -1 - link;
                    nn++; nf += 2;
                }
            }
            continue;
        }
        real fx = (L < 2) ? 1 : -1, sy = (L & 1) ? 1 : -1;
        real cand[5][3], ch[5]; int cbody[5] = {0, ih, it, it, ic}, clink[5] = {0, 2 + 4 * L, 3 + 4 * L, 3 + 4 * L, 4 + 4 * L};
        real cthr[5] = {THR_TRUNK, THR_HIP, THR_THIGH, THR_THIGH, THR_CALF};
        {   /* 0: trunk corner */
            real zsign = C.R0[2][2] < 0 ? 1 : -1;   /* the vertex on the side the plane normal points away from */
            real pb[3] = {fx * TRUNK_HALF[0], sy * TRUNK_HALF[1], zsign * TRUNK_HALF[2]}, pw[3];
            m3v(C.R0, pb, pw);
            for (int k = 0; k < 3; k++) cand[0][k] = s->pos[k] + pw[k];
        }
        {   /* 1: hip cylinder (axis = link y), lowest point of its rim */
            real a[3] = {C.Rw[ih][0][1], C.Rw[ih][1][1], C.Rw[ih][2][1]};
            real az = a[2], s2 = 1 - az * az; if (s2 < (real)1e-12) s2 = (real)1e-12;
            real sg = az < 0 ? -1 : 1, inv = HIP_CYL_R / sqrt(s2);
            for (int k = 0; k < 3; k++) cand[1][k] = C.ow[ih][k] - sg * HIP_CYL_H * a[k] - ((k == 2 ? 1 : 0) - az * a[k]) * inv;
        }
        {   /* 2, 3: lowest vertex of the thigh box at its hip end and at its knee end; 4: of the calf box at its knee end */
            real X[3] = {C.Rw[it][0][0], C.Rw[it][1][0], C.Rw[it][2][0]}, Y[3] = {C.Rw[it][0][1], C.Rw[it][1][1], C.Rw[it][2][1]};
            real sx = X[2] < 0 ? -1 : 1, sgy = Y[2] < 0 ? -1 : 1;
            for (int k = 0; k < 3; k++) {
                real off = sx * THIGH_HALF[0] * X[k] + sgy * THIGH_HALF[1] * Y[k];
                cand[2][k] = C.ow[it][k] - off; cand[3][k] = C.ow[ic][k] - off;
            }
            real X3[3] = {C.Rw[ic][0][0], C.Rw[ic][1][0], C.Rw[ic][2][0]};
            real s3 = X3[2] < 0 ? -1 : 1;
            for (int k = 0; k < 3; k++) cand[4][k] = C.ow[ic][k] - s3 * CALF_HALF[0] * X3[k] - sgy * CALF_HALF[1] * Y[k];
        }
        for (int i = 0; i < 5; i++) ch[i] = cand[i][2] < cthr[i] ? cand[i][2] : (real)1e9;
        /* how many: a leg carries at most three contact points -- its foot and two support points, or, with the foot off the ground, three
           support points (round 6; rounds 2-5: two whatever the foot did.  A robot on its back rests on trunk corner + both ends of each
           thigh box: with two of the three the choice flipped from substep to substep, the robot crept at 7 mm/s and lay 1-2 mm off where
           four points per primitive (mode 1) put it; with three: 3e-7 m, at rest.  Mode 3 keeps the old cap for the comparison) */
        const int n_slots = (e->manifold_mode == 3 || e->foot_contact[L]) ? 2 : 3;
        for (int slot = 0; slot < n_slots; slot++) {
            int bi = 0;
            for (int i = 1; i < 5; i++) if (ch[i] < ch[bi]) bi = i;
            if (!(ch[bi] < (real)1e8)) break;
            contact_rows(cfg, M, &C, s, v, e->mu, cbody[bi], cand[bi], ch[bi], e->manifold_mode == 2 ? e->warm_sup[L][bi] : 0, &nor[nn], &fr[nf], &fr[nf + 1], nn);
            nor_foot[nn] = -1; nor_contact[nn] = -1 - clink[bi];
            sup_of[nn] = 8 * L + bi; sel[L][bi] = 1;
            ch[bi] = (real)1e9;
            nn++; nf += 2;
        }
        for (int i = 0; i < 5; i++) if (!sel[L][i]) e->warm_sup[L][i] = 0;     /* (a candidate that is not selected has left the manifold) */
    }
    e->n_invalid = collect_invalid(cfg, e, &C, s);
    /* ---- projected Gauss-Seidel in velocity space (btMultiBodyConstraintSolver::solveSingleIteration order) ---- */
    real dv[NVX]; memset(dv, 0, sizeof(dv));
    for (int i = 0; i < nn; i++)
        if (nor[i].lam != 0) for (int k = 0; k < NVX; k++) dv[k] += nor[i].W[k] * nor[i].lam;
    for (int it = 0; it < cfg->solver_iters; it++) {
        real maxres2 = 0; /* btMultiBodyConstraintSolver: leastSquaredResidual = max over rows of (deltaImpulse / jacDiagABInv)^2 */
        for (int jj = 0; jj < nlim + nn + nf; jj++) {
            row* r;
            if (jj < nlim) r = &lim[(it & 1) ? jj : nlim - 1 - jj];
            else if (jj < nlim + nn) r = &nor[jj - nlim];
            else r = &fr[jj - nlim - nn];
            if (r->fric_of >= 0 && cfg->friction_cone) {
                /* btMultiBodyConstraintSolver::resolveConeFrictionConstraintRows (SOLVER_USE_2_FRICTION_DIRECTIONS without
                   SOLVER_DISABLE_IMPLICIT_CONE_FRICTION): both rows of the contact from the same velocities, the summed impulse
                   scaled back onto the disc of radius mu x normal impulse (its atan2 / sin / cos form is this radial projection) */
                row* a = r; row* b = &fr[jj + 1 - nlim - nn];
                jj++;
                real lim = a->mu * nor[a->fric_of].lam;
                real ja = 0, jb = 0;
                for (int k = 0; k < NVX; k++) { ja += a->J[k] * dv[k]; jb += b->J[k] * dv[k]; }
                real sa = a->lam + (a->rhs - ja * a->dinv), sb = b->lam + (b->rhs - jb * b->dinv);
                real r2 = sa * sa + sb * sb;
                if (r2 > lim * lim) { real sc = lim / sqrt(r2); sa *= sc; sb *= sc; }
                real da = sa - a->lam, db = sb - b->lam;
                a->lam = sa; b->lam = sb;
                for (int k = 0; k < NVX; k++) dv[k] += a->W[k] * da + b->W[k] * db;
                real ra = da / a->dinv, rb = db / b->dinv;
                if (ra * ra > maxres2) maxres2 = ra * ra;
                if (rb * rb > maxres2) maxres2 = rb * rb;
                continue;
            }
            if (r->fric_of >= 0) {
                real tot = nor[r->fric_of].lam;
                if (!(tot > 0)) continue;
                r->lo = -r->mu * tot; r->hi = r->mu * tot;
            }
            real jdv = 0; for (int k = 0; k < NVX; k++) jdv += r->J[k] * dv[k];
            real dl = r->rhs - jdv * r->dinv;
            real sum = r->lam + dl;
            if (sum < r->lo) { dl = r->lo - r->lam; sum = r->lo; }
            else if (sum > r->hi) { dl = r->hi - r->lam; sum = r->hi; }
            r->lam = sum;
            for (int k = 0; k < NVX; k++) dv[k] += r->W[k] * dl;
            real resid = dl / r->dinv;
            if (resid * resid > maxres2) maxres2 = resid * resid;
        }
        /* solverResidualThreshold (PyBullet's default 1e-7 is this build's default too; 0: only an exactly stationary sweep ends
           early, which cannot change the result) */
        if (maxres2 <= (real)cfg->solver_residual_threshold) break;
    }
    for (int i = 0; i < nn; i++) {
        if (nor_foot[i] >= 0) {
            e->foot_force[nor_foot[i]] = nor[i].lam / dt;
            e->warm[nor_foot[i]] = nor[i].lam;
            e->contacts[nor_contact[i]].force = nor[i].lam / dt;
        } else {   /* support point of another link: its force goes to that link's entry of the contact list */
            if (sup_of[i] >= 0) e->warm_sup[sup_of[i] / 8][sup_of[i] % 8] = nor[i].lam;
            for (int c = 0; c < e->n_contacts; c++)
                if (e->contacts[c].body_a == 1 && e->contacts[c].body_b == 0 && e->contacts[c].link_a == -1 - nor_contact[i]) { e->contacts[c].force += nor[i].lam / dt; break; }
        }
    }
    if (fixed0 >= 0) {
        for (int k = 0; k < 6; k++) e->blk.lam[k] = lim[fixed0 + k].lam;
        for (int k = 0; k < 3; k++) { e->blk.w[k] += dv[18 + k]; e->blk.v[k] += dv[21 + k]; }
        for (int k = 0; k < 3; k++) e->blk.pos[k] += dt * e->blk.v[k];
        real w[3] = {e->blk.w[0], e->blk.w[1], e->blk.w[2]};
        real th = sqrt(v3dot(w, w)) * dt, sc = th < 1e-6 ? (real)0.5 * dt : sin((real)0.5 * th) / (th / dt);
        real dq[4] = {w[0] * sc, w[1] * sc, w[2] * sc, cos((real)0.5 * th)};
        const real* q = e->blk.quat;
        real nq[4] = {dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1], dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0],
                      dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3], dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2]};
        real n = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
        for (int k = 0; k < 4; k++) e->blk.quat[k] = nq[k] / n;
    }
    /* apply constraint impulses, clamp, integrate positions */
    {
        real dw[3], dl[3];
        m3v(C.R0, dv, dw); m3v(C.R0, dv + 3, dl);
        for (int k = 0; k < 3; k++) {
            s->vang[k] += dw[k]; s->vlin[k] += dl[k];
            clamp_vel(&s->vang[k], cap); clamp_vel(&s->vlin[k], cap);
        }
        for (int j = 0; j < NJ; j++) { s->qd[j] += dv[6 + j]; clamp_vel(&s->qd[j], cap); }
    }
    for (int k = 0; k < 3; k++) s->pos[k] += dt * s->vlin[k];
    {
        real w[3] = {s->vang[0], s->vang[1], s->vang[2]};
        real th = sqrt(v3dot(w, w)) * dt, sc;
        if (th < 1e-6) sc = (real)0.5 * dt * (1 - th * th / 24); else sc = sin((real)0.5 * th) / (th / dt);
        real dq[4] = {w[0] * sc, w[1] * sc, w[2] * sc, cos((real)0.5 * th)};
        const real* q = s->quat;
        real nq[4] = {
            dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1],
            dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0],
            dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3],
            dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2]};
        real n = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
        for (int k = 0; k < 4; k++) s->quat[k] = nq[k] / n;
    }
    for (int j = 0; j < NJ; j++) s->q[j] += dt * s->qd[j];
}

/* ------------------------------------------------------------------ independent CRBA + RNEA (KAT K3) */
static int phys_crba_rnea(const qso_model* M, const qso_dyn* s, real g, real* H, real* Cb) {
    phys_cache C; kinematics(M, s, &C);
    /* RNEA with qdd = 0 and base acceleration 0, gravity as a field */
    real a[NB][6], f[NB][6];
    real gw[3] = {0, 0, -g}, gb[3];
    m3tv(C.R0, gw, gb);
    for (int k = 0; k < 6; k++) a[0][k] = 0;
    for (int k = 0; k < 3; k++) a[0][3 + k] = -gb[k];
    for (int i = 0; i < NB; i++) {
        if (i > 0) {
            int p = M->parent[i];
            real vJ[6] = {0, 0, 0, 0, 0, 0}, cc[6];
            vJ[M->jaxis[i]] = s->qd[i - 1];
            xm(&C.X[i], a[p], a[i]);
            crm(C.v[i], vJ, cc);
            for (int k = 0; k < 6; k++) a[i][k] += cc[k];
        }
        real Ia[6], Iv[6], vIv[6];
        m6v(M->I6[i], a[i], Ia); m6v(M->I6[i], C.v[i], Iv); crf(C.v[i], Iv, vIv);
        for (int k = 0; k < 6; k++) f[i][k] = Ia[k] + vIv[k];
    }
    for (int i = NB - 1; i >= 1; i--) {
        Cb[6 + i - 1] = f[i][M->jaxis[i]];
        real pp[6]; xft(&C.X[i], f[i], pp);
        for (int k = 0; k < 6; k++) f[M->parent[i]][k] += pp[k];
    }
    for (int k = 0; k < 6; k++) Cb[k] = f[0][k];
    /* CRBA */
    real Ic[NB][6][6];
    memcpy(Ic, M->I6, sizeof(Ic));
    for (int i = NB - 1; i >= 1; i--) {
        real X6[6][6], T[6][6]; x6(&C.X[i], X6);
        for (int r = 0; r < 6; r++)
            for (int q = 0; q < 6; q++) { real t = 0; for (int k = 0; k < 6; k++) t += Ic[i][r][k] * X6[k][q]; T[r][q] = t; }
        for (int r = 0; r < 6; r++)
            for (int q = 0; q < 6; q++) { real t = 0; for (int k = 0; k < 6; k++) t += X6[k][r] * T[k][q]; Ic[M->parent[i]][r][q] += t; }
    }
    memset(H, 0, NV * NV * sizeof(real));
    for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) H[r * NV + q] = Ic[0][r][q];
    for (int i = 1; i < NB; i++) {
        real F[6];
        for (int k = 0; k < 6; k++) F[k] = Ic[i][k][M->jaxis[i]];
        H[(6 + i - 1) * NV + 6 + i - 1] = F[M->jaxis[i]];
        int j = i;
        while (M->parent[j] > 0) {
            real Fp[6]; xft(&C.X[j], F, Fp); memcpy(F, Fp, sizeof(Fp));
            j = M->parent[j];
            real h = F[M->jaxis[j]];
            H[(6 + i - 1) * NV + 6 + j - 1] = h; H[(6 + j - 1) * NV + 6 + i - 1] = h;
        }
        real Fp[6]; xft(&C.X[j], F, Fp);
        for (int k = 0; k < 6; k++) { H[k * NV + 6 + i - 1] = Fp[k]; H[(6 + i - 1) * NV + k] = Fp[k]; }
    }
    return 0;
}

int qso_phys_crba_rnea(qso_handle* h, int env, real* H, real* C) {
    return phys_crba_rnea(&h->env[env].model, &h->env[env].s, h->gravity, H, C);
}
int qso_phys_aba(qso_handle* h, int env, const real* tau, real* acc) {
    phys_cache C; kinematics(&h->env[env].model, &h->env[env].s, &C);
    return aba(&h->env[env].model, &h->env[env].s, tau, h->gravity, &C, acc);
}
int qso_phys_step(qso_handle* h, int env, const real* tau) {
    qso_physics_substep(&h->cfg, &h->env[env], tau, h->gravity);
    return 0;
}
int qso_phys_set_gravity(qso_handle* h, real g) { h->gravity = g; return 0; }
int qso_phys_set_manifold(qso_handle* h, int mode) {
    if (mode < 0 || mode > 3) return -1;
    for (int i = 0; i < h->cfg.n_envs; i++) h->env[i].manifold_mode = mode;
    return 0;
}
int qso_get_block(qso_handle* h, real* out /*[N,20]: pos3 quat4 v3 w3 lam6 gap*/) {
    for (int i = 0; i < h->cfg.n_envs; i++) {
        const qso_env* e = &h->env[i]; real* o = out + 20 * i;
        memcpy(o, e->blk.pos, 3 * sizeof(real)); memcpy(o + 3, e->blk.quat, 4 * sizeof(real)); memcpy(o + 7, e->blk.v, 3 * sizeof(real));
        memcpy(o + 10, e->blk.w, 3 * sizeof(real)); memcpy(o + 13, e->blk.lam, 6 * sizeof(real)); o[19] = e->blk.gap;
    }
    return 0;
}
int qso_get_contacts(qso_handle* h, int env, int32_t* ids /*[max][4]*/, real* dist_force /*[max][2]*/, int max) {
    const qso_env* e = &h->env[env];
    int n = e->n_contacts < max ? e->n_contacts : max;
    for (int i = 0; i < n; i++) {
        ids[4 * i] = e->contacts[i].body_a; ids[4 * i + 1] = e->contacts[i].body_b; ids[4 * i + 2] = e->contacts[i].link_a; ids[4 * i + 3] = e->contacts[i].link_b;
        dist_force[2 * i] = e->contacts[i].dist; dist_force[2 * i + 1] = e->contacts[i].force;
    }
    return e->n_contacts;
}
/* geometry hooks for the known-answer tests: box / box overlap by edge clipping, sphere / box distance */
int qso_geom_boxes_overlap(const real* ca, const real* Ra /*row-major, columns = axes*/, const real* ha, const real* cb, const real* Rb, const real* hb) {
    obox A, B;
    memcpy(A.c, ca, sizeof(A.c)); memcpy(A.R, Ra, sizeof(A.R)); memcpy(A.h, ha, sizeof(A.h));
    memcpy(B.c, cb, sizeof(B.c)); memcpy(B.R, Rb, sizeof(B.R)); memcpy(B.h, hb, sizeof(B.h));
    return boxes_overlap(&A, &B);
}

int qso_phys_energy(qso_handle* h, int env, real* out) {
    const qso_model* M = &h->env[env].model; const qso_dyn* s = &h->env[env].s;
    phys_cache C; kinematics(M, s, &C);
    real KE = 0, PE = 0, p[3] = {0, 0, 0}, Lw[3] = {0, 0, 0}, com[3] = {0, 0, 0};
    for (int i = 0; i < NB; i++) {
        real Iv[6]; m6v(M->I6[i], C.v[i], Iv);
        KE += (real)0.5 * dot6(C.v[i], Iv);
        real cw[3]; m3v(C.Rw[i], M->com[i], cw);
        for (int k = 0; k < 3; k++) cw[k] += C.ow[i][k];
        PE += M->mass[i] * h->gravity * cw[2];
        /* com velocity in world: R (v + w x c) */
        real wxc[3]; v3cross(C.v[i], M->com[i], wxc);
        real vc[3] = {C.v[i][3] + wxc[0], C.v[i][4] + wxc[1], C.v[i][5] + wxc[2]}, vw[3];
        m3v(C.Rw[i], vc, vw);
        real Iw[3], Lb[3]; m3v(M->Ic[i], C.v[i], Iw); m3v(C.Rw[i], Iw, Lb);
        real cxv[3]; v3cross(cw, vw, cxv);
        for (int k = 0; k < 3; k++) {
            p[k] += M->mass[i] * vw[k];
            Lw[k] += Lb[k] + M->mass[i] * cxv[k];
            com[k] += M->mass[i] * cw[k] / M->total_mass;
        }
    }
    out[0] = KE; out[1] = PE;
    for (int k = 0; k < 3; k++) { out[2 + k] = p[k]; out[5 + k] = Lw[k]; out[8 + k] = com[k]; }
    return 0;
}
