/* qso_model.c -- ORACLE (test infrastructure). Go1 rigid-body tables restated from
 * quadruped_spring/go1/go1_description/urdf/go1.urdf (line numbers below refer to that file).
 * PyBullet keeps the 6 fixed joints as separate links; dynamically a fixed joint is a rigid weld, so the
 * oracle merges base+trunk+imu (and the optional payload block, quadruped.py:778-819) into body 0 and
 * calf+foot into the calf body.  Leg order FR, FL, RR, RL = PyBullet motor order (quadruped.py:586-596). */
#include "qso_internal.h"

/* trunk :80-85 */
const real QSO_M_TRUNK = 5.204;
static const real TRUNK_COM[3] = {0.0223, 0.0, -0.0005};
const real QSO_TRUNK_I[6] = {0.0168352186, 0.0004636141, 0.0002367952, 0.0656071082, 3.6671e-05, 0.0742720659};
/* base :55-59, imu :87-97 */
static const real BASE_M = 0.00001, BASE_I = 1e-5;
static const real IMU_M = 0.001, IMU_I = 0.0001;
static const real IMU_POS[3] = {-0.01592, -0.06659, -0.00617};
/* hip :134-136 (FR), mirrored :294-296 :454-456 :614-616 */
const real QSO_M_LEG[3] = {0.591, 0.92, 0.131};
static const real HIP_COM[3] = {0.00541, 0.00074, 6e-06};
const real QSO_HIP_I[6] = {0.000374268192, 3.6844422e-05, 9.86754e-07, 0.000635923669, 1.172894e-06, 0.000457647394};
/* thigh :186-188 (FR/RR), :346-348 (FL/RL) */
static const real THIGH_COM[3] = {-0.003468, 0.018947, -0.032736};
const real QSO_THIGH_I[6] = {0.005851561134, 1.783284e-06, 0.000328291374, 0.005596155105, 2.1430713e-05, 0.00107157026};
/* calf :212-216, identical on all legs */
static const real CALF_COM[3] = {0.006286, 0.001307, -0.122269};
const real QSO_CALF_I[6] = {0.002939186297, 1.440899e-06, -0.00010535955, 0.00295576935, -2.4397752e-05, 3.0273372e-05};
/* foot :218-240 */
static const real FOOT_M = 0.06, FOOT_I = 9.6e-06;
static const real FOOT_POS[3] = {0, 0, -0.213};
/* joint origins :113 :165 :192 and mirrors */
static const real HIP_X = 0.1881, HIP_Y = 0.04675, THIGH_Y = 0.08, CALF_Z = -0.213;
/* URDF joint limits :117 :169 :196 */
const real QSO_JOINT_LO[3] = {-1.0471975512, -0.663225115758, -2.72271363311};
const real QSO_JOINT_HI[3] = {1.0471975512, 2.96705972839, -0.837758040957};

static void sym6_to_mat(const real* s, real I[3][3]) {
    I[0][0] = s[0]; I[0][1] = I[1][0] = s[1]; I[0][2] = I[2][0] = s[2];
    I[1][1] = s[3]; I[1][2] = I[2][1] = s[4]; I[2][2] = s[5];
}

/* accumulate a rigid part (m, c, Ic) into composite sums */
typedef struct { real m; real mc[3]; real Io[3][3]; } accum; /* Io = inertia about link origin */
static void acc_add(accum* a, real m, const real* c, const real Ic[3][3]) {
    a->m += m;
    for (int i = 0; i < 3; i++) a->mc[i] += m * c[i];
    real cc = v3dot(c, c);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) a->Io[i][j] += Ic[i][j] + m * ((i == j ? cc : 0) - c[i] * c[j]);
}
static void acc_finish(const accum* a, real* mass, real* com, real Ic[3][3]) {
    *mass = a->m;
    for (int i = 0; i < 3; i++) com[i] = a->mc[i] / a->m;
    real cc = v3dot(com, com);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ic[i][j] = a->Io[i][j] - a->m * ((i == j ? cc : 0) - com[i] * com[j]);
}

static void spatial_inertia(real m, const real* c, const real Ic[3][3], real I6[6][6]) {
    real cc = v3dot(c, c);
    real cx[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            I6[i][j] = Ic[i][j] + m * ((i == j ? cc : 0) - c[i] * c[j]);
            I6[i][j + 3] = m * cx[i][j];
            I6[i + 3][j] = -m * cx[i][j];
            I6[i + 3][j + 3] = (i == j) ? m : 0;
        }
}

/* Mass randomisation (env_randomizer.py:56-83): link masses change; a link of mass m gets the inertia m x unit[link], the table
 * of inertia per unit mass that the host fills by the chosen rule (config.py unit_inertia_table: the URDF tensor scaled with the
 * mass, or what Bullet's changeDynamics(mass=...) leaves: the box inertia of the collision compound's AABB in the principal frame).
 * unit = float[4][6] for hip, thigh, calf, trunk: magnitudes in the FR-leg convention of the tables above. */
void qso_model_build(qso_model* M, const float (*unit)[6], real m_trunk, const real* m_leg3, real m_pay, const real* r_pay) {
    memset(M, 0, sizeof(*M));
    real Idiag[3][3];
    /* body 0 */
    accum a; memset(&a, 0, sizeof(a));
    real z3[3] = {0, 0, 0};
    memset(Idiag, 0, sizeof(Idiag)); Idiag[0][0] = Idiag[1][1] = Idiag[2][2] = BASE_I;
    acc_add(&a, BASE_M, z3, Idiag);
    real ut[6]; for (int i = 0; i < 6; i++) ut[i] = (real)unit[3][i];
    real It[3][3]; sym6_to_mat(ut, It);
    real sc = m_trunk;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) It[i][j] *= sc;
    acc_add(&a, m_trunk, TRUNK_COM, It);
    memset(Idiag, 0, sizeof(Idiag)); Idiag[0][0] = Idiag[1][1] = Idiag[2][2] = IMU_I;
    acc_add(&a, IMU_M, IMU_POS, Idiag);
    if (m_pay > 0) {
        /* payload: cube, half extent 0.05 (quadruped.py:793), solid-box inertia m/6*a^2 with a = 0.1 */
        memset(Idiag, 0, sizeof(Idiag)); Idiag[0][0] = Idiag[1][1] = Idiag[2][2] = m_pay * (real)(0.1 * 0.1 / 6.0);
        acc_add(&a, m_pay, r_pay, Idiag);
    }
    acc_finish(&a, &M->mass[0], M->com[0], M->Ic[0]);
    M->parent[0] = -1;
    for (int L = 0; L < 4; L++) {
        real fx = (L < 2) ? 1 : -1;        /* front +, rear - */
        real sy = (L & 1) ? 1 : -1;        /* right -, left + */
        int ih = 1 + 3 * L, it = ih + 1, ic = ih + 2;
        /* hip */
        real hc[3] = {-fx * HIP_COM[0], -sy * HIP_COM[1], HIP_COM[2]};
        const float* uh = unit[0];
        real hs[6] = {uh[0], fx * sy * uh[1], -fx * uh[2], uh[3], -sy * uh[4], uh[5]};
        real Ih[3][3]; sym6_to_mat(hs, Ih);
        sc = m_leg3[0];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M->Ic[ih][i][j] = Ih[i][j] * sc;
        M->mass[ih] = m_leg3[0]; memcpy(M->com[ih], hc, sizeof(hc));
        M->jpos[ih][0] = fx * HIP_X; M->jpos[ih][1] = sy * HIP_Y; M->jpos[ih][2] = 0;
        M->jaxis[ih] = 0; M->parent[ih] = 0;
        /* thigh */
        real tc[3] = {THIGH_COM[0], -sy * THIGH_COM[1], THIGH_COM[2]};
        const float* uth = unit[1];
        real ts[6] = {uth[0], sy * uth[1], uth[2], uth[3], sy * uth[4], uth[5]};
        real Ith[3][3]; sym6_to_mat(ts, Ith);
        sc = m_leg3[1];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M->Ic[it][i][j] = Ith[i][j] * sc;
        M->mass[it] = m_leg3[1]; memcpy(M->com[it], tc, sizeof(tc));
        M->jpos[it][0] = 0; M->jpos[it][1] = sy * THIGH_Y; M->jpos[it][2] = 0;
        M->jaxis[it] = 1; M->parent[it] = ih;
        /* calf + foot */
        memset(&a, 0, sizeof(a));
        real uc[6]; for (int i = 0; i < 6; i++) uc[i] = (real)unit[2][i];
        real Icf[3][3]; sym6_to_mat(uc, Icf);
        sc = m_leg3[2];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Icf[i][j] *= sc;
        acc_add(&a, m_leg3[2], CALF_COM, Icf);
        memset(Idiag, 0, sizeof(Idiag)); Idiag[0][0] = Idiag[1][1] = Idiag[2][2] = FOOT_I;
        acc_add(&a, FOOT_M, FOOT_POS, Idiag);
        acc_finish(&a, &M->mass[ic], M->com[ic], M->Ic[ic]);
        M->jpos[ic][0] = 0; M->jpos[ic][1] = 0; M->jpos[ic][2] = CALF_Z;
        M->jaxis[ic] = 1; M->parent[ic] = it;
    }
    M->total_mass = 0;
    for (int i = 0; i < NB; i++) {
        spatial_inertia(M->mass[i], M->com[i], M->Ic[i], M->I6[i]);
        M->total_mass += M->mass[i];
    }
}
