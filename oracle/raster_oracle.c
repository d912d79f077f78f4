/* raster_oracle.c — plain-C restatement of this repo's raw_pixels renderer, for parity tests of the
 * HIP tile rasteriser.  TEST INFRASTRUCTURE ONLY.
 *
 * What is restated from the reference: the camera calls of KukaButtonGymEnv.render /
 * MobileRobotGymEnv.render (/root/reference/environments/kuka_gym/kuka_button_gym_env.py:370-420 and
 * the mobile_robot_env.py:282-334 counterpart: computeViewMatrixFromYawPitchRoll(target, dist, yaw,
 * pitch, roll, upAxisIndex=2), computeProjectionMatrixFOV(60, 1, 0.1, 100), 224x224 RGB, row 0 on top),
 * the object poses (reset() of each env) and the colours (the urdf files, changeVisualShape calls).
 * What cannot be restated: TinyRenderer itself and the pybullet_data meshes (absent); scene = analytic
 * primitives, Lambert + ambient, no shadows.  Pinning: the camera model and the scene geometry are checked
 * against measurements on the reference's own rendered frames (imgs/kuka.gif, imgs/mobile_robot.gif ->
 * tests/golden/render_reference_measurements.json, tests/test_raster_reference_pin.py: button / table /
 * checker / walls project within 2 px at 168x168, flat colours within 16/255); per-pixel PARITY with
 * TinyRenderer's shading and meshes is UNPINNED.
 * float32 per pixel, compile with -ffp-contract=off. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "kuka_model.h"

void kuka_oracle_fk(const double *q, double *R63, double *p21);     /* (writes 9 / 3 doubles per link of the model in use: 7 or 12 links) */
int kuka_oracle_get_full(void);

enum { P_PLANE = 0, P_BOX = 1, P_CYL = 2, P_CAPSULE = 3 };
typedef struct { int type; float col[3]; float a[3]; float b[3]; float rad, cs, sn; } prim_t;
typedef struct { float eye[3], f[3], r[3], u[3], tanh; } cam_t;

static prim_t mk(int type, float cr, float cg, float cb, float ax, float ay, float az, float bx, float by, float bz,
                 float rad, float cs, float sn) {
    prim_t p; p.type = type; p.col[0] = cr; p.col[1] = cg; p.col[2] = cb; p.a[0] = ax; p.a[1] = ay; p.a[2] = az;
    p.b[0] = bx; p.b[1] = by; p.b[2] = bz; p.rad = rad; p.cs = cs; p.sn = sn; return p;
}

static cam_t camera(const double target[3], double dist, double yaw, double pitch, double roll, double fov) {
    const double d2r = KM_PI / 180.0;
    double cy = cos(yaw * d2r), sy = sin(yaw * d2r), cp = cos(pitch * d2r), sp = sin(pitch * d2r), cr = cos(roll * d2r), sr = sin(roll * d2r);
    double Rm[3][3] = {{cy * cr, cy * sr * sp - sy * cp, cy * sr * cp + sy * sp},
                       {sy * cr, sy * sr * sp + cy * cp, sy * sr * cp - cy * sp},
                       {-sr, cr * sp, cr * cp}};
    double eye[3], up[3], f[3], r[3], u[3], nf, nr; int k; cam_t c;
    for (k = 0; k < 3; k++) { eye[k] = target[k] + Rm[k][1] * (-dist); up[k] = Rm[k][2]; f[k] = target[k] - eye[k]; }
    nf = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (k = 0; k < 3; k++) f[k] /= nf;
    r[0] = f[1] * up[2] - f[2] * up[1]; r[1] = f[2] * up[0] - f[0] * up[2]; r[2] = f[0] * up[1] - f[1] * up[0];
    nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (k = 0; k < 3; k++) r[k] /= nr;
    u[0] = r[1] * f[2] - r[2] * f[1]; u[1] = r[2] * f[0] - r[0] * f[2]; u[2] = r[0] * f[1] - r[1] * f[0];
    for (k = 0; k < 3; k++) { c.eye[k] = (float)eye[k]; c.f[k] = (float)f[k]; c.r[k] = (float)r[k]; c.u[k] = (float)u[k]; }
    c.tanh = (float)tan(fov * d2r / 2);
    return c;
}

static float hit(const prim_t *p, const float o[3], const float d[3], float n[3]) {
    if (p->type == P_PLANE) {
        float t;
        if (d[2] == 0.0f) return -1.0f;
        t = (p->a[2] - o[2]) / d[2];
        n[0] = 0; n[1] = 0; n[2] = 1;
        return t > 0.0f ? t : -1.0f;
    }
    if (p->type == P_BOX) {
        float px = o[0] - p->a[0], py = o[1] - p->a[1], pz = o[2] - p->a[2];
        float lo[3], ld[3], tmin = -3.0e38f, tmax = 3.0e38f, sign = 0.0f, lnx, lny; int axis = 0, k;
        lo[0] = p->cs * px + p->sn * py; lo[1] = p->cs * py - p->sn * px; lo[2] = pz;
        ld[0] = p->cs * d[0] + p->sn * d[1]; ld[1] = p->cs * d[1] - p->sn * d[0]; ld[2] = d[2];
        for (k = 0; k < 3; k++) {
            if (ld[k] == 0.0f) { if (lo[k] < -p->b[k] || lo[k] > p->b[k]) return -1.0f; }
            else {
                float inv = 1.0f / ld[k], t0 = (-p->b[k] - lo[k]) * inv, t1 = (p->b[k] - lo[k]) * inv, s = -1.0f;
                if (t0 > t1) { float tt = t0; t0 = t1; t1 = tt; s = 1.0f; }
                if (t0 > tmin) { tmin = t0; axis = k; sign = s; }
                if (t1 < tmax) tmax = t1;
            }
        }
        if (tmin > tmax || tmin <= 0.0f) return -1.0f;
        lnx = axis == 0 ? sign : 0.0f; lny = axis == 1 ? sign : 0.0f;
        n[0] = p->cs * lnx - p->sn * lny; n[1] = p->sn * lnx + p->cs * lny; n[2] = axis == 2 ? sign : 0.0f;
        return tmin;
    }
    if (p->type == P_CYL) {
        float R = p->b[0], z0 = p->a[2], z1 = p->a[2] + p->b[2], px = o[0] - p->a[0], py = o[1] - p->a[1], best = -1.0f;
        float a = d[0] * d[0] + d[1] * d[1];
        if (a > 0.0f) {
            float b = px * d[0] + py * d[1], c = px * px + py * py - R * R, disc = b * b - a * c;
            if (disc >= 0.0f) {
                float t = (-b - sqrtf(disc)) / a, z = o[2] + t * d[2];
                if (t > 0.0f && z >= z0 && z <= z1) { best = t; n[0] = (px + t * d[0]) / R; n[1] = (py + t * d[1]) / R; n[2] = 0.0f; }
            }
        }
        if (d[2] != 0.0f) {
            float zc = d[2] < 0.0f ? z1 : z0, t = (zc - o[2]) / d[2];
            if (t > 0.0f && (best < 0.0f || t < best)) {
                float hx = px + t * d[0], hy = py + t * d[1];
                if (hx * hx + hy * hy <= R * R) { best = t; n[0] = 0; n[1] = 0; n[2] = d[2] < 0.0f ? 1.0f : -1.0f; }
            }
        }
        return best;
    }
    {   /* capsule */
        float ba[3], oa[3], baba, bard, baoa, rdoa, oaoa, a, b, c, h, t = -1.0f, y = 0.0f, k; int body = 0, j;
        for (j = 0; j < 3; j++) { ba[j] = p->b[j] - p->a[j]; oa[j] = o[j] - p->a[j]; }
        baba = ba[0] * ba[0] + ba[1] * ba[1] + ba[2] * ba[2];
        bard = ba[0] * d[0] + ba[1] * d[1] + ba[2] * d[2];
        baoa = ba[0] * oa[0] + ba[1] * oa[1] + ba[2] * oa[2];
        rdoa = d[0] * oa[0] + d[1] * oa[1] + d[2] * oa[2];
        oaoa = oa[0] * oa[0] + oa[1] * oa[1] + oa[2] * oa[2];
        a = baba - bard * bard; b = baba * rdoa - baoa * bard; c = baba * oaoa - baoa * baoa - p->rad * p->rad * baba;
        h = b * b - a * c;
        if (h < 0.0f || baba == 0.0f) return -1.0f;
        if (a > 0.0f) { t = (-b - sqrtf(h)) / a; y = baoa + t * bard; body = y > 0.0f && y < baba; }
        if (!body) {
            float oc[3];
            for (j = 0; j < 3; j++) oc[j] = y <= 0.0f ? oa[j] : o[j] - p->b[j];
            b = d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2];
            c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - p->rad * p->rad;
            h = b * b - c;
            if (h < 0.0f) return -1.0f;
            t = -b - sqrtf(h);
            y = y <= 0.0f ? 0.0f : baba;
        }
        if (t <= 0.0f) return -1.0f;
        k = y / baba;
        for (j = 0; j < 3; j++) n[j] = (oa[j] + t * d[j] - ba[j] * k) / p->rad;
        return t;
    }
}

static void draw(const prim_t *prims, int np, const cam_t *c, int h, int w, int channels, int choff, uint8_t *img) {
    int row, col, k, j;
    for (row = 0; row < h; row++) for (col = 0; col < w; col++) {
        float sx = (((float)col + 0.5f) / (float)w * 2.0f - 1.0f) * c->tanh, sy = (1.0f - ((float)row + 0.5f) / (float)h * 2.0f) * c->tanh;
        float d[3], inv, best = 3.0e38f, bn[3] = {0, 0, 1}, colr[3] = {0.92f, 0.92f, 0.92f}, shade = 1.0f; int any = 0;
        uint8_t *px = img + ((size_t)row * w + col) * channels + choff;
        for (j = 0; j < 3; j++) d[j] = c->f[j] + sx * c->r[j] + sy * c->u[j];
        inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (j = 0; j < 3; j++) d[j] *= inv;
        for (k = 0; k < np; k++) {
            float n[3], t = hit(&prims[k], c->eye, d, n);
            if (t > 0.0f && t < best) {
                best = t; memcpy(bn, n, sizeof bn); memcpy(colr, prims[k].col, sizeof colr); any = 1;
                if (prims[k].type == P_PLANE) {   /* 1 m checker of plane.urdf, phase/colours read off the reference's gifs */
                    int par = (int)floorf(c->eye[0] + t * d[0]) + (int)floorf(c->eye[1] + t * d[1]);
                    if ((par & 1) == 0) colr[0] = colr[1] = colr[2] = 1.0f;
                }
            }
        }
        if (any) {
            const float lx = -0.40824829f, ly = 0.40824829f, lz = 0.81649658f;
            float ndl = fmaxf(bn[0] * lx + bn[1] * ly + bn[2] * lz, 0.0f);
            shade = 0.6f + 0.4f * ndl;
        }
        for (j = 0; j < 3; j++) px[j] = (uint8_t)(uint32_t)(fminf(colr[j] * shade, 1.0f) * 255.0f + 0.5f);
    }
}

/* kuka state per env: q[7], bq, bx, by (kind 4) + b2q, b2x, b2y (kind 6, Kuka2Button); mobile state per env: x, y, tx, ty,
 * t2x, t2y.  kind: 0..3 mobile family, 4 kuka, 6 kuka with two buttons, 7 kuka with the ten RandButton distractors
 * (state [n][40]: the 10 kuka values + (x, y, present) x 10), 8 the same with the distractors and the ball as free bodies
 * (state [n][106]: those 40 + (x y z vx vy vz) x 11 as srlhip_get_state(KUKA_BODIES) returns them). */
int raster_oracle_render(int kind, int n, int h, int w, int multi_view, const double *state, uint8_t *img) {
    int e, ncam = multi_view ? 2 : 1, channels = 3 * ncam;
    /* kind | 16 (Kuka kinds): every state row carries 5 more doubles at its end — q of the gripper joints 7, 8, 10, 11, 13 — and the
     * gripper is drawn from them with the full model's link frames (needs the oracle in full-model mode); without the flag the
     * gripper is drawn welded to link 7 (the lumped model of rounds 1-2). */
    const int fingers = kind >= 4 && (kind & 16) != 0;
    cam_t cams[2];
    kind &= ~16;
    if (fingers && !kuka_oracle_get_full()) return -1;
    if (kind >= 4) {
        const double t1[3] = {0.316, -0.2, -0.1}, t2[3] = {0.316, 0.316, -0.105};
        cams[0] = camera(t1, 1.1, 145, -36, 0, 60); cams[1] = camera(t2, 1.05, 32, -13, 0, 60);
    } else {
        double t[3] = {2, 2, 0};
        if (kind == 1) t[1] = 0.0;
        const double tf[3] = {-0.25, 0.0, 0.15};   /* fpv camera relative to the robot: mobile_robot_env.py:313-332 */
        cams[0] = camera(t, 4.4, 90, -90, 0, 60); cams[1] = camera(tf, 0.3, 90, -17, 0, 90);
    }
#pragma omp parallel for
    for (e = 0; e < n; e++) {
        prim_t prims[28]; int np = 0, cam;
        if (kind >= 4) {
            const int row = (kind == 6 ? 13 : kind == 7 ? 40 : kind == 8 ? 106 : 10) + (fingers ? 5 : 0);
            const double *s = state + row * (size_t)e; double R[108], p[36], a[3], b[3], q12[12]; float jp[7][3]; int i, k;
            const double locs[5][3] = {{0, 0, 0.10}, {0, 0.030, 0.10}, {0, 0.020, 0.255}, {0, -0.030, 0.10}, {0, -0.020, 0.255}};
            float pts[5][3];
            for (i = 0; i < 12; i++) q12[i] = i < 7 ? s[i] : fingers ? s[row - 5 + (i - 7)] : 0.0;
            kuka_oracle_fk(q12, R, p);
            for (i = 0; i < 7; i++) for (k = 0; k < 3; k++) jp[i][k] = (float)p[3 * i + k];
            for (i = 0; i < 5; i++) {
                for (k = 0; k < 3; k++) { a[k] = p[18 + k] + R[54 + 3 * k] * locs[i][0] + R[54 + 3 * k + 1] * locs[i][1] + R[54 + 3 * k + 2] * locs[i][2]; pts[i][k] = (float)a[k]; }
            }
            (void)b;
            prims[np++] = mk(P_PLANE, 0.68f, 0.78f, 0.94f, 0, 0, -1.0f, 0, 0, 0, 0, 1, 0);
            prims[np++] = mk(P_BOX, 0.85f, 0.75f, 0.62f, 0.5f, 0.0f, -0.22f, 0.75f, 0.5f, 0.025f, 0, 1.0f, 0.0f);
            prims[np++] = mk(P_CYL, 0.0f, 1.0f, 0.0f, (float)s[8], (float)s[9], (float)KM_BUTTON_BASE_Z, 0.10f, 0, 0.03f, 0, 1, 0);
            prims[np++] = mk(P_CYL, 1.0f, 1.0f, 0.0f, (float)s[8], (float)s[9], (float)(KM_BUTTON_BASE_Z + KM_GLIDER_ORIGIN_Z + s[7]), 0.09f, 0, 0.03f, 0, 1, 0);
            if (kind == 6) {   /* urdf/simple_button_2.urdf: cap rgba (0.2, 0.6, 0.38) */
                prims[np++] = mk(P_CYL, 0.0f, 1.0f, 0.0f, (float)s[11], (float)s[12], (float)KM_BUTTON_BASE_Z, 0.10f, 0, 0.03f, 0, 1, 0);
                prims[np++] = mk(P_CYL, 0.2f, 0.6f, 0.38f, (float)s[11], (float)s[12], (float)(KM_BUTTON_BASE_Z + KM_GLIDER_ORIGIN_Z + s[10]), 0.09f, 0, 0.03f, 0, 1, 0);
            }
            prims[np++] = mk(P_CAPSULE, 0.35f, 0.35f, 0.38f, (float)KM_BASE_POS[0], (float)KM_BASE_POS[1], (float)KM_BASE_POS[2], jp[0][0], jp[0][1], jp[0][2], 0.07f, 1, 0);
            for (i = 0; i < 6; i++) prims[np++] = mk(P_CAPSULE, 1.0f, 0.45f, 0.05f, jp[i][0], jp[i][1], jp[i][2], jp[i + 1][0], jp[i + 1][1], jp[i + 1][2], 0.06f, 1, 0);
            if (fingers) {   /* full model: body capsule up to 0.05 above the gripper body's origin, finger link -> tip joint, tip link 0.045 long */
                int side;
                for (k = 0; k < 3; k++) a[k] = p[21 + k] + R[63 + 3 * k + 2] * 0.05;
                prims[np++] = mk(P_CAPSULE, 0.20f, 0.20f, 0.22f, jp[6][0], jp[6][1], jp[6][2], (float)a[0], (float)a[1], (float)a[2], 0.045f, 1, 0);
                for (side = 0; side < 2; side++) {
                    const int f = 8 + 2 * side, t = f + 1;
                    prims[np++] = mk(P_CAPSULE, 0.10f, 0.10f, 0.10f, (float)p[3 * f], (float)p[3 * f + 1], (float)p[3 * f + 2], (float)p[3 * t], (float)p[3 * t + 1], (float)p[3 * t + 2], 0.012f, 1, 0);
                    for (k = 0; k < 3; k++) b[k] = p[3 * t + k] + R[9 * t + 3 * k + 2] * 0.045;
                    prims[np++] = mk(P_CAPSULE, 0.10f, 0.10f, 0.10f, (float)p[3 * t], (float)p[3 * t + 1], (float)p[3 * t + 2], (float)b[0], (float)b[1], (float)b[2], 0.010f, 1, 0);
                }
            } else {
            prims[np++] = mk(P_CAPSULE, 0.20f, 0.20f, 0.22f, jp[6][0], jp[6][1], jp[6][2], pts[0][0], pts[0][1], pts[0][2], 0.045f, 1, 0);
            prims[np++] = mk(P_CAPSULE, 0.10f, 0.10f, 0.10f, pts[1][0], pts[1][1], pts[1][2], pts[2][0], pts[2][1], pts[2][2], 0.015f, 1, 0);
            prims[np++] = mk(P_CAPSULE, 0.10f, 0.10f, 0.10f, pts[3][0], pts[3][1], pts[3][2], pts[4][0], pts[4][1], pts[4][2], 0.015f, 1, 0);
            }
            if (kind == 7 || kind == 8) {   /* KukaRandButton: kept distractors + the ball — scenery at rest (7) or free bodies at their centres (8) */
                const float top = (float)KM_TABLE_TOP_Z;
                const double *rb = kind == 8 ? s + 40 : NULL;
                for (i = 0; i < 10; i++) {
                    double ox = s[10 + 3 * i], oy = s[11 + 3 * i]; uint64_t bx, by; uint32_t type; float x = (float)ox, y = (float)oy, z;
                    if (s[12 + 3 * i] == 0.0) continue;
                    memcpy(&bx, &ox, 8); memcpy(&by, &oy, 8);
                    type = (uint32_t)((bx >> 20) ^ (by >> 20)) % 3u;    /* the reference's type comes from the unseeded global RNG */
                    z = top + (type == 0 ? 0.035f : type == 1 ? 0.012f : 0.025f);
                    if (rb) { x = (float)rb[6 * i]; y = (float)rb[6 * i + 1]; z = (float)rb[6 * i + 2]; }
                    if (type == 0) prims[np++] = mk(P_CAPSULE, 1.0f, 0.85f, 0.1f, x - 0.015f, y, z, x + 0.015f, y, z, 0.035f, 1, 0);
                    else if (type == 1) prims[np++] = mk(P_BOX, 0.8f, 0.1f, 0.1f, x, y, z, 0.016f, 0.032f, 0.012f, 0, 1.0f, 0.0f);
                    else prims[np++] = mk(P_BOX, 0.9f, 0.9f, 0.9f, x, y, z, 0.025f, 0.025f, 0.025f, 0, 1.0f, 0.0f);
                }
                {
                    float x = 0.25f, y = -0.2f, z = top + 0.03f;
                    if (rb) { x = (float)rb[60]; y = (float)rb[61]; z = (float)rb[62]; }
                    prims[np++] = mk(P_CAPSULE, 0.9f, 0.2f, 0.2f, x, y, z, x, y, z + 0.001f, 0.03f, 1, 0);
                }
            }
        } else {
            const double *s = state + 6 * (size_t)e;
            float x = (float)s[0], y = (float)s[1], tx = (float)s[2], ty = (float)s[3], t2x = (float)s[4], t2y = (float)s[5];
            prims[np++] = mk(P_PLANE, 0.68f, 0.78f, 0.94f, 0, 0, 0.0f, 0, 0, 0, 0, 1, 0);
            prims[np++] = mk(P_BOX, 0.66f, 0.0f, 0.0f, 2.0f, 0.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 1.0f, 0.0f);
            if (kind != 1) {
                prims[np++] = mk(P_BOX, 0.0f, 0.0f, 0.0f, 4.0f, 2.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 0.0f, 1.0f);
                prims[np++] = mk(P_BOX, 0.0f, 0.65f, 0.0f, 2.0f, 4.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 1.0f, 0.0f);
                prims[np++] = mk(P_BOX, 0.0f, 0.0f, 0.79f, 0.0f, 2.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 0.0f, 1.0f);
            }
            if (kind == 3) prims[np++] = mk(P_BOX, 1.0f, 1.0f, 0.0f, tx, 2.0f, -0.045f, 2.0f, 0.25f, 0.05f, 0, 0.0f, 1.0f);
            else prims[np++] = mk(P_CYL, 1.0f, 1.0f, 0.0f, tx, ty, 0.0f, 0.18f, 0, 0.03f, 0, 1, 0);
            if (kind == 2) prims[np++] = mk(P_CYL, 0.8f, 0.0f, 0.0f, t2x, t2y, 0.0f, 0.18f, 0, 0.03f, 0, 1, 0);
            prims[np++] = mk(P_BOX, 0.15f, 0.15f, 0.60f, x, y, 0.075f, 0.325f, 0.1f, 0.075f, 0, 1.0f, 0.0f);
        }
        for (cam = 0; cam < ncam; cam++) {
            cam_t c = cams[cam];
            if (kind < 4 && cam == 1) { c.eye[0] += (float)state[6 * (size_t)e]; c.eye[1] += (float)state[6 * (size_t)e + 1]; }
            draw(prims, np, &c, h, w, channels, 3 * cam, img + (size_t)e * h * w * channels);
        }
    }
    return 0;
}

/* runtime model table (oracle/kuka_model.h): this translation unit's copy */
void raster_oracle_set_model(const double *table138) { km_set_model(table138); }
void raster_oracle_get_model(double *table138) { km_get_model(table138); }
