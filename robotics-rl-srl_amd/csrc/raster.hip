// raster.hip — batched tile rasteriser for the raw_pixels observation (gfx950).
//
// Replaces KukaButtonGymEnv.render / MobileRobotGymEnv.render
// (kuka_button_gym_env.py:370-420, mobile_robot_env.py:282-334): pybullet's
// computeViewMatrixFromYawPitchRoll + computeProjectionMatrixFOV(60, 1, 0.1, 100) +
// getCameraImage(ER_TINY_RENDERER) -> RGB u8.  TinyRenderer's mesh visuals, texture
// and shadows are not reproducible without pybullet_data; the contract kept is the
// camera model, object poses and colours, z-buffered, Lambert + ambient shading
// (DESIGN.md §Rasteriser).
//
// Mapping: one 256-lane workgroup per (env, camera).  The env's scene — at most 28
// analytic primitives (plane, z-rotated box, upright cylinder, capsule) — is built
// once per workgroup into LDS from the SoA state (for the Kuka: float64 forward
// kinematics from the cached joint sin/cos).
// The cameras are fixed and the first primitives of a scene (floor plane, table / arena walls)
// never move, so a per-handle setup kernel (raster_bg_k) stores for every pixel its unit ray,
// the depth of the nearest static hit and the shaded static colour.  raster_k then rasterises
// only the MOVING primitives, in object order: each one's conservative screen footprint (for a
// capsule: a slab around its projected axis, row by row) is cut into 64-pixel chunks shared
// by the wavefronts; the exact ray test of that primitive runs for those pixels only and the
// nearest hit is kept by a 64-bit atomic min in an LDS z-buffer that starts from the static
// depth.  The hit pixels are then shaded from a dense list, every other pixel is a copy of the
// cached background, and the band leaves with 12 bytes per lane: the path's HBM traffic is the
// 12 288-byte image per env and nothing else.  Same arithmetic per (primitive, pixel) as a
// full per-pixel z-test, same bytes — culling decides only which tests are skipped.
// (The MobileRobot fpv camera rides on the robot: raster_tiles_k walks 8x8 tiles with the full test.)
// float32 throughout, -ffp-contract=off so that the C oracle (oracle/raster_oracle.c)
// reproduces the bytes.
#include "internal.hpp"
#include "kuka_core.hpp"
#include "kuka_tree_model.hpp"

namespace srl {

// KukaState / planes are private to kuka.hip; the rasteriser gets raw plane pointers (RasterKukaView: internal.hpp).
struct RasterMobileView { const double *x, *y, *tx, *ty, *t2x, *t2y; const int32_t *cur; };

namespace {

constexpr int kRasterBlock = 256;
constexpr int kMaxPrims = 28;   // Kuka scene 16 (full model: body + two finger + two tip capsules; lumped 14) + second button 2 or ten distractors + ball 11
constexpr int kTilePixels = 8192;       // LDS band buffer (24 KiB): whole 8-row tile strips, image width <= 1024

enum { PRIM_PLANE = 0, PRIM_BOX = 1, PRIM_CYL = 2, PRIM_CAPSULE = 3 };

struct Prim {
    int type;
    float r, g, b;
    float ax, ay, az;      // plane: (., ., z0)  box: centre  cylinder: base centre  capsule: end a
    float bx, by, bz;      // box: half extents             cylinder: (radius, ., height)  capsule: end b
    float rad;             // capsule radius
    float cs, sn;          // box: cos/sin of the yaw about z
};

struct Camera {
    float ex, ey, ez;      // eye
    float fx, fy, fz;      // forward (unit)
    float rx, ry, rz;      // right   (unit)
    float ux, uy, uz;      // up      (unit)
    float tan_half_fov;
};

struct RasterParams {
    int32_t kind, n, h, w, channels, ncam;
    int32_t fpv;           // mobile family, second camera: rides on the robot (cam[1] is stored relative to the robot position)
    int32_t cam0;          // first camera this launch renders (blockIdx.y counts from it)
    int32_t nstatic;       // the first nstatic primitives of every env's scene are identical and never move
    const float4 *rays[2]; // per fixed camera: (unit ray, depth of the nearest static hit or 3e38) per pixel; null = full path
    const uint32_t *bg[2]; //                   shaded static colour per pixel (r | g << 8 | b << 16)
    Camera cam[2];
};

__device__ __forceinline__ void set_prim(Prim &p, int type, float r, float g, float b, float ax, float ay, float az,
                                         float bx, float by, float bz, float rad, float cs, float sn) {
    p.type = type; p.r = r; p.g = g; p.b = b; p.ax = ax; p.ay = ay; p.az = az; p.bx = bx; p.by = by; p.bz = bz;
    p.rad = rad; p.cs = cs; p.sn = sn;
}

// ---- ray / primitive intersection: returns t (> 0) or -1, and the surface normal ----------------------
// The hit functions return t only, plus a 3-bit `code` that says which face / part was hit: normals are evaluated once, for
// the nearest hit (prim_normal), not for every candidate — their divisions were a third of a capsule test.
//   box: axis | 4 when the face normal points along +axis;  cylinder: 0 side, 1 cap;  capsule: 0 body, 1 sphere at a, 2 sphere at b
__device__ __forceinline__ float hit_plane(const Prim &p, float oz, float dz) {
    if (dz == 0.0f) return -1.0f;
    const float t = (p.az - oz) / dz;
    return t > 0.0f ? t : -1.0f;
}

__device__ __forceinline__ float hit_box(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz, int &code) {
    // into the box frame (rotation about z by -yaw)
    const float px = ox - p.ax, py = oy - p.ay, pz = oz - p.az;
    const float lox = p.cs * px + p.sn * py, loy = p.cs * py - p.sn * px;
    const float ldx = p.cs * dx + p.sn * dy, ldy = p.cs * dy - p.sn * dx;
    float tmin = -3.0e38f, tmax = 3.0e38f;
    int axis = 0; float sign = 0.0f;
    const float o[3] = {lox, loy, pz}, d[3] = {ldx, ldy, dz}, h[3] = {p.bx, p.by, p.bz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (d[k] == 0.0f) {
            if (o[k] < -h[k] || o[k] > h[k]) return -1.0f;
        } else {
            const float inv = 1.0f / d[k];
            float t0 = (-h[k] - o[k]) * inv, t1 = (h[k] - o[k]) * inv;
            float s = -1.0f;
            if (t0 > t1) { const float tt = t0; t0 = t1; t1 = tt; s = 1.0f; }
            if (t0 > tmin) { tmin = t0; axis = k; sign = s; }
            if (t1 < tmax) tmax = t1;
        }
    }
    if (tmin > tmax || tmin <= 0.0f) return -1.0f;
    code = axis | (sign > 0.0f ? 4 : 0);
    return tmin;
}

__device__ __forceinline__ float hit_cylinder(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz, int &code) {
    const float R = p.bx, z0 = p.az, z1 = p.az + p.bz;
    const float px = ox - p.ax, py = oy - p.ay;
    float best = -1.0f;
    const float a = dx * dx + dy * dy;
    if (a > 0.0f) {
        const float b = px * dx + py * dy, c = px * px + py * py - R * R;
        const float disc = b * b - a * c;
        if (disc >= 0.0f) {
            const float t = (-b - sqrtf(disc)) / a;
            const float z = oz + t * dz;
            if (t > 0.0f && z >= z0 && z <= z1) { best = t; code = 0; }
        }
    }
    if (dz != 0.0f) {       // caps (the top one is what a camera above ever sees)
        const float zc = dz < 0.0f ? z1 : z0;
        const float t = (zc - oz) / dz;
        if (t > 0.0f && (best < 0.0f || t < best)) {
            const float hx = px + t * dx, hy = py + t * dy;
            if (hx * hx + hy * hy <= R * R) { best = t; code = 1; }
        }
    }
    return best;
}

__device__ __forceinline__ float hit_capsule(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz, int &code) {
    const float bax = p.bx - p.ax, bay = p.by - p.ay, baz = p.bz - p.az;
    const float oax = ox - p.ax, oay = oy - p.ay, oaz = oz - p.az;
    const float baba = bax * bax + bay * bay + baz * baz;
    const float bard = bax * dx + bay * dy + baz * dz;
    const float baoa = bax * oax + bay * oay + baz * oaz;
    const float rdoa = dx * oax + dy * oay + dz * oaz;
    const float oaoa = oax * oax + oay * oay + oaz * oaz;
    const float a = baba - bard * bard;
    float b = baba * rdoa - baoa * bard;
    float c = baba * oaoa - baoa * baoa - p.rad * p.rad * baba;
    float h = b * b - a * c;
    if (h < 0.0f || baba == 0.0f) return -1.0f;
    float t = -1.0f, y = 0.0f;
    bool body = false;
    code = 0;
    if (a > 0.0f) {
        t = (-b - sqrtf(h)) / a;
        y = baoa + t * bard;
        body = y > 0.0f && y < baba;
    }
    if (!body) {            // one of the two end spheres
        const float ocx = y <= 0.0f ? oax : ox - p.bx, ocy = y <= 0.0f ? oay : oy - p.by, ocz = y <= 0.0f ? oaz : oz - p.bz;
        b = dx * ocx + dy * ocy + dz * ocz;
        c = ocx * ocx + ocy * ocy + ocz * ocz - p.rad * p.rad;
        h = b * b - c;
        if (h < 0.0f) return -1.0f;
        t = -b - sqrtf(h);
        code = y <= 0.0f ? 1 : 2;
    }
    if (t <= 0.0f) return -1.0f;
    return t;
}

// normal of primitive p where the ray (eye o, unit direction d) hits it at parameter t; code as left by its hit function
__device__ __forceinline__ void prim_normal(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz, float t, int code,
                                            float &nx, float &ny, float &nz) {
    if (p.type == PRIM_PLANE) { nx = 0.0f; ny = 0.0f; nz = 1.0f; }
    else if (p.type == PRIM_BOX) {
        const int axis = code & 3;
        const float sign = (code & 4) ? 1.0f : -1.0f;
        const float lnx = axis == 0 ? sign : 0.0f, lny = axis == 1 ? sign : 0.0f;
        nx = p.cs * lnx - p.sn * lny; ny = p.sn * lnx + p.cs * lny; nz = axis == 2 ? sign : 0.0f;
    } else if (p.type == PRIM_CYL) {
        if (code == 0) { nx = (ox - p.ax + t * dx) / p.bx; ny = (oy - p.ay + t * dy) / p.bx; nz = 0.0f; }
        else { nx = 0.0f; ny = 0.0f; nz = dz < 0.0f ? 1.0f : -1.0f; }
    } else {
        const float bax = p.bx - p.ax, bay = p.by - p.ay, baz = p.bz - p.az;
        const float oax = ox - p.ax, oay = oy - p.ay, oaz = oz - p.az;
        const float baba = bax * bax + bay * bay + baz * baz;
        const float bard = bax * dx + bay * dy + baz * dz;
        const float baoa = bax * oax + bay * oay + baz * oaz;
        const float y = code == 0 ? baoa + t * bard : (code == 1 ? 0.0f : baba);      // where along the axis, as hit_capsule had it
        const float k = y / baba;
        nx = (oax + t * dx - bax * k) / p.rad; ny = (oay + t * dy - bay * k) / p.rad; nz = (oaz + t * dz - baz * k) / p.rad;
    }
}

// colour of the hit (primitive p, parameter t, code) seen along the unit ray d from the eye: what the z-test winner becomes
__device__ __forceinline__ uint32_t shade_hit(const Prim &p, const Camera &c, float dx, float dy, float dz, float t, int code);

// ---- scenes ----------------------------------------------------------------------------------------------
__device__ int build_mobile_scene(const RasterParams &rp, const RasterMobileView &v, int e, Prim *prims) {
    const float x = (float)v.x[e], y = (float)v.y[e];
    const float tx = (float)v.tx[e], ty = (float)v.ty[e], t2x = (float)v.t2x[e], t2y = (float)v.t2y[e];
    int n = 0;
    set_prim(prims[n++], PRIM_PLANE, 0.68f, 0.78f, 0.94f, 0, 0, 0.0f, 0, 0, 0, 0, 1, 0);
    set_prim(prims[n++], PRIM_BOX, 0.66f, 0.0f, 0.0f, 2.0f, 0.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 1.0f, 0.0f);           // wall_left
    if (rp.kind != SRLHIP_ENV_MOBILE_1D) {
        set_prim(prims[n++], PRIM_BOX, 0.0f, 0.0f, 0.0f, 4.0f, 2.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 0.0f, 1.0f);       // wall_bottom
        set_prim(prims[n++], PRIM_BOX, 0.0f, 0.65f, 0.0f, 2.0f, 4.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 1.0f, 0.0f);       // wall_right
        set_prim(prims[n++], PRIM_BOX, 0.0f, 0.0f, 0.79f, 0.0f, 2.0f, 0.0f, 2.0f, 0.05f, 0.05f, 0, 0.0f, 1.0f);       // wall_top
    }
    if (rp.kind == SRLHIP_ENV_MOBILE_LINE)
        set_prim(prims[n++], PRIM_BOX, 1.0f, 1.0f, 0.0f, tx, 2.0f, -0.045f, 2.0f, 0.25f, 0.05f, 0, 0.0f, 1.0f);      // line target
    else
        set_prim(prims[n++], PRIM_CYL, 1.0f, 1.0f, 0.0f, tx, ty, 0.0f, 0.18f, 0, 0.03f, 0, 1, 0);
    if (rp.kind == SRLHIP_ENV_MOBILE_2TARGET)
        set_prim(prims[n++], PRIM_CYL, 0.8f, 0.0f, 0.0f, t2x, t2y, 0.0f, 0.18f, 0, 0.03f, 0, 1, 0);
    set_prim(prims[n++], PRIM_BOX, 0.15f, 0.15f, 0.60f, x, y, 0.075f, 0.325f, 0.1f, 0.075f, 0, 1.0f, 0.0f);          // robot
    return n;
}

// frame of a tree link from its parent's: child = parent * (xyz, Rj * Rot(axis, q)), R as columns x y z.  The same operations in the
// same order as oracle/kuka_oracle.c (axis_rotation / rpy_to_mat for z axes, mat3_mul, mat3_vec): identical float64 results.
__device__ void tree_child_frame(const double Rp[9], const double pp[3], const RasterGripJoint &J, double s, double c, double Rc[9], double pc[3]) {
    double Rq[3][3], L[3][3];                                  // row-major like the oracle's mat3
    const double a0 = J.axis[0], a1 = J.axis[1], a2 = J.axis[2], vv = 1.0 - c;
    if (a2 == 1.0) {                                           // rpy_to_mat(0, 0, q)
        Rq[0][0] = c; Rq[0][1] = -s; Rq[0][2] = 0.0; Rq[1][0] = s; Rq[1][1] = c; Rq[1][2] = 0.0; Rq[2][0] = 0.0; Rq[2][1] = 0.0; Rq[2][2] = 1.0;
    } else {
        Rq[0][0] = c + a0 * a0 * vv;      Rq[0][1] = a0 * a1 * vv - a2 * s; Rq[0][2] = a0 * a2 * vv + a1 * s;
        Rq[1][0] = a1 * a0 * vv + a2 * s; Rq[1][1] = c + a1 * a1 * vv;      Rq[1][2] = a1 * a2 * vv - a0 * s;
        Rq[2][0] = a2 * a0 * vv - a1 * s; Rq[2][1] = a2 * a1 * vv + a0 * s; Rq[2][2] = c + a2 * a2 * vv;
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) L[i][j] = J.Rj[3 * i] * Rq[0][j] + J.Rj[3 * i + 1] * Rq[1][j] + J.Rj[3 * i + 2] * Rq[2][j];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        pc[i] = pp[i] + (Rp[i] * J.xyz[0] + Rp[3 + i] * J.xyz[1] + Rp[6 + i] * J.xyz[2]);
#pragma unroll
        for (int j = 0; j < 3; j++) Rc[3 * j + i] = Rp[i] * L[0][j] + Rp[3 + i] * L[1][j] + Rp[6 + i] * L[2][j];
    }
}

// Full model, pre-pass of every render: one lane per env composes the frames of the tree's links 7..11 — gripper_to_arm, left finger -> left
// tip, right finger -> right tip — from the installed table exactly as oracle/kuka_oracle.c::forward_kinematics does, and stores the
// seven points the gripper's capsules span as float32 [21][n]: body end (0.05 above the gripper body's origin), per side finger joint
// origin, tip joint origin, tip end (0.045 along the tip link).
__global__ void __launch_bounds__(64) raster_grip_k(RasterKukaView v, float *__restrict__ out) {
    using namespace kuka;
    const int e = blockIdx.x * 64 + threadIdx.x;
    const int64_t n = v.n;
    if (e >= n) return;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {kBasePos[0], kBasePos[1], kBasePos[2]};
    // planes 21..41: the arm's joint origins jp[0..6] — the float64 forward kinematics build_kuka_scene used to run on ONE lane of every
    // 256-lane rasteriser workgroup (with the other 255 waiting at the barrier) is done here once per env, one env per lane
#define SRL_FK(I) { fk_forward<I>(R, p, v.sq[(I) * n + e], v.cq[(I) * n + e]); \
                    out[(int64_t)(21 + 3 * (I)) * n + e] = (float)p[0]; out[(int64_t)(22 + 3 * (I)) * n + e] = (float)p[1]; out[(int64_t)(23 + 3 * (I)) * n + e] = (float)p[2]; }
    SRL_FK(0) SRL_FK(1) SRL_FK(2) SRL_FK(3) SRL_FK(4) SRL_FK(5) SRL_FK(6)
#undef SRL_FK
    double Rg[5][9], pg[5][3];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const RasterGripJoint &J = v.gj[i];
        const int par = (int)J.parent;                        // 6 (link_7) or an earlier gripper link
        double Rp[9], pp[3];                                  // (selects over the unrolled earlier links: no dynamically indexed local array)
#pragma unroll
        for (int k = 0; k < 9; k++) Rp[k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) pp[k] = p[k];
#pragma unroll
        for (int a_ = 0; a_ < i; a_++) {
            const bool take = par == 7 + a_;
#pragma unroll
            for (int k = 0; k < 9; k++) Rp[k] = take ? Rg[a_][k] : Rp[k];
#pragma unroll
            for (int k = 0; k < 3; k++) pp[k] = take ? pg[a_][k] : pp[k];
        }
        tree_child_frame(Rp, pp, J, v.gsq[i * n + e], v.gcq[i * n + e], Rg[i], pg[i]);
    }
    const double up5[3] = {0, 0, 0.05}, up45[3] = {0, 0, 0.045};
    double a[3];
    tip_point(Rg[0], pg[0], up5, a);
#pragma unroll
    for (int k = 0; k < 3; k++) out[(int64_t)k * n + e] = (float)a[k];
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int f = 1 + 2 * side, t = f + 1;
        tip_point(Rg[t], pg[t], up45, a);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            out[(int64_t)(3 + 9 * side + k) * n + e] = (float)pg[f][k];
            out[(int64_t)(6 + 9 * side + k) * n + e] = (float)pg[t][k];
            out[(int64_t)(9 + 9 * side + k) * n + e] = (float)a[k];
        }
    }
}

__device__ int build_kuka_scene(const RasterKukaView &v, int e, Prim *prims) {
    using namespace kuka;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {kBasePos[0], kBasePos[1], kBasePos[2]};
    float jp[ND][3];
    const int64_t n = v.n;
    if (v.has_tm) {                // full model: the pre-pass (raster_grip_k) has run the arm's forward kinematics for this env
#pragma unroll
        for (int i = 0; i < ND; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) jp[i][k] = v.grip[(int64_t)(21 + 3 * i + k) * n + e];
    } else {
#define SRL_FK(I) { fk_forward<I>(R, p, v.sq[(I) * n + e], v.cq[(I) * n + e]); jp[I][0] = (float)p[0]; jp[I][1] = (float)p[1]; jp[I][2] = (float)p[2]; }
        SRL_FK(0) SRL_FK(1) SRL_FK(2) SRL_FK(3) SRL_FK(4) SRL_FK(5) SRL_FK(6)
#undef SRL_FK
    }
    const float bx = (float)v.bx[e], by = (float)v.by[e], cap_z = (float)(v.bz[e] + kGliderOriginZ + v.bq[e]);
    int k = 0;
    set_prim(prims[k++], PRIM_PLANE, 0.68f, 0.78f, 0.94f, 0, 0, -1.0f, 0, 0, 0, 0, 1, 0);
    set_prim(prims[k++], PRIM_BOX, 0.85f, 0.75f, 0.62f, 0.5f, 0.0f, -0.22f, 0.75f, 0.5f, 0.025f, 0, 1.0f, 0.0f);     // table top
    set_prim(prims[k++], PRIM_CYL, 0.0f, 1.0f, 0.0f, bx, by, (float)v.bz[e], 0.10f, 0, 0.03f, 0, 1, 0);              // button base
    set_prim(prims[k++], PRIM_CYL, 1.0f, 1.0f, 0.0f, bx, by, cap_z, 0.09f, 0, 0.03f, 0, 1, 0);                       // button cap
    if (v.two) {                                                      // urdf/simple_button_2.urdf: cap rgba (0.2, 0.6, 0.38)
        const float b2x = (float)v.b2x[e], b2y = (float)v.b2y[e], cap2_z = (float)(v.bz[e] + kGliderOriginZ + v.b2q[e]);
        set_prim(prims[k++], PRIM_CYL, 0.0f, 1.0f, 0.0f, b2x, b2y, (float)v.bz[e], 0.10f, 0, 0.03f, 0, 1, 0);
        set_prim(prims[k++], PRIM_CYL, 0.2f, 0.6f, 0.38f, b2x, b2y, cap2_z, 0.09f, 0, 0.03f, 0, 1, 0);
    }
    set_prim(prims[k++], PRIM_CAPSULE, 0.35f, 0.35f, 0.38f, (float)kBasePos[0], (float)kBasePos[1], (float)kBasePos[2],
             jp[0][0], jp[0][1], jp[0][2], 0.07f, 1, 0);
    for (int i = 0; i < ND - 1; i++)
        set_prim(prims[k++], PRIM_CAPSULE, 1.0f, 0.45f, 0.05f, jp[i][0], jp[i][1], jp[i][2], jp[i + 1][0], jp[i + 1][1],
                 jp[i + 1][2], 0.06f, 1, 0);
    double a[3], b[3];
    if (v.has_tm) {
        // Full model: the gripper from its own joint state (kuka_button_gym_env.py:370-420 renders the bodies where they are): body, two
        // finger and two tip capsules between the seven points raster_grip_k has computed for this env (a pre-pass: the float64 frame
        // compositions inlined here cost raster_k its 72-register footprint — 256 VGPRs, occupancy 6 -> 1, 0.10 -> 0.22 ms measured)
        float g[21];
#pragma unroll
        for (int k = 0; k < 21; k++) g[k] = v.grip[(int64_t)k * n + e];
        set_prim(prims[k++], PRIM_CAPSULE, 0.20f, 0.20f, 0.22f, jp[6][0], jp[6][1], jp[6][2], g[0], g[1], g[2], 0.045f, 1, 0);
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const float *q = g + 3 + 9 * side;                  // finger joint origin, tip joint origin, tip end
            set_prim(prims[k++], PRIM_CAPSULE, 0.10f, 0.10f, 0.10f, q[0], q[1], q[2], q[3], q[4], q[5], 0.012f, 1, 0);
            set_prim(prims[k++], PRIM_CAPSULE, 0.10f, 0.10f, 0.10f, q[3], q[4], q[5], q[6], q[7], q[8], 0.010f, 1, 0);
        }
    } else {
        // lumped model (rounds 1-2): the gripper welded to link 7
        const double body[3] = {0, 0, 0.10}, fa0[3] = {0, 0.030, 0.10}, fa1[3] = {0, 0.020, 0.255}, fb0[3] = {0, -0.030, 0.10},
                     fb1[3] = {0, -0.020, 0.255};
        tip_point(R, p, body, a);
        set_prim(prims[k++], PRIM_CAPSULE, 0.20f, 0.20f, 0.22f, jp[6][0], jp[6][1], jp[6][2], (float)a[0], (float)a[1], (float)a[2], 0.045f, 1, 0);
        tip_point(R, p, fa0, a); tip_point(R, p, fa1, b);
        set_prim(prims[k++], PRIM_CAPSULE, 0.10f, 0.10f, 0.10f, (float)a[0], (float)a[1], (float)a[2], (float)b[0], (float)b[1], (float)b[2], 0.015f, 1, 0);
        tip_point(R, p, fb0, a); tip_point(R, p, fb1, b);
        set_prim(prims[k++], PRIM_CAPSULE, 0.10f, 0.10f, 0.10f, (float)a[0], (float)a[1], (float)a[2], (float)b[0], (float)b[1], (float)b[2], 0.015f, 1, 0);
    }
    if (v.rand_objects) {
        // KukaRandButtonGymEnv scenery (kuka_rand_button_gym_env.py:60-71): kept distractors resting on the table, and the
        // ball at its drop position.  The reference draws the object TYPE from the global unseeded np.random; here it is a
        // hash of the position: 0 duck (yellow blob), 1 lego (small red brick), 2 cube_small (5 cm cube).
        // Full model (v.rb): they are free bodies (kuka_tree.hpp, free-body section) and are drawn at their current centres.
        const float top = (float)kTableTopZ;
        for (int i = 0; i < 10; i++) {
            const double ox = v.objs[(3 * i) * n + e], oy = v.objs[(3 * i + 1) * n + e];
            if (v.objs[(3 * i + 2) * n + e] == 0.0) continue;
            const uint32_t type = (uint32_t)(((uint64_t)__double_as_longlong(ox) >> 20) ^ ((uint64_t)__double_as_longlong(oy) >> 20)) % 3u;
            float x = (float)ox, y = (float)oy, z = top + (type == 0 ? 0.035f : type == 1 ? 0.012f : 0.025f);
            if (v.rb) { x = (float)v.rb[(6 * i) * n + e]; y = (float)v.rb[(6 * i + 1) * n + e]; z = (float)v.rb[(6 * i + 2) * n + e]; }
            if (type == 0) set_prim(prims[k++], PRIM_CAPSULE, 1.0f, 0.85f, 0.1f, x - 0.015f, y, z, x + 0.015f, y, z, 0.035f, 1, 0);
            else if (type == 1) set_prim(prims[k++], PRIM_BOX, 0.8f, 0.1f, 0.1f, x, y, z, 0.016f, 0.032f, 0.012f, 0, 1.0f, 0.0f);
            else set_prim(prims[k++], PRIM_BOX, 0.9f, 0.9f, 0.9f, x, y, z, 0.025f, 0.025f, 0.025f, 0, 1.0f, 0.0f);
        }
        {
            float x = 0.25f, y = -0.2f, z = top + 0.03f;
            if (v.rb) { x = (float)v.rb[60 * n + e]; y = (float)v.rb[61 * n + e]; z = (float)v.rb[62 * n + e]; }
            set_prim(prims[k++], PRIM_CAPSULE, 0.9f, 0.2f, 0.2f, x, y, z, x, y, z + 0.001f, 0.03f, 1, 0);   // sphere_small
        }
    }
    return k;
}

__device__ __forceinline__ uint32_t finish_pixel(bool hit, float bnx, float bny, float bnz, float cr, float cg, float cb) {
    float shade = 1.0f;
    if (hit) {
        // one directional light, ambient 0.6 + diffuse 0.4 (TinyRenderer-like proportions), no shadows
        const float lx = -0.40824829f, ly = 0.40824829f, lz = 0.81649658f;
        const float ndl = fmaxf(bnx * lx + bny * ly + bnz * lz, 0.0f);
        shade = 0.6f + 0.4f * ndl;
    }
    const uint32_t r8 = (uint32_t)(fminf(cr * shade, 1.0f) * 255.0f + 0.5f);
    const uint32_t g8 = (uint32_t)(fminf(cg * shade, 1.0f) * 255.0f + 0.5f);
    const uint32_t b8 = (uint32_t)(fminf(cb * shade, 1.0f) * 255.0f + 0.5f);
    return r8 | (g8 << 8) | (b8 << 16);
}

__device__ __forceinline__ uint32_t shade_hit(const Prim &p, const Camera &c, float dx, float dy, float dz, float t, int code) {
    float nx, ny, nz, cr = p.r, cg = p.g, cb = p.b;
    prim_normal(p, c.ex, c.ey, c.ez, dx, dy, dz, t, code, nx, ny, nz);
    if (p.type == PRIM_PLANE) {
        // plane.urdf's texture: 1 m blue/white checker aligned with the world axes (period, phase and the two
        // colours measured on the reference's imgs/mobile_robot.gif: x in [0,1) x y in [0,1) is white)
        const int par = (int)floorf(c.ex + t * dx) + (int)floorf(c.ey + t * dy);
        if ((par & 1) == 0) { cr = 1.0f; cg = 1.0f; cb = 1.0f; }
    }
    return finish_pixel(true, nx, ny, nz, cr, cg, cb);
}

__device__ __forceinline__ float hit_prim(const Prim &p, const Camera &c, float dx, float dy, float dz, int &code) {
    code = 0;
    if (p.type == PRIM_PLANE) return hit_plane(p, c.ez, dz);
    if (p.type == PRIM_BOX) return hit_box(p, c.ex, c.ey, c.ez, dx, dy, dz, code);
    if (p.type == PRIM_CYL) return hit_cylinder(p, c.ex, c.ey, c.ez, dx, dy, dz, code);
    return hit_capsule(p, c.ex, c.ey, c.ez, dx, dy, dz, code);
}

// z-test of the unit ray (dx, dy, dz) from the eye against the primitives in `mask` (wave-uniform), starting from `best`;
// false = nothing nearer than `best`, else rgb is the shaded winner
__device__ __forceinline__ bool trace(const Prim *prims, uint64_t mask, const Camera &c, float dx, float dy, float dz, float best,
                                      uint32_t &rgb) {
    int win = -1, wcode = 0;
    while (mask) {                                   // wave-uniform list of the primitives that can touch this tile
        const int k = __builtin_ctzll(mask);
        mask &= mask - 1;
        int code;
        const float t = hit_prim(prims[k], c, dx, dy, dz, code);
        if (t > 0.0f && t < best) { best = t; win = k; wcode = code; }
    }
    if (win < 0) return false;
    rgb = shade_hit(prims[win], c, dx, dy, dz, best, wcode);
    return true;
}

__device__ __forceinline__ void pixel_ray(const Camera &c, float sx, float sy, float &dx, float &dy, float &dz) {
    dx = c.fx + sx * c.rx + sy * c.ux; dy = c.fy + sx * c.ry + sy * c.uy; dz = c.fz + sx * c.rz + sy * c.uz;
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= inv; dy *= inv; dz *= inv;
}

__device__ __forceinline__ uint32_t shade_pixel(const Prim *prims, uint64_t mask, const Camera &c, float sx, float sy) {
    float dx, dy, dz;
    pixel_ray(c, sx, sy, dx, dy, dz);
    uint32_t rgb;
    if (trace(prims, mask, c, dx, dy, dz, 3.0e38f, rgb)) return rgb;
    return finish_pixel(false, 0.0f, 0.0f, 1.0f, 0.92f, 0.92f, 0.92f);                                   // background
}

// Conservative screen rectangle (tangent-space x = X/Z, y = Y/Z in the camera frame) of a primitive: the eight
// corners of its world-space bounding box are projected; anything reaching behind the near plane covers the screen.
__device__ void prim_screen_rect(const Prim &p, const Camera &c, float rect[4]) {
    float lo[3], hi[3];
    if (p.type == PRIM_PLANE) { rect[0] = -3.0e38f; rect[1] = 3.0e38f; rect[2] = -3.0e38f; rect[3] = 3.0e38f; return; }
    if (p.type == PRIM_BOX) {
        const float hx = fabsf(p.cs) * p.bx + fabsf(p.sn) * p.by, hy = fabsf(p.sn) * p.bx + fabsf(p.cs) * p.by;
        lo[0] = p.ax - hx; hi[0] = p.ax + hx; lo[1] = p.ay - hy; hi[1] = p.ay + hy; lo[2] = p.az - p.bz; hi[2] = p.az + p.bz;
    } else if (p.type == PRIM_CYL) {
        lo[0] = p.ax - p.bx; hi[0] = p.ax + p.bx; lo[1] = p.ay - p.bx; hi[1] = p.ay + p.bx; lo[2] = p.az; hi[2] = p.az + p.bz;
    } else {
        lo[0] = fminf(p.ax, p.bx) - p.rad; hi[0] = fmaxf(p.ax, p.bx) + p.rad;
        lo[1] = fminf(p.ay, p.by) - p.rad; hi[1] = fmaxf(p.ay, p.by) + p.rad;
        lo[2] = fminf(p.az, p.bz) - p.rad; hi[2] = fmaxf(p.az, p.bz) + p.rad;
    }
    const float eps = 1.0e-3f;                      // slack for float rounding of the projection
    float x0 = 3.0e38f, x1 = -3.0e38f, y0 = 3.0e38f, y1 = -3.0e38f;
    bool behind = false;
    for (int k = 0; k < 8; k++) {
        const float wx = ((k & 1) ? hi[0] : lo[0]) - c.ex, wy = ((k & 2) ? hi[1] : lo[1]) - c.ey, wz = ((k & 4) ? hi[2] : lo[2]) - c.ez;
        const float z = wx * c.fx + wy * c.fy + wz * c.fz;
        if (z <= 0.05f) { behind = true; break; }
        const float x = (wx * c.rx + wy * c.ry + wz * c.rz) / z, y = (wx * c.ux + wy * c.uy + wz * c.uz) / z;
        x0 = fminf(x0, x); x1 = fmaxf(x1, x); y0 = fminf(y0, y); y1 = fmaxf(y1, y);
    }
    if (behind) { rect[0] = -3.0e38f; rect[1] = 3.0e38f; rect[2] = -3.0e38f; rect[3] = 3.0e38f; return; }
    rect[0] = x0 - eps; rect[1] = x1 + eps; rect[2] = y0 - eps; rect[3] = y1 + eps;
}

// Tile-order path: every wavefront walks 8x8 pixel tiles and z-tests, in registers, the primitives whose rectangle overlaps
// the tile.  It serves the cameras without a cached background (the MobileRobot fpv camera, cam0 = 1), where every pixel
// needs its own ray and the floor plane covers the screen anyway.
__global__ void __launch_bounds__(kRasterBlock)
raster_tiles_k(RasterParams rp, RasterKukaView kv, RasterMobileView mv, uint8_t *img) {
    __shared__ Prim prims[kMaxPrims];
    __shared__ float rects[kMaxPrims][4];
    __shared__ int nprims;
    __shared__ uint32_t tile_masks[kTilePixels / 64];                     // per 8x8 tile of the current band: primitives whose rectangle overlaps it
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTilePixels * 3];
    const int e = blockIdx.x, cam = rp.cam0 + blockIdx.y;
    Camera c = rp.cam[cam];
    if (rp.fpv && cam == 1) { c.ex += (float)mv.x[e]; c.ey += (float)mv.y[e]; }      // mobile_robot_env.py:315-323
    if (threadIdx.x == 0)
        nprims = rp.kind >= SRLHIP_ENV_KUKA_BUTTON ? build_kuka_scene(kv, e, prims) : build_mobile_scene(rp, mv, e, prims);
    __syncthreads();
    if ((int)threadIdx.x < nprims) prim_screen_rect(prims[threadIdx.x], c, rects[threadIdx.x]);
    __syncthreads();
    const int npix = rp.h * rp.w, np = nprims;
    const float4 *__restrict__ rays = rp.rays[cam];
    const uint32_t *__restrict__ bg = rp.bg[cam];
    const bool use_bg = rays != nullptr && !(rp.fpv && cam == 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lx = lane & 7, ly = lane >> 3;
    uint8_t *out = img + (int64_t)e * npix * rp.channels;
    // bands of whole 8-row tile strips that fit the LDS tile buffer; inside a band every wavefront walks 8x8 tiles
    const int band_rows = max(8, (kTilePixels / rp.w) & ~7);
    const int tiles_x = (rp.w + 7) >> 3;
    const float inv_w2 = 2.0f / (float)rp.w, inv_h2 = 2.0f / (float)rp.h;
    for (int row0 = 0; row0 < rp.h; row0 += band_rows) {
        const int rows = min(band_rows, rp.h - row0), base = row0 * rp.w, count = rows * rp.w;
        const int ntiles = ((rows + 7) >> 3) * tiles_x;
        // tile -> primitive masks, one tile per lane (instead of one ballot per tile per wavefront): tile rectangle in
        // tangent space (pixel edges; y grows upwards while rows grow downwards), culling only — the primitives' rectangles
        // carry 1e-3 of slack — so reciprocals instead of divisions
        for (int t = threadIdx.x; t < ntiles; t += kRasterBlock) {
            const int tyy = t / tiles_x, txx = t - tyy * tiles_x, col0 = txx * 8, r0 = row0 + tyy * 8;
            const float tx0 = ((float)col0 * inv_w2 - 1.0f) * c.tan_half_fov;
            const float tx1 = ((float)(col0 + 8) * inv_w2 - 1.0f) * c.tan_half_fov;
            const float ty1 = (1.0f - (float)r0 * inv_h2) * c.tan_half_fov;
            const float ty0 = (1.0f - (float)(r0 + 8) * inv_h2) * c.tan_half_fov;
            uint32_t m = 0;
            for (int k = 0; k < np; k++)
                if (rects[k][0] <= tx1 && rects[k][1] >= tx0 && rects[k][2] <= ty1 && rects[k][3] >= ty0) m |= 1u << k;
            tile_masks[t] = m;
        }
        __syncthreads();
        int ty = wave / tiles_x, tx = wave - ty * tiles_x;                 // one division per band, not per tile
        for (int t = wave; t < ntiles; t += kRasterBlock / 64, tx += kRasterBlock / 64) {
            while (tx >= tiles_x) { tx -= tiles_x; ty++; }
            const int col0 = tx * 8, r0 = row0 + ty * 8;
            uint64_t mask = (uint64_t)__builtin_amdgcn_readfirstlane((int)tile_masks[t]) & 0xffffffffull;
            const int row = r0 + ly, col = col0 + lx;
            if (row < rp.h && col < rp.w) {
                uint32_t rgb;
                if (use_bg) {
                    // static scenery comes from the per-handle background; only moving primitives are traced
                    mask &= ~((1ull << rp.nstatic) - 1ull);
                    rgb = bg[row * rp.w + col];
                    if (mask) {
                        const float4 ray = rays[row * rp.w + col];
                        trace(prims, mask, c, ray.x, ray.y, ray.z, ray.w, rgb);
                    }
                } else {
                    const float sx = (((float)col + 0.5f) / (float)rp.w * 2.0f - 1.0f) * c.tan_half_fov;
                    const float sy = (1.0f - ((float)row + 0.5f) / (float)rp.h * 2.0f) * c.tan_half_fov;
                    rgb = shade_pixel(prims, mask, c, sx, sy);
                }
                const int i = (row - row0) * rp.w + col;
                tile[3 * i] = (uint8_t)rgb; tile[3 * i + 1] = (uint8_t)(rgb >> 8); tile[3 * i + 2] = (uint8_t)(rgb >> 16);
            }
        }
        __syncthreads();
        if (rp.channels == 3 && ((count * 3) & 15) == 0 && (((int64_t)base * 3) & 15) == 0 && ((reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 *src = reinterpret_cast<const u32x4 *>(tile);
            u32x4 *dst = reinterpret_cast<u32x4 *>(out + (int64_t)base * 3);
            for (int i = threadIdx.x; i < count * 3 / 16; i += kRasterBlock) __builtin_nontemporal_store(src[i], dst + i);
        } else {
            for (int i = threadIdx.x; i < count; i += kRasterBlock) {
                uint8_t *d = out + (int64_t)(base + i) * rp.channels + 3 * cam;
                d[0] = tile[3 * i]; d[1] = tile[3 * i + 1]; d[2] = tile[3 * i + 2];
            }
        }
        __syncthreads();
    }
}

// ---- object-order path (cameras with a cached background) ------------------------------------------------------------------
// Screen footprint of one moving primitive in pixel units, conservative: rows r0..r1, columns c0..c1 and — for capsules — the
// slab around the projected axis: in row r the columns within hw of xc0 + (r - r0) * kx.  W bounds the columns of any one row.
struct Span {
    int32_t r0, r1, c0, c1, W;
    float xc0, kx, hw;
    int32_t rpc, cch, inv_w;       // chunking: row spans per 64-lane chunk, chunks per row (W > 64), ceil(65536 / W)
};

// what hit_capsule derives from the capsule and the eye alone, evaluated once per (env, camera) with the same expressions
struct CapsK { float bax, bay, baz, oax, oay, oaz, obx, oby, obz, baba, baoa, c, ca, cb; };

__device__ void capsule_constants(const Prim &p, const Camera &cam, CapsK &k) {
    k.bax = p.bx - p.ax; k.bay = p.by - p.ay; k.baz = p.bz - p.az;
    k.oax = cam.ex - p.ax; k.oay = cam.ey - p.ay; k.oaz = cam.ez - p.az;
    k.obx = cam.ex - p.bx; k.oby = cam.ey - p.by; k.obz = cam.ez - p.bz;
    k.baba = k.bax * k.bax + k.bay * k.bay + k.baz * k.baz;
    k.baoa = k.bax * k.oax + k.bay * k.oay + k.baz * k.oaz;
    const float oaoa = k.oax * k.oax + k.oay * k.oay + k.oaz * k.oaz;
    k.c = k.baba * oaoa - k.baoa * k.baoa - p.rad * p.rad * k.baba;
    k.ca = k.oax * k.oax + k.oay * k.oay + k.oaz * k.oaz - p.rad * p.rad;
    k.cb = k.obx * k.obx + k.oby * k.oby + k.obz * k.obz - p.rad * p.rad;
}

// hit_capsule with the per-capsule part taken from CapsK: same operations in the same order, so the same t
__device__ __forceinline__ float hit_capsule_k(const CapsK &k, float dx, float dy, float dz, int &code) {
    const float bard = k.bax * dx + k.bay * dy + k.baz * dz;
    const float rdoa = dx * k.oax + dy * k.oay + dz * k.oaz;
    const float a = k.baba - bard * bard;
    float b = k.baba * rdoa - k.baoa * bard;
    float h = b * b - a * k.c;
    if (h < 0.0f || k.baba == 0.0f) return -1.0f;
    float t = -1.0f, y = 0.0f;
    bool body = false;
    code = 0;
    if (a > 0.0f) {
        t = (-b - sqrtf(h)) / a;
        y = k.baoa + t * bard;
        body = y > 0.0f && y < k.baba;
    }
    if (!body) {            // one of the two end spheres
        const bool at_a = y <= 0.0f;
        b = dx * (at_a ? k.oax : k.obx) + dy * (at_a ? k.oay : k.oby) + dz * (at_a ? k.oaz : k.obz);
        h = b * b - (at_a ? k.ca : k.cb);
        if (h < 0.0f) return -1.0f;
        t = -b - sqrtf(h);
        code = at_a ? 1 : 2;
    }
    return t > 0.0f ? t : -1.0f;
}

// A capsule's surface point q = c + rad * n (c on the axis, |n| = 1) projects within rad * sqrt(1 + |c'|^2) / (Zc - rad) of
// c' = (Xc, Yc) / Zc (Cauchy-Schwarz on Zc * n_xy - n_z * (Xc, Yc)); c' runs along the 2-D segment a'b', |c'|^2 is convex on
// it and Zc is linear, so one radius R2 from the endpoints bounds the whole silhouette: a stadium around a'b'.
__device__ void prim_span(const Prim &p, const Camera &c, int w, int h, Span &s) {
    float rect[4];
    prim_screen_rect(p, c, rect);
    s.xc0 = 0.0f; s.kx = 0.0f; s.hw = 3.0e38f;
    const float half_w = 0.5f * (float)w, half_h = 0.5f * (float)h, inv_tan = 1.0f / c.tan_half_fov;
    bool slab = false;
    float xa = 0.0f, ya = 0.0f, xb = 0.0f, yb = 0.0f, R2 = 0.0f;
    if (p.type == PRIM_CAPSULE) {
        const float wax = p.ax - c.ex, way = p.ay - c.ey, waz = p.az - c.ez, wbx = p.bx - c.ex, wby = p.by - c.ey, wbz = p.bz - c.ez;
        const float za = wax * c.fx + way * c.fy + waz * c.fz, zb = wbx * c.fx + wby * c.fy + wbz * c.fz;
        const float zmin = fminf(za, zb) - p.rad;
        if (zmin > 0.05f) {
            xa = (wax * c.rx + way * c.ry + waz * c.rz) / za; ya = (wax * c.ux + way * c.uy + waz * c.uz) / za;
            xb = (wbx * c.rx + wby * c.ry + wbz * c.rz) / zb; yb = (wbx * c.ux + wby * c.uy + wbz * c.uz) / zb;
            R2 = p.rad * sqrtf(1.0f + fmaxf(xa * xa + ya * ya, xb * xb + yb * yb)) / zmin * 1.001f + 1.0e-3f;
            rect[0] = fmaxf(rect[0], fminf(xa, xb) - R2); rect[1] = fminf(rect[1], fmaxf(xa, xb) + R2);
            rect[2] = fmaxf(rect[2], fminf(ya, yb) - R2); rect[3] = fminf(rect[3], fmaxf(ya, yb) + R2);
            slab = true;
        }
    }
    // tangent space -> pixel index of the pixel whose centre is there: col = (x / tan + 1) * w / 2 - 1/2, row = (1 - y / tan) * h / 2 - 1/2
    const float cf0 = (rect[0] * inv_tan + 1.0f) * half_w - 0.5f, cf1 = (rect[1] * inv_tan + 1.0f) * half_w - 0.5f;
    const float rf0 = (1.0f - rect[3] * inv_tan) * half_h - 0.5f, rf1 = (1.0f - rect[2] * inv_tan) * half_h - 0.5f;
    s.c0 = (int)ceilf(fminf(fmaxf(cf0, 0.0f), (float)w)); s.c1 = (int)floorf(fminf(fmaxf(cf1, -1.0f), (float)(w - 1)));
    s.r0 = (int)ceilf(fminf(fmaxf(rf0, 0.0f), (float)h)); s.r1 = (int)floorf(fminf(fmaxf(rf1, -1.0f), (float)(h - 1)));
    s.W = max(s.c1 - s.c0 + 1, 0);
    if (slab && s.W > 0 && s.r1 >= s.r0) {
        const float dx = xb - xa, dy = yb - ya, len = sqrtf(dx * dx + dy * dy);
        if (fabsf(dy) > 1.0e-4f * len && len > 0.0f) {
            // the axis' column in row r: x = xa + (y_r - ya) * dx / dy with y_r = (1 - (r + 1/2) * 2 / h) * tan
            const float slope = dx / dy, px = half_w * inv_tan;
            const float y0 = (1.0f - ((float)s.r0 + 0.5f) / half_h) * c.tan_half_fov;
            s.xc0 = ((xa + (y0 - ya) * slope) * inv_tan + 1.0f) * half_w - 0.5f;
            s.kx = -slope * px * c.tan_half_fov / half_h;
            s.hw = R2 * len / fabsf(dy) * px + 0.01f;
            if (s.hw < 0.5f * (float)w) s.W = min(s.W, (int)floorf(2.0f * s.hw) + 2); else s.hw = 3.0e38f;
        }
    }
    const int W = max(s.W, 1);
    s.rpc = max(1, 64 / W); s.cch = (W + 63) >> 6; s.inv_w = W <= 64 ? (65536 + W - 1) / W : 0;
}

// One 256-lane workgroup per (env, cached camera).  The moving primitives are rasterised in object order: each one's footprint
// (Span) is cut into 64-pixel chunks — several short row spans side by side — that the workgroup's wavefronts share; a lane
// fetches its pixel's cached ray (the next chunk's load is in flight while this one is tested), runs the exact ray test of
// that one primitive (type and constants wave-uniform) and, when it is nearer than the static depth, atomic-mins
// (t, primitive, face) into a 64-bit LDS z-buffer — ties go to the lower primitive index, as in a sequential z-test.
// Resolve: pixels without a hit take the cached background colour, the others are gathered into a dense list and shaded
// (so the divergent normal code runs on full wavefronts); the colours replace the z-buffer words in place and the band is
// flushed 4 pixels = 12 bytes per lane.  Pixels no footprint covers are never ray-tested.
constexpr int kBandPixels = 2048;          // 16 KiB z-buffer + 4 KiB hit list: seven workgroups per CU
constexpr uint32_t kNoHit = 0xffffffffu;

struct Item { int k, row, col; bool valid; };

__global__ void __launch_bounds__(kRasterBlock)
raster_k(RasterParams rp, RasterKukaView kv, RasterMobileView mv, uint8_t *img) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ Prim prims[kMaxPrims];
    __shared__ Span spans[kMaxPrims];
    __shared__ CapsK capsk[kMaxPrims];
    __shared__ int32_t chunks[kMaxPrims];
    __shared__ int nprims, nhits;
    __shared__ __attribute__((aligned(16))) unsigned long long zbuf[kBandPixels];
    __shared__ uint16_t hits[kBandPixels];
    const int e = blockIdx.x, cam = blockIdx.y;
    const Camera c = rp.cam[cam];
    if (threadIdx.x == 0)
        nprims = rp.kind >= SRLHIP_ENV_KUKA_BUTTON ? build_kuka_scene(kv, e, prims) : build_mobile_scene(rp, mv, e, prims);
    __syncthreads();
    const int np = __builtin_amdgcn_readfirstlane(nprims), first = rp.nstatic, tid = threadIdx.x;
    if (tid >= first && tid < np) {
        prim_span(prims[tid], c, rp.w, rp.h, spans[tid]);
        if (prims[tid].type == PRIM_CAPSULE) capsule_constants(prims[tid], c, capsk[tid]);
    }
    const int npix = rp.h * rp.w;
    const float4 *__restrict__ rays = rp.rays[cam];
    const uint32_t *__restrict__ bg = rp.bg[cam];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t *out = img + (int64_t)e * npix * rp.channels;
    // bands: whole rows, a multiple of 4 pixels (the resolve and the flush work on groups of 4)
    const int band_rows = max(1, kBandPixels / rp.w);
    const bool quads = (rp.w & 3) == 0 && rp.w <= kBandPixels;
    const bool packed = quads && rp.channels == 3 && ((reinterpret_cast<uintptr_t>(out)) & 3) == 0;
    for (int row0 = 0; row0 < rp.h; row0 += band_rows) {
        const int rows = min(band_rows, rp.h - row0), base = row0 * rp.w, count = rows * rp.w, last_row = row0 + rows - 1;
        for (int i = tid; i < (count + 1) / 2; i += kRasterBlock) reinterpret_cast<u32x4 *>(zbuf)[i] = u32x4{kNoHit, kNoHit, kNoHit, kNoHit};
        if (tid == 0) nhits = 0;
        __syncthreads();                                                   // spans written (first band) / previous band flushed
        if (tid >= first && tid < np) {
            const Span &s = spans[tid];
            const int pr0 = max(s.r0, row0), pr1 = min(s.r1, last_row);
            chunks[tid] = (pr1 >= pr0 && s.W > 0) ? ((pr1 - pr0 + s.rpc) / s.rpc) * s.cch : 0;
        }
        __syncthreads();
        // ---- z-test, object order --------------------------------------------------------------------------------------
        int k = first, kbase = 0;
        int kcnt = k < np ? __builtin_amdgcn_readfirstlane(chunks[k]) : 0;
        auto locate = [&](int q) {                                         // chunk q of the band -> this lane's (primitive, pixel)
            Item it;
            while (k < np && q >= kbase + kcnt) { kbase += kcnt; k++; kcnt = k < np ? __builtin_amdgcn_readfirstlane(chunks[k]) : 0; }
            it.k = k; it.row = 0; it.col = 0; it.valid = false;
            if (k >= np) return it;
            const Span &s = spans[k];
            const int W = s.W, qq = q - kbase;
            int group = qq, cc = 0;
            if (s.cch > 1) { group = qq / s.cch; cc = qq - group * s.cch; }
            const int rr = (lane * s.inv_w) >> 16;                                      // lane / W (0 when W > 64)
            const int i = cc * 64 + lane - rr * W;
            it.row = max(s.r0, row0) + group * s.rpc + rr;
            const float centre = s.xc0 + (float)(it.row - s.r0) * s.kx;
            const int lo = s.hw < 1.0e30f ? max(s.c0, (int)ceilf(centre - s.hw)) : s.c0;
            const int hi = s.hw < 1.0e30f ? min(s.c1, (int)floorf(centre + s.hw)) : s.c1;
            it.col = lo + i;
            it.valid = rr < s.rpc && i < W && it.row <= min(s.r1, last_row) && it.col <= hi;
            return it;
        };
        Item cur = locate(wave);
        float4 ray = rays[cur.valid ? cur.row * rp.w + cur.col : 0];
        for (int q = wave; cur.k < np; q += kRasterBlock / 64) {
            const Item nxt = locate(q + kRasterBlock / 64);
            const float4 nray = rays[nxt.valid ? nxt.row * rp.w + nxt.col : 0];
            if (cur.valid) {
                int code;
                const Prim &p = prims[cur.k];
                const float t = p.type == PRIM_CAPSULE ? hit_capsule_k(capsk[cur.k], ray.x, ray.y, ray.z, code) : hit_prim(p, c, ray.x, ray.y, ray.z, code);
                if (t > 0.0f && t < ray.w)
                    atomicMin(&zbuf[(cur.row - row0) * rp.w + cur.col],
                              ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)((cur.k << 3) | code));
            }
            cur = nxt; ray = nray;
        }
        __syncthreads();
        // ---- resolve: the pixels with a hit are gathered into a dense list and shaded; the colour goes to the low word ------
        for (int i0 = 4 * tid; i0 < count; i0 += 4 * kRasterBlock) {
            bool hit[4];
            uint64_t m[4];
            int total = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                hit[u] = i0 + u < count && (uint32_t)zbuf[i0 + u] != kNoHit;
                m[u] = __ballot(hit[u]);
                total += (int)__popcll(m[u]);
            }
            if (total) {
                int at = 0;
                if (lane == 0) at = atomicAdd(&nhits, total);
                at = __builtin_amdgcn_readfirstlane(at);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (hit[u]) hits[at + (int)__popcll(m[u] & ((1ull << lane) - 1ull))] = (uint16_t)(i0 + u);
                    at += (int)__popcll(m[u]);
                }
            }
        }
        __syncthreads();
        for (int j = tid; j < nhits; j += kRasterBlock) {
            const int i = hits[j];
            const unsigned long long key = zbuf[i];
            const float4 r = rays[base + i];
            reinterpret_cast<uint32_t *>(&zbuf[i])[0] =
                shade_hit(prims[((uint32_t)key >> 3) & 31u], c, r.x, r.y, r.z, __uint_as_float((uint32_t)(key >> 32)), (int)((uint32_t)key & 7u));
        }
        __syncthreads();
        // ---- flush: 4 pixels = 12 bytes per lane; entries still all-ones take the cached background colour ---------------------
        if (packed) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(out + (int64_t)base * 3);
            const u32x4 *bg4 = reinterpret_cast<const u32x4 *>(bg + base);
            for (int j = tid; j < count / 4; j += kRasterBlock) {
                const u32x4 z01 = reinterpret_cast<const u32x4 *>(zbuf)[2 * j], z23 = reinterpret_cast<const u32x4 *>(zbuf)[2 * j + 1];
                const u32x4 b = bg4[j];
                const uint32_t p0 = z01.y != kNoHit ? z01.x : b.x, p1 = z01.w != kNoHit ? z01.z : b.y;
                const uint32_t p2 = z23.y != kNoHit ? z23.x : b.z, p3 = z23.w != kNoHit ? z23.z : b.w;
                __builtin_nontemporal_store(p0 | (p1 << 24), dst + 3 * j);
                __builtin_nontemporal_store((p1 >> 8) | (p2 << 16), dst + 3 * j + 1);
                __builtin_nontemporal_store((p2 >> 16) | (p3 << 8), dst + 3 * j + 2);
            }
        } else {
            for (int i = tid; i < count; i += kRasterBlock) {
                const unsigned long long key = zbuf[i];
                const uint32_t rgb = (uint32_t)(key >> 32) != kNoHit ? (uint32_t)key : bg[base + i];
                uint8_t *d = out + (int64_t)(base + i) * rp.channels + 3 * cam;
                d[0] = (uint8_t)rgb; d[1] = (uint8_t)(rgb >> 8); d[2] = (uint8_t)(rgb >> 16);
            }
        }
        __syncthreads();
    }
}

// Per-handle setup: unit ray, nearest static depth and shaded static colour of every pixel of fixed camera `cam`.  The
// static primitives are the first rp.nstatic ones of ANY env's scene (env 0 is used); same code path as raster_k's
// full trace, so the bytes are the ones the full path would produce.
__global__ void __launch_bounds__(kRasterBlock)
raster_bg_k(RasterParams rp, RasterKukaView kv, RasterMobileView mv, int cam, float4 *rays, uint32_t *bg) {
    __shared__ Prim prims[kMaxPrims];
    const Camera c = rp.cam[cam];
    if (threadIdx.x == 0) {
        if (rp.kind >= SRLHIP_ENV_KUKA_BUTTON) build_kuka_scene(kv, 0, prims); else build_mobile_scene(rp, mv, 0, prims);
    }
    __syncthreads();
    const int i = blockIdx.x * kRasterBlock + threadIdx.x;
    if (i >= rp.h * rp.w) return;
    const int row = i / rp.w, col = i - row * rp.w;
    const float sx = (((float)col + 0.5f) / (float)rp.w * 2.0f - 1.0f) * c.tan_half_fov;
    const float sy = (1.0f - ((float)row + 0.5f) / (float)rp.h * 2.0f) * c.tan_half_fov;
    float dx, dy, dz;
    pixel_ray(c, sx, sy, dx, dy, dz);
    // nearest static hit: the depth is kept beside the ray so that the per-env pass starts its z-test from it
    float best = 3.0e38f;
    int win = -1, wcode = 0;
    for (int k = 0; k < rp.nstatic; k++) {
        int code;
        const float t = hit_prim(prims[k], c, dx, dy, dz, code);
        if (t > 0.0f && t < best) { best = t; win = k; wcode = code; }
    }
    rays[i] = make_float4(dx, dy, dz, best);
    bg[i] = win >= 0 ? shade_hit(prims[win], c, dx, dy, dz, best, wcode) : finish_pixel(false, 0.0f, 0.0f, 1.0f, 0.92f, 0.92f, 0.92f);
}

// pybullet computeViewMatrixFromYawPitchRoll (upAxisIndex = 2): eye = target + Rz(yaw) Ry(roll) Rx(pitch) (0,-d,0),
// up = R (0,0,1); negative pitch looks down from above.
Camera make_camera(const double target[3], double dist, double yaw_deg, double pitch_deg, double roll_deg, double fov_deg) {
    const double d2r = 3.14159265358979323846 / 180.0;
    const double cy = cos(yaw_deg * d2r), sy = sin(yaw_deg * d2r), cp = cos(pitch_deg * d2r), sp = sin(pitch_deg * d2r);
    const double cr = cos(roll_deg * d2r), sr = sin(roll_deg * d2r);
    // R = Rz(yaw) * Ry(roll) * Rx(pitch)
    const double Rm[3][3] = {{cy * cr, cy * sr * sp - sy * cp, cy * sr * cp + sy * sp},
                             {sy * cr, sy * sr * sp + cy * cp, sy * sr * cp - cy * sp},
                             {-sr, cr * sp, cr * cp}};
    double eye[3], up[3], f[3], r[3], u[3];
    for (int k = 0; k < 3; k++) { eye[k] = target[k] + Rm[k][1] * (-dist); up[k] = Rm[k][2]; f[k] = target[k] - eye[k]; }
    double nf = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (int k = 0; k < 3; k++) f[k] /= nf;
    r[0] = f[1] * up[2] - f[2] * up[1]; r[1] = f[2] * up[0] - f[0] * up[2]; r[2] = f[0] * up[1] - f[1] * up[0];
    double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (int k = 0; k < 3; k++) r[k] /= nr;
    u[0] = r[1] * f[2] - r[2] * f[1]; u[1] = r[2] * f[0] - r[0] * f[2]; u[2] = r[0] * f[1] - r[1] * f[0];
    Camera c;
    c.ex = (float)eye[0]; c.ey = (float)eye[1]; c.ez = (float)eye[2];
    c.fx = (float)f[0]; c.fy = (float)f[1]; c.fz = (float)f[2];
    c.rx = (float)r[0]; c.ry = (float)r[1]; c.rz = (float)r[2];
    c.ux = (float)u[0]; c.uy = (float)u[1]; c.uz = (float)u[2];
    c.tan_half_fov = (float)tan(fov_deg * d2r / 2);
    return c;
}

}  // namespace

// views of the env state, provided by mobile.hip / kuka.hip
void kuka_raster_view(Handle *h, RasterKukaView *v);

int raster_render(Handle *h, void *d_img) {
    RasterParams rp;
    const srlhip_config &c = h->cfg;
    rp.kind = c.env_kind; rp.n = h->n; rp.h = c.img_h; rp.w = c.img_w;
    rp.ncam = c.multi_view ? 2 : 1;
    rp.fpv = (c.env_kind < SRLHIP_ENV_KUKA_BUTTON && c.multi_view) ? 1 : 0;
    rp.channels = 3 * rp.ncam;
    RasterKukaView kv = {};
    RasterMobileView mv = {};
    if (c.env_kind >= SRLHIP_ENV_KUKA_BUTTON) {
        const double t1[3] = {0.316, -0.2, -0.1}, t2[3] = {0.316, 0.316, -0.105};
        rp.cam[0] = make_camera(t1, 1.1, 145, -36, 0, 60);        // kuka_button_gym_env.py:94-102
        rp.cam[1] = make_camera(t2, 1.05, 32, -13, 0, 60);        // :403-409 (multi_view)
        kuka_raster_view(h, &kv);
    } else {
        const double t[3] = {2, c.env_kind == SRLHIP_ENV_MOBILE_1D ? 0.0 : 2.0, 0};
        rp.cam[0] = make_camera(t, 4.4, 90, -90, 0, 60);          // mobile_robot_env.py:76-84, 1D :33
        // fpv=True (mobile_robot_env.py:313-332): camera target (robot_x - 0.25, robot_y, 0.15), distance 0.3, yaw = the
        // env's camera yaw, pitch -17, fov 90; stored relative to the robot, the kernel adds each env's (x, y)
        const double tf[3] = {-0.25, 0.0, 0.15};
        rp.cam[1] = make_camera(tf, 0.3, 90, -17, 0, 90);
        const MobileState &s = h->mobile;
        mv.x = s.pos_x; mv.y = s.pos_y; mv.tx = s.tgt_x; mv.ty = s.tgt_y; mv.t2x = s.tgt2_x; mv.t2y = s.tgt2_y; mv.cur = s.cur_target;
    }
    // static scenery: Kuka = floor plane + table; MobileRobot = floor plane + arena walls (1 in the 1-D env, 4 otherwise)
    rp.nstatic = c.env_kind >= SRLHIP_ENV_KUKA_BUTTON ? 2 : (c.env_kind == SRLHIP_ENV_MOBILE_1D ? 2 : 5);
    const int ncached = rp.fpv ? 1 : rp.ncam;                  // the fpv camera moves with the robot
    for (int cam = 0; cam < 2; cam++) { rp.rays[cam] = nullptr; rp.bg[cam] = nullptr; }
    if (kv.has_tm) {
        // full Kuka model: the gripper's capsule end points of every env, before anything builds a scene (raster_grip_k)
        if (!h->raster_grip) {
            int rc;
            if ((rc = h->dalloc(&h->raster_grip, (size_t)42 * h->n))) return rc;
        }
        kv.grip = h->raster_grip;
        hipLaunchKernelGGL(raster_grip_k, dim3((h->n + 63) / 64), dim3(64), 0, h->stream, kv, h->raster_grip);
        SRL_HIP_CHECK(h, hipGetLastError());
    }
    for (int cam = 0; cam < ncached; cam++) {
        if (!h->raster_rays[cam]) {
            const size_t npix = (size_t)rp.h * rp.w;
            int rc;
            if ((rc = h->dalloc(&h->raster_rays[cam], npix)) || (rc = h->dalloc(&h->raster_bg[cam], npix))) return rc;
            hipLaunchKernelGGL(raster_bg_k, dim3((unsigned)((npix + kRasterBlock - 1) / kRasterBlock)), dim3(kRasterBlock), 0, h->stream,
                               rp, kv, mv, cam, h->raster_rays[cam], h->raster_bg[cam]);
            SRL_HIP_CHECK(h, hipGetLastError());
        }
        rp.rays[cam] = h->raster_rays[cam];
        rp.bg[cam] = h->raster_bg[cam];
    }
    rp.cam0 = 0;
    hipLaunchKernelGGL(raster_k, dim3(h->n, ncached), dim3(kRasterBlock), 0, h->stream, rp, kv, mv, static_cast<uint8_t *>(d_img));
    SRL_HIP_CHECK(h, hipGetLastError());
    if (ncached < rp.ncam) {
        rp.cam0 = ncached;
        hipLaunchKernelGGL(raster_tiles_k, dim3(h->n, rp.ncam - ncached), dim3(kRasterBlock), 0, h->stream, rp, kv, mv, static_cast<uint8_t *>(d_img));
        SRL_HIP_CHECK(h, hipGetLastError());
    }
    return 0;
}

}  // namespace srl
