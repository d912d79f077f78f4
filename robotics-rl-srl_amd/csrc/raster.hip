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
// Mapping: one 256-lane workgroup per (env, camera).  The env's scene — at most 16
// analytic primitives (plane, z-rotated box, upright cylinder, capsule) — is built
// once per workgroup into LDS from the SoA state (for the Kuka: float64 forward
// kinematics from the cached joint sin/cos), then every lane ray-casts its pixels
// against the MOVING primitives whose projected bounding box overlaps its wavefront's 8x8
// tile (one ballot per tile; nearest hit = z-buffer in registers) and parks 3 bytes per
// pixel in an LDS band buffer that is flushed with coalesced 16-byte stores:
// the path's HBM traffic is the 12 288-byte image per env and nothing else.
// The cameras are fixed and the first primitives of a scene (floor plane, table / arena walls)
// never move, so a per-handle setup kernel (raster_bg_k) stores for every pixel its unit ray,
// the depth of the nearest static hit and the shaded static colour: a tile no moving primitive
// can touch is a copy of that background, and the other tiles start their z-test from it —
// same arithmetic per primitive, same bytes, without normalising the ray and re-hitting the
// floor for every env.  (The MobileRobot fpv camera rides on the robot: it takes the full path.)
// float32 throughout, -ffp-contract=off so that the C oracle (oracle/raster_oracle.c)
// reproduces the bytes.
#include "internal.hpp"
#include "kuka_core.hpp"

namespace srl {

// KukaState / planes are private to kuka.hip; the rasteriser gets raw plane pointers.
struct RasterKukaView { const double *sq, *cq, *bq, *bx, *by, *bz, *b2q, *b2x, *b2y, *objs; int64_t n; int32_t two, rand_objects; };     // sq/cq: [7][n]
struct RasterMobileView { const double *x, *y, *tx, *ty, *t2x, *t2y; const int32_t *cur; };

namespace {

constexpr int kRasterBlock = 256;
constexpr int kMaxPrims = 28;   // Kuka scene 14 + second button 2 or ten distractors + ball 11
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
// The hit functions return t only, plus what the normal of THIS hit needs later (aux, code): normals are evaluated once, for
// the nearest hit (prim_normal), not for every candidate — their divisions were a third of a capsule test.
__device__ __forceinline__ float hit_plane(const Prim &p, float oz, float dz) {
    if (dz == 0.0f) return -1.0f;
    const float t = (p.az - oz) / dz;
    return t > 0.0f ? t : -1.0f;
}

__device__ __forceinline__ float hit_box(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz,
                                         float &aux, int &code) {
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
    aux = sign; code = axis;
    return tmin;
}

__device__ __forceinline__ float hit_cylinder(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz,
                                              float &aux, int &code) {
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

__device__ __forceinline__ float hit_capsule(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz,
                                             float &aux, int &code) {
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
        y = y <= 0.0f ? 0.0f : baba;
    }
    if (t <= 0.0f) return -1.0f;
    aux = y; code = 0;
    return t;
}

// normal of primitive p where the ray (eye o, unit direction d) hits it at parameter t; aux / code as left by its hit function
__device__ __forceinline__ void prim_normal(const Prim &p, float ox, float oy, float oz, float dx, float dy, float dz, float t, float aux,
                                            int code, float &nx, float &ny, float &nz) {
    if (p.type == PRIM_PLANE) { nx = 0.0f; ny = 0.0f; nz = 1.0f; }
    else if (p.type == PRIM_BOX) {
        const float lnx = code == 0 ? aux : 0.0f, lny = code == 1 ? aux : 0.0f;
        nx = p.cs * lnx - p.sn * lny; ny = p.sn * lnx + p.cs * lny; nz = code == 2 ? aux : 0.0f;
    } else if (p.type == PRIM_CYL) {
        if (code == 0) { nx = (ox - p.ax + t * dx) / p.bx; ny = (oy - p.ay + t * dy) / p.bx; nz = 0.0f; }
        else { nx = 0.0f; ny = 0.0f; nz = dz < 0.0f ? 1.0f : -1.0f; }
    } else {
        const float bax = p.bx - p.ax, bay = p.by - p.ay, baz = p.bz - p.az;
        const float oax = ox - p.ax, oay = oy - p.ay, oaz = oz - p.az;
        const float baba = bax * bax + bay * bay + baz * baz;
        const float k = aux / baba;
        nx = (oax + t * dx - bax * k) / p.rad; ny = (oay + t * dy - bay * k) / p.rad; nz = (oaz + t * dz - baz * k) / p.rad;
    }
}

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

__device__ int build_kuka_scene(const RasterKukaView &v, int e, Prim *prims) {
    using namespace kuka;
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {kBasePos[0], kBasePos[1], kBasePos[2]};
    float jp[ND][3];
    const int64_t n = v.n;
#define SRL_FK(I) { fk_forward<I>(R, p, v.sq[(I) * n + e], v.cq[(I) * n + e]); jp[I][0] = (float)p[0]; jp[I][1] = (float)p[1]; jp[I][2] = (float)p[2]; }
    SRL_FK(0) SRL_FK(1) SRL_FK(2) SRL_FK(3) SRL_FK(4) SRL_FK(5) SRL_FK(6)
#undef SRL_FK
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
    const double body[3] = {0, 0, 0.10}, fa0[3] = {0, 0.030, 0.10}, fa1[3] = {0, 0.020, 0.255}, fb0[3] = {0, -0.030, 0.10},
                 fb1[3] = {0, -0.020, 0.255};
    double a[3], b[3];
    tip_point(R, p, body, a);
    set_prim(prims[k++], PRIM_CAPSULE, 0.20f, 0.20f, 0.22f, jp[6][0], jp[6][1], jp[6][2], (float)a[0], (float)a[1], (float)a[2], 0.045f, 1, 0);
    tip_point(R, p, fa0, a); tip_point(R, p, fa1, b);
    set_prim(prims[k++], PRIM_CAPSULE, 0.10f, 0.10f, 0.10f, (float)a[0], (float)a[1], (float)a[2], (float)b[0], (float)b[1], (float)b[2], 0.015f, 1, 0);
    tip_point(R, p, fb0, a); tip_point(R, p, fb1, b);
    set_prim(prims[k++], PRIM_CAPSULE, 0.10f, 0.10f, 0.10f, (float)a[0], (float)a[1], (float)a[2], (float)b[0], (float)b[1], (float)b[2], 0.015f, 1, 0);
    if (v.rand_objects) {
        // KukaRandButtonGymEnv scenery (kuka_rand_button_gym_env.py:60-71): kept distractors resting on the table, and the
        // ball at its drop position.  The reference draws the object TYPE from the global unseeded np.random; here it is a
        // hash of the position: 0 duck (yellow blob), 1 lego (small red brick), 2 cube_small (5 cm cube).
        const float top = (float)kTableTopZ;
        for (int i = 0; i < 10; i++) {
            const double ox = v.objs[(3 * i) * n + e], oy = v.objs[(3 * i + 1) * n + e];
            if (v.objs[(3 * i + 2) * n + e] == 0.0) continue;
            const uint32_t type = (uint32_t)(((uint64_t)__double_as_longlong(ox) >> 20) ^ ((uint64_t)__double_as_longlong(oy) >> 20)) % 3u;
            const float x = (float)ox, y = (float)oy;
            if (type == 0) set_prim(prims[k++], PRIM_CAPSULE, 1.0f, 0.85f, 0.1f, x - 0.015f, y, top + 0.035f, x + 0.015f, y, top + 0.035f, 0.035f, 1, 0);
            else if (type == 1) set_prim(prims[k++], PRIM_BOX, 0.8f, 0.1f, 0.1f, x, y, top + 0.012f, 0.016f, 0.032f, 0.012f, 0, 1.0f, 0.0f);
            else set_prim(prims[k++], PRIM_BOX, 0.9f, 0.9f, 0.9f, x, y, top + 0.025f, 0.025f, 0.025f, 0.025f, 0, 1.0f, 0.0f);
        }
        set_prim(prims[k++], PRIM_CAPSULE, 0.9f, 0.2f, 0.2f, 0.25f, -0.2f, top + 0.03f, 0.25f, -0.2f, top + 0.031f, 0.03f, 1, 0);   // sphere_small
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

// z-test of the unit ray (dx, dy, dz) from the eye against the primitives in `mask` (wave-uniform), starting from `best`
__device__ __forceinline__ bool trace(const Prim *prims, uint64_t mask, const Camera &c, float dx, float dy, float dz, float &best,
                                      float &bnx, float &bny, float &bnz, float &cr, float &cg, float &cb) {
    int win = -1, wcode = 0;
    float waux = 0.0f;
    while (mask) {                                   // wave-uniform list of the primitives that can touch this tile
        const int k = __builtin_ctzll(mask);
        mask &= mask - 1;
        const Prim &p = prims[k];
        float t, aux = 0.0f;
        int code = 0;
        if (p.type == PRIM_PLANE) t = hit_plane(p, c.ez, dz);
        else if (p.type == PRIM_BOX) t = hit_box(p, c.ex, c.ey, c.ez, dx, dy, dz, aux, code);
        else if (p.type == PRIM_CYL) t = hit_cylinder(p, c.ex, c.ey, c.ez, dx, dy, dz, aux, code);
        else t = hit_capsule(p, c.ex, c.ey, c.ez, dx, dy, dz, aux, code);
        if (t > 0.0f && t < best) { best = t; win = k; waux = aux; wcode = code; }
    }
    if (win < 0) return false;
    const Prim &p = prims[win];
    prim_normal(p, c.ex, c.ey, c.ez, dx, dy, dz, best, waux, wcode, bnx, bny, bnz);
    cr = p.r; cg = p.g; cb = p.b;
    if (p.type == PRIM_PLANE) {
        // plane.urdf's texture: 1 m blue/white checker aligned with the world axes (period, phase and the two
        // colours measured on the reference's imgs/mobile_robot.gif: x in [0,1) x y in [0,1) is white)
        const int par = (int)floorf(c.ex + best * dx) + (int)floorf(c.ey + best * dy);
        if ((par & 1) == 0) { cr = 1.0f; cg = 1.0f; cb = 1.0f; }
    }
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
    float best = 3.0e38f, bnx = 0.0f, bny = 0.0f, bnz = 1.0f, cr = 0.92f, cg = 0.92f, cb = 0.92f;     // background
    const bool hit = trace(prims, mask, c, dx, dy, dz, best, bnx, bny, bnz, cr, cg, cb);
    return finish_pixel(hit, bnx, bny, bnz, cr, cg, cb);
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

__global__ void __launch_bounds__(kRasterBlock)
raster_k(RasterParams rp, RasterKukaView kv, RasterMobileView mv, uint8_t *img) {
    __shared__ Prim prims[kMaxPrims];
    __shared__ float rects[kMaxPrims][4];
    __shared__ int nprims;
    __shared__ uint32_t tile_masks[kTilePixels / 64];                     // per 8x8 tile of the current band: primitives whose rectangle overlaps it
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTilePixels * 3];
    const int e = blockIdx.x, cam = blockIdx.y;
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
                        float best = ray.w, bnx = 0.0f, bny = 0.0f, bnz = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
                        if (trace(prims, mask, c, ray.x, ray.y, ray.z, best, bnx, bny, bnz, cr, cg, cb))
                            rgb = finish_pixel(true, bnx, bny, bnz, cr, cg, cb);
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
    float best = 3.0e38f, bnx = 0.0f, bny = 0.0f, bnz = 1.0f, cr = 0.92f, cg = 0.92f, cb = 0.92f;
    const bool hit = trace(prims, (1ull << rp.nstatic) - 1ull, c, dx, dy, dz, best, bnx, bny, bnz, cr, cg, cb);
    rays[i] = make_float4(dx, dy, dz, best);
    bg[i] = finish_pixel(hit, bnx, bny, bnz, cr, cg, cb);
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
    hipLaunchKernelGGL(raster_k, dim3(h->n, rp.ncam), dim3(kRasterBlock), 0, h->stream, rp, kv, mv, static_cast<uint8_t *>(d_img));
    SRL_HIP_CHECK(h, hipGetLastError());
    return 0;
}

}  // namespace srl
