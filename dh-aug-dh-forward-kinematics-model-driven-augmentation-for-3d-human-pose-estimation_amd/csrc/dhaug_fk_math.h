// Per-pose DH forward-kinematics arithmetic shared by the forward and backward kernels.
//
// One lane owns one pose and keeps the whole skeleton in registers: 36 sin/cos pairs, 30 affine frame
// updates (the arm chains start from the body frame at index 8 instead of re-multiplying the nine-matrix
// body prefix: bit-identical), 15 global-rotation mat-vecs.  No dependence between lanes.
//
// Reference arithmetic that is reproduced on purpose (R/ = DH-AUG_master/):
//   * degrees -> radians as fp32(fp32(x / 180) * fp32(pi))   R/models_Fk_GAN/forward_kinematics_DH_model.py:89-90
//   * cos(alpha = +-90 deg) = -4.371139e-08 (not 0), sin = +-1; the tiny terms are kept (kEps90)
//   * modified-DH layout of dh_matrix (:99-114); chains multiplied left to right (:659-677)
//   * global rotation Rx*Ry*Rz applied to the chain translations (:706-743), root added last (:819-820)
#pragma once
#include "dhaug_common.h"

#define DHAUG_HD __host__ __device__ __forceinline__

namespace dhaug_fk {

constexpr float kPiF     = 3.14159274101257324f;        // fp32(pi)
constexpr float kInv180  = 0.0055555556900799274f;      // fp32(1/180)
constexpr float kEps90   = -4.371138828673793e-08f;     // cosf(fp32(pi/2)) as ATen/CPU evaluates it
constexpr float kDeg2Rad = 0.017453292519943295f;

struct V3 { float x, y, z; };

DHAUG_HD V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
DHAUG_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DHAUG_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DHAUG_HD V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DHAUG_HD V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// r + s*a
DHAUG_HD V3 axpy(float s, V3 a, V3 r) { return mk(fmaf(s, a.x, r.x), fmaf(s, a.y, r.y), fmaf(s, a.z, r.z)); }
DHAUG_HD float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
DHAUG_HD V3 cross(V3 a, V3 b) {
    return mk(fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x));
}

// x / 180 correctly rounded without the v_div sequence: one Newton correction in FMA.
DHAUG_HD float div180(float x) {
    float q = x * kInv180;
    float e = fmaf(-q, 180.0f, x);
    return fmaf(e, kInv180, q);
}

// sin/cos of an fp32 radian argument, ~1 ulp: 3-term Cody-Waite reduction by pi/2 in FMA, Cephes minimax
// polynomials on [-pi/4, pi/4].  ~22 VALU instructions per pair (ocml sincosf is ~3x that and branches).
DHAUG_HD void sincos_rad_bounded(float x, float& s, float& c) {
    float kf = rintf(x * 0.6366197466850281f);
    float r = fmaf(-kf, 1.5707963705062866f, x);
    r = fmaf(-kf, -4.371138828673793e-08f, r);
    r = fmaf(-kf, -1.7151245100058819e-15f, r);
    int k = (int)kf;
    float z = r * r;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                    z * z, fmaf(-0.5f, z, 1.0f));
    float ss = (k & 1) ? cp : sp;
    float cc = (k & 1) ? sp : cp;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
}
DHAUG_HD void sincos_rad(float x, float& s, float& c) {
    // far outside any joint angle, or not finite: library path.  The test is on the bit pattern (131072.0f = 0x48000000;
    // inf / NaN compare greater): this file is built with -ffinite-math-only, under which a float comparison may be
    // assumed false for NaN -- and the (int) conversion below is undefined for NaN (found by UBSan on the host build,
    // tests/test_cpu_boundary.py).
    if (__builtin_expect((__builtin_bit_cast(uint32_t, x) & 0x7fffffffu) > 0x48000000u, 0)) {
        sincosf(x, &s, &c);
        return;
    }
    sincos_rad_bounded(x, s, c);
}

// sin/cos of an angle given in degrees, with the reference's fp32 conversion.
DHAUG_HD void sincos_deg(float deg, float& s, float& c) {
    sincos_rad(div180(deg) * kPiF, s, c);
}

// Cumulative frame of a chain: rotation columns c0,c1,c2 and translation t (world = R*local + t).
struct Frame { V3 c0, c1, c2, t; };

// ALPHA: 0 -> alpha = 0;  +1 -> +90 deg;  -1 -> -90 deg.
template <int ALPHA, bool EPS = true> DHAUG_HD constexpr float cos_alpha() { return ALPHA == 0 ? 1.0f : (EPS ? kEps90 : 0.0f); }
template <int ALPHA> DHAUG_HD constexpr float sin_alpha() { return (float)ALPHA; }

// First joint of a chain (cumulative = local matrix).
template <int ALPHA, bool EPS = true>
DHAUG_HD Frame dh_first(float a, float d, float st, float ct) {
    constexpr float ca = cos_alpha<ALPHA, EPS>(), sa = sin_alpha<ALPHA>();
    Frame F;
    F.c0 = mk(ct, st * ca, st * sa);
    F.c1 = mk(-st, ct * ca, ct * sa);
    F.c2 = mk(0.0f, -sa, ca);
    F.t = mk(a, -sa * d, ca * d);
    return F;
}

// F <- F * DH(alpha, a, d, theta).  ROT=false skips the rotation update (leaf joints: only t is read).
template <int ALPHA, bool HAS_A, bool HAS_D, bool ROT, bool EPS = true>
DHAUG_HD void dh_apply(Frame& F, float a, float d, float st, float ct) {
    constexpr float ca = cos_alpha<ALPHA, EPS>(), sa = sin_alpha<ALPHA>();
    // translation: t += R * (a, -sa*d, ca*d)
    if (HAS_A) F.t = axpy(a, F.c0, F.t);
    if (HAS_D) {
        if (ALPHA != 0) F.t = axpy(-sa * d, F.c1, F.t);
        F.t = axpy(ca * d, F.c2, F.t);
    }
    if (ROT) {
        V3 c0 = F.c0, c1 = F.c1, c2 = F.c2;
        if (ALPHA == 0) {
            F.c0 = axpy(st, c1, ct * c0);
            F.c1 = axpy(ct, c1, (-st) * c0);
        } else {
            float sca = st * ca, cca = ct * ca, ssa = st * sa, csa = ct * sa;
            F.c0 = axpy(ssa, c2, axpy(sca, c1, ct * c0));
            F.c1 = axpy(csa, c2, axpy(cca, c1, (-st) * c0));
            F.c2 = axpy(ca, c2, (-sa) * c1);
        }
    }
}

// Rg = Rx(ax) * Ry(ay) * Rz(az), rows r0,r1,r2.   R/models_Fk_GAN/forward_kinematics_DH_model.py:141-191
struct Rot { V3 r0, r1, r2; };
DHAUG_HD Rot global_rot_sc(float sx, float cx, float sy, float cy, float sz, float cz) {
    // M = Rx*Ry
    V3 m0 = mk(cy, 0.0f, sy);
    V3 m1 = mk(sx * sy, cx, -sx * cy);
    V3 m2 = mk(-cx * sy, sx, cx * cy);
    Rot R;
    R.r0 = mk(m0.x * cz, -m0.x * sz, m0.z);
    R.r1 = mk(fmaf(m1.y, sz, m1.x * cz), fmaf(m1.y, cz, -m1.x * sz), m1.z);
    R.r2 = mk(fmaf(m2.y, sz, m2.x * cz), fmaf(m2.y, cz, -m2.x * sz), m2.z);
    return R;
}
DHAUG_HD Rot global_rot(float ax, float ay, float az, float& sx, float& cx) {
    float sy, cy, sz, cz;
    sincos_deg(ax, sx, cx); sincos_deg(ay, sy, cy); sincos_deg(az, sz, cz);
    return global_rot_sc(sx, cx, sy, cy, sz, cz);
}
DHAUG_HD V3 rot_apply(const Rot& R, V3 p) { return mk(dot(R.r0, p), dot(R.r1, p), dot(R.r2, p)); }
DHAUG_HD V3 rot_apply_t(const Rot& R, V3 g) {      // R^T * g
    return axpy(g.z, R.r2, axpy(g.y, R.r1, g.x * R.r0));
}

// ---- critic-side features of a 16-joint pose (same arithmetic as dhaug_pose.hip's standalone kernels) ----
// bone i = joint[c] - joint[p], used_16key_15bone_len_table (R/models_Fk_GAN/forward_kinematics_DH_model.py:46-49)
DHAUG_HD constexpr int kcs_bone_p(int i) { constexpr int t[15] = {5, 2, 4, 1, 0, 0, 0, 7, 8, 8, 10, 13, 11, 14, 8}; return t[i]; }
DHAUG_HD constexpr int kcs_bone_c(int i) { constexpr int t[15] = {6, 3, 5, 2, 4, 1, 7, 8, 10, 13, 11, 14, 12, 15, 9}; return t[i]; }
// KCS cosine pairs (R/models_Fk_GAN/Fk_discriminator.py:81-140)
DHAUG_HD constexpr int kcs_i(int k) { constexpr int t[15] = {0, 1, 2, 3, 4, 4, 5, 6, 7, 7, 7, 8, 9, 10, 11}; return t[k]; }
DHAUG_HD constexpr int kcs_j(int k) { constexpr int t[15] = {2, 3, 4, 5, 5, 6, 6, 7, 14, 8, 9, 10, 11, 12, 13}; return t[k]; }

// 15 cosines between adjacent bones, then the 15 bone lengths (special_KCS_Input_transform)
DHAUG_HD void kcs_features(const V3* __restrict__ p, float* __restrict__ f) {
    V3 b[15];
    float len[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) { b[i] = p[kcs_bone_c(i)] - p[kcs_bone_p(i)]; len[i] = sqrtf(dot(b[i], b[i])); }
#pragma unroll
    for (int k = 0; k < 15; ++k) f[k] = dot(b[kcs_i(k)], b[kcs_j(k)]) / (len[kcs_i(k)] * len[kcs_j(k)]);
#pragma unroll
    for (int i = 0; i < 15; ++i) f[15 + i] = len[i];
}

// world -> camera by the inverse of quaternion q = (w, x, y, z) after subtracting t, then the H36M projection with
// c = f(2) c(2) k(3) p(2)  (R/common/camera.py:36-38, 62-94; R/common/quaternion.py:6-24)
DHAUG_HD void w2c_project(V3 xw, const float* __restrict__ q, const float* __restrict__ t, const float* __restrict__ c,
                          float& ox, float& oy) {
    const V3 x = mk(xw.x - t[0], xw.y - t[1], xw.z - t[2]);
    const V3 qv = mk(-q[1], -q[2], -q[3]);
    const V3 uv = cross(qv, x);
    const V3 uuv = cross(qv, uv);
    const V3 xc = x + 2.0f * (q[0] * uv + uuv);
    const float u = dhaug_clamp_pm1(xc.x / xc.z), v = dhaug_clamp_pm1(xc.y / xc.z);
    const float r2 = u * u + v * v;
    const float radial = 1.0f + (c[4] * r2 + c[5] * (r2 * r2) + c[6] * (r2 * r2 * r2));
    const float tan = c[7] * u + c[8] * v;
    ox = c[0] * (u * (radial + tan) + c[7] * r2) + c[2];
    oy = c[1] * (v * (radial + tan) + c[8] * r2) + c[3];
}

// theta0 tables (degrees), R/models_Fk_GAN/forward_kinematics_DH_model.py:234-261
//   right leg  alpha[0,-90,-90,0,0]  theta0[0,-90,180,0,0]     a[+hipR,0,0,thighR,shinR]
//   left  leg  alpha[0,+90,+90,0,0]  theta0[180,-90,0,0,0]     a[-hipL,0,0,thighL,shinL]
//   body (13)  alpha[0,-90 x11,+90]  theta0[90,-90 x10,0,0]    d[3]=waist d[6]=thorax a[12]=neck
//   right arm  alpha[-90,-90,-90,0,0] theta0[-180,-90,180,0,0] a[-shR,0,0,uarmR,farmR]   (starts at body frame 8)
//   left  arm  alpha[-90,+90,+90,0,0] theta0[0,-90,0,0,0]      a[+shL,0,0,uarmL,farmL]   (starts at body frame 8)
// bone_len order (used_16key_15bone_len_table): 0 l-shin 1 r-shin 2 l-thigh 3 r-thigh 4 l-hip 5 r-hip 6 waist
//   7 thorax 8 l-shoulder 9 r-shoulder 10 l-upper-arm 11 r-upper-arm 12 l-forearm 13 r-forearm 14 neck
// 16-joint output order: 0 Hip 1 RHip 2 RKnee 3 RAnkle 4 LHip 5 LKnee 6 LAnkle 7 Spine 8 Thorax 9 Head
//   10 LShoulder 11 LElbow 12 LWrist 13 RShoulder 14 RElbow 15 RWrist

// GUARD = false: the caller bounds the angle (the generator tail's angles are tanh * range), no library path
template <bool GUARD> DHAUG_HD void sincos_deg_t(float deg, float& s, float& c) {
    if (GUARD) sincos_deg(deg, s, c);
    else sincos_rad_bounded(div180(deg) * kPiF, s, c);
}
#define DHAUG_SC(idx, theta0) float s##idx, c##idx; sincos_deg_t<GUARD>((theta0) + ang[idx], s##idx, c##idx)

// The skeleton chain by chain (one wave per chain in the generator-tail kernel, all of them in fk_pose): joints are written
// root-free, globally rotated.  Each piece reads only its own slots of ang[37] / bl[15].
template <bool GUARD = true>
DHAUG_HD Rot fk_global(const float* __restrict__ ang, float& sgx, float& cgx) {
    float sy, cy, sz, cz;
    sincos_deg_t<GUARD>(ang[34], sgx, cgx); sincos_deg_t<GUARD>(ang[35], sy, cy); sincos_deg_t<GUARD>(ang[36], sz, cz);
    return global_rot_sc(sgx, cgx, sy, cy, sz, cz);
}
// (the *_sc forms take the sines / cosines of the chain's angles: the generator-tail kernel computes them before the bone
// lengths and the global rotation arrive)
DHAUG_HD void fk_right_leg_sc(const float* __restrict__ s, const float* __restrict__ c, const float* __restrict__ bl, const Rot& Rg,
                              V3* __restrict__ p) {
    Frame F = dh_first<0>(bl[5], 0.0f, s[0], c[0]);
    p[1] = rot_apply(Rg, F.t);
    dh_apply<-1, false, false, true>(F, 0.0f, 0.0f, s[1], c[1]);
    dh_apply<-1, false, false, true>(F, 0.0f, 0.0f, s[2], c[2]);
    dh_apply<0, true, false, true>(F, bl[3], 0.0f, s[3], c[3]);
    p[2] = rot_apply(Rg, F.t);
    dh_apply<0, true, false, false>(F, bl[1], 0.0f, 0.0f, 1.0f);
    p[3] = rot_apply(Rg, F.t);
}
DHAUG_HD void fk_left_leg_sc(const float* __restrict__ s, const float* __restrict__ c, const float* __restrict__ bl, const Rot& Rg,
                             V3* __restrict__ p) {
    Frame F = dh_first<0>(-bl[4], 0.0f, s[0], c[0]);
    p[4] = rot_apply(Rg, F.t);
    dh_apply<1, false, false, true>(F, 0.0f, 0.0f, s[1], c[1]);
    dh_apply<1, false, false, true>(F, 0.0f, 0.0f, s[2], c[2]);
    dh_apply<0, true, false, true>(F, bl[2], 0.0f, s[3], c[3]);
    p[5] = rot_apply(Rg, F.t);
    dh_apply<0, true, false, false>(F, bl[0], 0.0f, 0.0f, 1.0f);
    p[6] = rot_apply(Rg, F.t);
}
// body frames 0..8 (angles 10..18): Spine = frame 3, Thorax = frame 6; returns frame 8, where the head and the arms start
DHAUG_HD Frame fk_body_sc(const float* __restrict__ s, const float* __restrict__ c, const float* __restrict__ bl, const Rot& Rg,
                          V3* __restrict__ p) {
    Frame B = dh_first<0>(0.0f, 0.0f, s[0], c[0]);
    dh_apply<-1, false, false, true>(B, 0.0f, 0.0f, s[1], c[1]);
    dh_apply<-1, false, false, true>(B, 0.0f, 0.0f, s[2], c[2]);
    dh_apply<-1, false, true, true>(B, 0.0f, bl[6], s[3], c[3]);
    p[7] = rot_apply(Rg, B.t);
    dh_apply<-1, false, false, true>(B, 0.0f, 0.0f, s[4], c[4]);
    dh_apply<-1, false, false, true>(B, 0.0f, 0.0f, s[5], c[5]);
    dh_apply<-1, false, true, true>(B, 0.0f, bl[7], s[6], c[6]);
    p[8] = rot_apply(Rg, B.t);
    dh_apply<-1, false, false, true>(B, 0.0f, 0.0f, s[7], c[7]);
    dh_apply<-1, false, false, true>(B, 0.0f, 0.0f, s[8], c[8]);
    return B;
}
template <bool GUARD = true>
DHAUG_HD void fk_right_leg(const float* __restrict__ ang, const float* __restrict__ bl, const Rot& Rg, V3* __restrict__ p) {
    DHAUG_SC(0, 0.0f); DHAUG_SC(1, -90.0f); DHAUG_SC(2, 180.0f); DHAUG_SC(3, 0.0f);
    const float s[4] = {s0, s1, s2, s3}, c[4] = {c0, c1, c2, c3};
    fk_right_leg_sc(s, c, bl, Rg, p);
}
template <bool GUARD = true>
DHAUG_HD void fk_left_leg(const float* __restrict__ ang, const float* __restrict__ bl, const Rot& Rg, V3* __restrict__ p) {
    DHAUG_SC(5, 180.0f); DHAUG_SC(6, -90.0f); DHAUG_SC(7, 0.0f); DHAUG_SC(8, 0.0f);
    const float s[4] = {s5, s6, s7, s8}, c[4] = {c5, c6, c7, c8};
    fk_left_leg_sc(s, c, bl, Rg, p);
}
template <bool GUARD = true>
DHAUG_HD Frame fk_body(const float* __restrict__ ang, const float* __restrict__ bl, const Rot& Rg, V3* __restrict__ p) {
    DHAUG_SC(10, 90.0f); DHAUG_SC(11, -90.0f); DHAUG_SC(12, -90.0f); DHAUG_SC(13, -90.0f); DHAUG_SC(14, -90.0f);
    DHAUG_SC(15, -90.0f); DHAUG_SC(16, -90.0f); DHAUG_SC(17, -90.0f); DHAUG_SC(18, -90.0f);
    const float s[9] = {s10, s11, s12, s13, s14, s15, s16, s17, s18}, c[9] = {c10, c11, c12, c13, c14, c15, c16, c17, c18};
    return fk_body_sc(s, c, bl, Rg, p);
}
// head: body frames 9..12 (angles 19..21)
template <bool GUARD = true>
DHAUG_HD void fk_head(const float* __restrict__ ang, const float* __restrict__ bl, const Rot& Rg, Frame F, V3* __restrict__ p) {
    DHAUG_SC(19, -90.0f); DHAUG_SC(20, -90.0f); DHAUG_SC(21, 0.0f);
    dh_apply<-1, false, false, true>(F, 0.0f, 0.0f, s19, c19);
    dh_apply<-1, false, false, true>(F, 0.0f, 0.0f, s20, c20);
    dh_apply<-1, false, false, true>(F, 0.0f, 0.0f, s21, c21);
    dh_apply<1, true, false, false>(F, bl[14], 0.0f, 0.0f, 1.0f);
    p[9] = rot_apply(Rg, F.t);
}
// the sines / cosines of an arm do not depend on the body frame: ArmSC lets a wave compute them before frame 8 arrives
struct ArmSC { float s[4], c[4]; };
template <bool GUARD = true>
DHAUG_HD ArmSC fk_right_arm_sc(const float* __restrict__ ang) {
    DHAUG_SC(23, -180.0f); DHAUG_SC(24, -90.0f); DHAUG_SC(25, 180.0f); DHAUG_SC(26, 0.0f);
    ArmSC a; a.s[0] = s23; a.c[0] = c23; a.s[1] = s24; a.c[1] = c24; a.s[2] = s25; a.c[2] = c25; a.s[3] = s26; a.c[3] = c26;
    return a;
}
template <bool GUARD = true>
DHAUG_HD ArmSC fk_left_arm_sc(const float* __restrict__ ang) {
    DHAUG_SC(28, 0.0f); DHAUG_SC(29, -90.0f); DHAUG_SC(30, 0.0f); DHAUG_SC(31, 0.0f);
    ArmSC a; a.s[0] = s28; a.c[0] = c28; a.s[1] = s29; a.c[1] = c29; a.s[2] = s30; a.c[2] = c30; a.s[3] = s31; a.c[3] = c31;
    return a;
}
DHAUG_HD void fk_right_arm(const ArmSC& a, const float* __restrict__ bl, const Rot& Rg, Frame F, V3* __restrict__ p) {
    dh_apply<-1, true, false, true>(F, -bl[9], 0.0f, a.s[0], a.c[0]);
    p[13] = rot_apply(Rg, F.t);
    dh_apply<-1, false, false, true>(F, 0.0f, 0.0f, a.s[1], a.c[1]);
    dh_apply<-1, false, false, true>(F, 0.0f, 0.0f, a.s[2], a.c[2]);
    dh_apply<0, true, false, true>(F, bl[11], 0.0f, a.s[3], a.c[3]);
    p[14] = rot_apply(Rg, F.t);
    dh_apply<0, true, false, false>(F, bl[13], 0.0f, 0.0f, 1.0f);
    p[15] = rot_apply(Rg, F.t);
}
DHAUG_HD void fk_left_arm(const ArmSC& a, const float* __restrict__ bl, const Rot& Rg, Frame F, V3* __restrict__ p) {
    dh_apply<-1, true, false, true>(F, bl[8], 0.0f, a.s[0], a.c[0]);
    p[10] = rot_apply(Rg, F.t);
    dh_apply<1, false, false, true>(F, 0.0f, 0.0f, a.s[1], a.c[1]);
    dh_apply<1, false, false, true>(F, 0.0f, 0.0f, a.s[2], a.c[2]);
    dh_apply<0, true, false, true>(F, bl[10], 0.0f, a.s[3], a.c[3]);
    p[11] = rot_apply(Rg, F.t);
    dh_apply<0, true, false, false>(F, bl[12], 0.0f, 0.0f, 1.0f);
    p[12] = rot_apply(Rg, F.t);
}

// Forward kinematics of one pose.  ang[37] degrees, bl[15] metres -> p[16] root-free, globally rotated.
template <bool GUARD = true>
DHAUG_HD void fk_pose(const float* __restrict__ ang, const float* __restrict__ bl, V3* __restrict__ p) {
    float sgx, cgx;
    const Rot Rg = fk_global<GUARD>(ang, sgx, cgx);
    p[0] = mk(0.0f, 0.0f, 0.0f);                        // Hip: body frame 0 has a = d = 0
    fk_right_leg<GUARD>(ang, bl, Rg, p);
    fk_left_leg<GUARD>(ang, bl, Rg, p);
    const Frame B = fk_body<GUARD>(ang, bl, Rg, p);
    fk_head<GUARD>(ang, bl, Rg, B, p);
    fk_right_arm(fk_right_arm_sc<GUARD>(ang), bl, Rg, B, p);
    fk_left_arm(fk_left_arm_sc<GUARD>(ang), bl, Rg, B, p);
}


// ---------------------------------------------------------------------------------------------------
// Reverse mode.  Every chain is recomputed forward to its leaf, then walked back leaf -> base by
// inverting one joint at a time (C_{k-1} = C_k * T_k^{-1}: rigid transforms, O(1) state), carrying the
// suffix sums over downstream output joints  G = sum h_j,  M = sum p_j x h_j  (h = Rg^T * upstream grad):
//     d/d theta_k = (pi/180) * z_k . (M - t_k x G)        z_k = rotation axis = column 2 of C_k
//     d/d a_k     = x_{k-1} . G                           x   = column 0 of C_{k-1} Rx(alpha)
//     d/d d_k     = z_k . G
// Global rotation: Mw = Rg * (sum over chains of M);  d/d(ax,ay,az) = (pi/180) * (x^, Rx y^, Rg z^) . Mw.
// alpha = +-90 deg is taken as exact (eps dropped) on the way back: a 4e-8 perturbation of a gradient.
// ---------------------------------------------------------------------------------------------------
struct GM { V3 G, M; };
DHAUG_HD GM gm_zero() { GM s; s.G = mk(0.f, 0.f, 0.f); s.M = mk(0.f, 0.f, 0.f); return s; }
DHAUG_HD void gm_add(GM& s, const GM& o) { s.G = s.G + o.G; s.M = s.M + o.M; }
DHAUG_HD void add_out(GM& s, const Frame& F, V3 h) { s.G = s.G + h; s.M = s.M + cross(F.t, h); }

// F = C_k on entry, C_{k-1} on exit.
template <int ALPHA, bool HAS_A, bool HAS_D>
DHAUG_HD void bw_step(Frame& F, const GM& s, float a, float d, float st, float ct, float& g_theta, float& g_a, float& g_d) {
    g_theta = kDeg2Rad * dot(F.c2, s.M - cross(F.t, s.G));
    if (HAS_D) g_d = dot(F.c2, s.G);
    V3 x = axpy(-st, F.c1, ct * F.c0);
    V3 y = axpy(ct, F.c1, st * F.c0);
    V3 z = F.c2;
    if (HAS_A) g_a = dot(x, s.G);
    V3 t = F.t;
    if (HAS_D) t = axpy(-d, z, t);
    if (HAS_A) t = axpy(-a, x, t);
    F.t = t;
    F.c0 = x;
    if (ALPHA == 0) { F.c1 = y; F.c2 = z; }
    else            { F.c1 = (-(float)ALPHA) * z; F.c2 = ((float)ALPHA) * y; }
}
// leaf joint: forward skipped its rotation, so F.R is still C_{k-1}.R
DHAUG_HD void bw_leaf(Frame& F, const GM& s, float a, float& g_a) {
    g_a = dot(F.c0, s.G);
    F.t = axpy(-a, F.c0, F.t);
}

// ang[37], bl[15]; gload(j) -> upstream gradient of joint j (world frame).
// Writes gang[37], gbl[15], groot.
template <typename GLoad>
DHAUG_HD void fk_pose_backward(const float* __restrict__ ang, const float* __restrict__ bl, GLoad gload,
                               float* __restrict__ gang, float* __restrict__ gbl, V3& groot) {
    constexpr bool GUARD = true;
    float sgx, cgx, du;
    Rot Rg = global_rot(ang[34], ang[35], ang[36], sgx, cgx);
    groot = mk(0.f, 0.f, 0.f);
    V3 Mtot = mk(0.f, 0.f, 0.f);
    auto H = [&](int j) { V3 g = gload(j); groot = groot + g; return rot_apply_t(Rg, g); };
    gang[4] = gang[9] = gang[22] = gang[27] = gang[32] = gang[33] = 0.0f;
    {   // right leg
        DHAUG_SC(0, 0.0f); DHAUG_SC(1, -90.0f); DHAUG_SC(2, 180.0f); DHAUG_SC(3, 0.0f);
        Frame F = dh_first<0, false>(bl[5], 0.0f, s0, c0);
        dh_apply<-1, false, false, true, false>(F, 0.0f, 0.0f, s1, c1);
        dh_apply<-1, false, false, true, false>(F, 0.0f, 0.0f, s2, c2);
        dh_apply<0, true, false, true, false>(F, bl[3], 0.0f, s3, c3);
        dh_apply<0, true, false, false, false>(F, bl[1], 0.0f, 0.0f, 1.0f);
        GM s = gm_zero();
        add_out(s, F, H(3));
        bw_leaf(F, s, bl[1], gbl[1]);
        add_out(s, F, H(2));
        bw_step<0, true, false>(F, s, bl[3], 0.0f, s3, c3, gang[3], gbl[3], du);
        bw_step<-1, false, false>(F, s, 0.0f, 0.0f, s2, c2, gang[2], du, du);
        bw_step<-1, false, false>(F, s, 0.0f, 0.0f, s1, c1, gang[1], du, du);
        add_out(s, F, H(1));
        bw_step<0, true, false>(F, s, bl[5], 0.0f, s0, c0, gang[0], gbl[5], du);
        Mtot = Mtot + s.M;
    }
    {   // left leg
        DHAUG_SC(5, 180.0f); DHAUG_SC(6, -90.0f); DHAUG_SC(7, 0.0f); DHAUG_SC(8, 0.0f);
        Frame F = dh_first<0, false>(-bl[4], 0.0f, s5, c5);
        dh_apply<1, false, false, true, false>(F, 0.0f, 0.0f, s6, c6);
        dh_apply<1, false, false, true, false>(F, 0.0f, 0.0f, s7, c7);
        dh_apply<0, true, false, true, false>(F, bl[2], 0.0f, s8, c8);
        dh_apply<0, true, false, false, false>(F, bl[0], 0.0f, 0.0f, 1.0f);
        GM s = gm_zero();
        float ga;
        add_out(s, F, H(6));
        bw_leaf(F, s, bl[0], gbl[0]);
        add_out(s, F, H(5));
        bw_step<0, true, false>(F, s, bl[2], 0.0f, s8, c8, gang[8], gbl[2], du);
        bw_step<1, false, false>(F, s, 0.0f, 0.0f, s7, c7, gang[7], du, du);
        bw_step<1, false, false>(F, s, 0.0f, 0.0f, s6, c6, gang[6], du, du);
        add_out(s, F, H(4));
        bw_step<0, true, false>(F, s, -bl[4], 0.0f, s5, c5, gang[5], ga, du);
        gbl[4] = -ga;
        Mtot = Mtot + s.M;
    }
    {   // body + head + arms
        DHAUG_SC(10, 90.0f); DHAUG_SC(11, -90.0f); DHAUG_SC(12, -90.0f); DHAUG_SC(13, -90.0f); DHAUG_SC(14, -90.0f);
        DHAUG_SC(15, -90.0f); DHAUG_SC(16, -90.0f); DHAUG_SC(17, -90.0f); DHAUG_SC(18, -90.0f);
        Frame B = dh_first<0, false>(0.0f, 0.0f, s10, c10);
        dh_apply<-1, false, false, true, false>(B, 0.0f, 0.0f, s11, c11);
        dh_apply<-1, false, false, true, false>(B, 0.0f, 0.0f, s12, c12);
        dh_apply<-1, false, true, true, false>(B, 0.0f, bl[6], s13, c13);
        dh_apply<-1, false, false, true, false>(B, 0.0f, 0.0f, s14, c14);
        dh_apply<-1, false, false, true, false>(B, 0.0f, 0.0f, s15, c15);
        dh_apply<-1, false, true, true, false>(B, 0.0f, bl[7], s16, c16);
        dh_apply<-1, false, false, true, false>(B, 0.0f, 0.0f, s17, c17);
        dh_apply<-1, false, false, true, false>(B, 0.0f, 0.0f, s18, c18);
        GM sb = gm_zero();
        {   // head: body joints 9..12
            DHAUG_SC(19, -90.0f); DHAUG_SC(20, -90.0f); DHAUG_SC(21, 0.0f);
            Frame F = B;
            dh_apply<-1, false, false, true, false>(F, 0.0f, 0.0f, s19, c19);
            dh_apply<-1, false, false, true, false>(F, 0.0f, 0.0f, s20, c20);
            dh_apply<-1, false, false, true, false>(F, 0.0f, 0.0f, s21, c21);
            dh_apply<1, true, false, false, false>(F, bl[14], 0.0f, 0.0f, 1.0f);
            GM s = gm_zero();
            add_out(s, F, H(9));
            bw_leaf(F, s, bl[14], gbl[14]);
            bw_step<-1, false, false>(F, s, 0.0f, 0.0f, s21, c21, gang[21], du, du);
            bw_step<-1, false, false>(F, s, 0.0f, 0.0f, s20, c20, gang[20], du, du);
            bw_step<-1, false, false>(F, s, 0.0f, 0.0f, s19, c19, gang[19], du, du);
            gm_add(sb, s);
        }
        {   // right arm
            DHAUG_SC(23, -180.0f); DHAUG_SC(24, -90.0f); DHAUG_SC(25, 180.0f); DHAUG_SC(26, 0.0f);
            Frame F = B;
            dh_apply<-1, true, false, true, false>(F, -bl[9], 0.0f, s23, c23);
            dh_apply<-1, false, false, true, false>(F, 0.0f, 0.0f, s24, c24);
            dh_apply<-1, false, false, true, false>(F, 0.0f, 0.0f, s25, c25);
            dh_apply<0, true, false, true, false>(F, bl[11], 0.0f, s26, c26);
            dh_apply<0, true, false, false, false>(F, bl[13], 0.0f, 0.0f, 1.0f);
            GM s = gm_zero();
            float ga;
            add_out(s, F, H(15));
            bw_leaf(F, s, bl[13], gbl[13]);
            add_out(s, F, H(14));
            bw_step<0, true, false>(F, s, bl[11], 0.0f, s26, c26, gang[26], gbl[11], du);
            bw_step<-1, false, false>(F, s, 0.0f, 0.0f, s25, c25, gang[25], du, du);
            bw_step<-1, false, false>(F, s, 0.0f, 0.0f, s24, c24, gang[24], du, du);
            add_out(s, F, H(13));
            bw_step<-1, true, false>(F, s, -bl[9], 0.0f, s23, c23, gang[23], ga, du);
            gbl[9] = -ga;
            gm_add(sb, s);
        }
        {   // left arm
            DHAUG_SC(28, 0.0f); DHAUG_SC(29, -90.0f); DHAUG_SC(30, 0.0f); DHAUG_SC(31, 0.0f);
            Frame F = B;
            dh_apply<-1, true, false, true, false>(F, bl[8], 0.0f, s28, c28);
            dh_apply<1, false, false, true, false>(F, 0.0f, 0.0f, s29, c29);
            dh_apply<1, false, false, true, false>(F, 0.0f, 0.0f, s30, c30);
            dh_apply<0, true, false, true, false>(F, bl[10], 0.0f, s31, c31);
            dh_apply<0, true, false, false, false>(F, bl[12], 0.0f, 0.0f, 1.0f);
            GM s = gm_zero();
            add_out(s, F, H(12));
            bw_leaf(F, s, bl[12], gbl[12]);
            add_out(s, F, H(11));
            bw_step<0, true, false>(F, s, bl[10], 0.0f, s31, c31, gang[31], gbl[10], du);
            bw_step<1, false, false>(F, s, 0.0f, 0.0f, s30, c30, gang[30], du, du);
            bw_step<1, false, false>(F, s, 0.0f, 0.0f, s29, c29, gang[29], du, du);
            add_out(s, F, H(10));
            bw_step<-1, true, false>(F, s, bl[8], 0.0f, s28, c28, gang[28], gbl[8], du);
            gm_add(sb, s);
        }
        // body frames 8 -> 0
        Frame F = B;
        bw_step<-1, false, false>(F, sb, 0.0f, 0.0f, s18, c18, gang[18], du, du);
        bw_step<-1, false, false>(F, sb, 0.0f, 0.0f, s17, c17, gang[17], du, du);
        add_out(sb, F, H(8));                                  // Thorax = frame 6
        bw_step<-1, false, true>(F, sb, 0.0f, bl[7], s16, c16, gang[16], du, gbl[7]);
        bw_step<-1, false, false>(F, sb, 0.0f, 0.0f, s15, c15, gang[15], du, du);
        bw_step<-1, false, false>(F, sb, 0.0f, 0.0f, s14, c14, gang[14], du, du);
        add_out(sb, F, H(7));                                  // Spine = frame 3
        bw_step<-1, false, true>(F, sb, 0.0f, bl[6], s13, c13, gang[13], du, gbl[6]);
        bw_step<-1, false, false>(F, sb, 0.0f, 0.0f, s12, c12, gang[12], du, du);
        bw_step<-1, false, false>(F, sb, 0.0f, 0.0f, s11, c11, gang[11], du, du);
        bw_step<0, false, false>(F, sb, 0.0f, 0.0f, s10, c10, gang[10], du, du);
        Mtot = Mtot + sb.M;
    }
    (void)H(0);                                                // Hip moves only with the root
    V3 Mw = rot_apply(Rg, Mtot);
    gang[34] = kDeg2Rad * Mw.x;
    gang[35] = kDeg2Rad * fmaf(sgx, Mw.z, cgx * Mw.y);
    gang[36] = kDeg2Rad * dot(mk(Rg.r0.z, Rg.r1.z, Rg.r2.z), Mw);
}

// ---------------------------------------------------------------------------------------------------
// Generator tail: head (35 pre-activations) -> 37 angles (deg), root, jittered bone lengths.
// R/models_Fk_GAN/Fk_generator.py:121-168 and :216-230.
// ---------------------------------------------------------------------------------------------------
// joint limits per slot (degrees), slots 34..36 = global rotation
static constexpr float kAngLo[37] = {-110, -110, -110, -180, 0, -65, -65, -110, -180, 0,
                                         -180, -180, -180, -180, -180, -180, -180, -180, -180, -180, -180, -180, 0, 0,
                                         -155, -155, -100, 0, 0, -65, -65, -100, 0, 0, -180, -180, -180};
static constexpr float kAngHi[37] = {65, 65, 180, 0, 0, 110, 110, 180, 0, 0,
                                         180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 0, 0,
                                         65, 65, 180, 180, 0, 155, 155, 180, 180, 0, 180, 180, 180};
// head column feeding each angle slot (-1: slot forced to 0)
static constexpr int kSlotCol[37] = {0, 1, 2, 3, -1, 4, 5, 6, 7, -1, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19,
                                         -1, -1, 20, 21, 22, 23, -1, 24, 25, 26, 27, -1, 28, 29, 30};
// jitter column per bone (-1: thorax, never jittered)
static constexpr int kJitterCol[15] = {0, 0, 1, 1, 2, 2, 3, -1, 4, 4, 5, 5, 6, 6, 7};

// th[35] = tanh(head) (already computed).  Writes ang[37].
template <bool PREANGLE>
DHAUG_HD void tail_angles(const float* __restrict__ th, float* __restrict__ ang) {
#pragma unroll
    for (int i = 0; i < 37; ++i) {
        const int col = kSlotCol[i];
        if (col < 0) { ang[i] = 0.0f; continue; }
        float t = th[col];
        if (PREANGLE) {
            // generator_angle * (hi - lo) / 2 + (hi + lo) / 2, evaluated left to right in fp32
            ang[i] = (t * (kAngHi[i] - kAngLo[i])) * 0.5f + (kAngHi[i] + kAngLo[i]) * 0.5f;
        } else {
            ang[i] = t * 180.0f;
        }
    }
}
// d ang[i] / d th[col]
template <bool PREANGLE> DHAUG_HD constexpr float tail_scale(int i) {
    return PREANGLE ? (kAngHi[i] - kAngLo[i]) * 0.5f : 180.0f;
}

}  // namespace dhaug_fk
