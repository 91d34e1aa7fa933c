// The pixel-beam kernel's own tables and per-ray tests.
//
// Every ray of an orthographic camera has the camera's direction d (camera_orthographic.go:44), so the part of
// Face.Collide (shape_mesh.go:95-124) that does not depend on the ray's origin is the same for every ray of a frame:
//   edge1 = p1 - p0, edge2 = p2 - p0, h = d x edge2, a = edge1 . h, the test -1e-8 < a < 1e-8, f = 1 / a.
// k_beam_prepare computes these once per face and direction, with the operations and their order exactly as in mt_core
// (dev_shapes.h) -- the library is compiled without floating-point contraction -- so what a ray computes from them,
//   s = o - p0, u = f (s . h), q = s x edge1, v = f (d . q), t = f (edge2 . q),
// are the reference's numbers bit for bit.  The conservative float32 pre-test is prepared the same way: with
// U = s . (d x e2) and V = s . (e1 x d) the face can be rejected when U < -eu, V < -ev or U + V > |a| + eu + ev + ea (signs
// mirrored for a < 0; margins as in tri32_rejects), which per ray is a subtraction, two dot products and three compares.
//
// Rays from ONE point (PinholeCamera, camera_pinhole.go:20-34; FocusableCamera with no aperture, camera_focusable.go:26-63)
// share the other half: s = o - p0, q = s x edge1 and edge2 . q are the same for every ray, and what a ray computes is
//   h = d x edge2, a = edge1 . h, f = 1 / a, u = f (s . h), v = f (d . q), t = f (edge2 . q).
// The tables have the same shape with other contents (BeamFace64: s, edge1, edge2, q, edge2 . q; BeamFace32: the vectors
// e2 x s, s x e1, e2 x e1, whose products with the direction are U, V and a, and the margins eu, ev, ea).  PYR selects them.
#pragma once

#include "dev_bvh.h"
#include "dev_shade.h"

namespace rt {

RT_DEV BeamFace32 load_face32_scalar(const BeamFace32* t, int32_t index) {
    const RT_CONST_AS i32x4* p = (const RT_CONST_AS i32x4*)(t + index);
    union {
        i32x4 v[3];
        BeamFace32 f;
    } u;
    u.v[0] = p[0]; u.v[1] = p[1]; u.v[2] = p[2];
    return u.f;
}

// One ray (float32 origin in the bundle's frame) against one prepared face: true = the float64 test cannot accept it.
RT_DEV bool face32_rejects(const BeamFace32& t, float ox, float oy, float oz) {
#pragma clang fp contract(fast)
    const float sx = ox - t.p0[0], sy = oy - t.p0[1], sz = oz - t.p0[2];
    const float us = sx * t.h[0] + sy * t.h[1] + sz * t.h[2];
    const float vs = sx * t.g[0] + sy * t.g[1] + sz * t.g[2];
    return us < t.neg_eu || vs < t.neg_ev || us + vs > t.lim;
}

// The same for a whole bundle: origins within (ex, ey, ez) of (ox, oy, oz) per axis.  U and V are linear in the origin,
// so over the footprint they stay within dU = sum e_k |h_k| and dV = sum e_k |g_k| of their values at the centre.
RT_DEV bool face32_rejects_bundle(const BeamFace32& t, float ox, float oy, float oz, float ex, float ey, float ez) {
#pragma clang fp contract(fast)
    const float sx = ox - t.p0[0], sy = oy - t.p0[1], sz = oz - t.p0[2];
    const float us = sx * t.h[0] + sy * t.h[1] + sz * t.h[2];
    const float vs = sx * t.g[0] + sy * t.g[1] + sz * t.g[2];
    const float slack = 1.0f + 1.0f / 65536.0f;  // the bounds themselves are rounded
    const float dU = (ex * fabsf(t.h[0]) + ey * fabsf(t.h[1]) + ez * fabsf(t.h[2])) * slack;
    const float dV = (ex * fabsf(t.g[0]) + ey * fabsf(t.g[1]) + ez * fabsf(t.g[2])) * slack;
    // U + V is linear in the origin too: its own bound is tighter than dU + dV (the rounding of the sums h + g is inside the slack's share of dU + dV)
    const float dW = fminf((ex * fabsf(t.h[0] + t.g[0]) + ey * fabsf(t.h[1] + t.g[1]) + ez * fabsf(t.h[2] + t.g[2])) * slack + (dU + dV) * (1.0f / 65536.0f), dU + dV);
    return us + dU < t.neg_eu || vs + dV < t.neg_ev || (us + vs) - dW > t.lim;
}

// The bundle's footprint on the image plane, centre (cu, cv) and half sizes (hu, hv) along the camera's axes (margins for the
// rounding of both sides included), beside the face's extent along them.
RT_DEV bool face32_beside_footprint(const BeamFace32& t, float cu, float cv, float hu, float hv) {
    return t.umax < cu - hu || t.umin > cu + hu || t.vmax < cv - hv || t.vmin > cv + hv;
}

// Rays from one point: one ray's direction against a face prepared for that point.  The table's fields hold
// A = e2 x s (p0), Q = s x e1 (h), N = e2 x e1 (g) and eu, ev, ea (neg_eu, neg_ev, lim).
RT_DEV bool face32_rejects_dir(const BeamFace32& t, float dx, float dy, float dz) {
#pragma clang fp contract(fast)
    const float U = dx * t.p0[0] + dy * t.p0[1] + dz * t.p0[2];
    const float V = dx * t.h[0] + dy * t.h[1] + dz * t.h[2];
    const float a = dx * t.g[0] + dy * t.g[1] + dz * t.g[2];
    const float eu = t.neg_eu, ev = t.neg_ev, ea = t.lim, aa = fabsf(a);
    if (!(aa > ea)) return false;  // sign of a uncertain (grazing or degenerate): let float64 decide
    const float us = a < 0 ? -U : U, vs = a < 0 ? -V : V;
    return us < -eu || vs < -ev || us + vs > aa + eu + ev + ea;
}

// The same for every direction in the box (cx, cy, cz) +- (ex, ey, ez): U, V and a are linear in the direction.
RT_DEV bool face32_rejects_dirs(const BeamFace32& t, float cx, float cy, float cz, float ex, float ey, float ez) {
#pragma clang fp contract(fast)
    const float slack = 1.0f + 1.0f / 65536.0f;  // the bounds themselves are rounded
    const float U = cx * t.p0[0] + cy * t.p0[1] + cz * t.p0[2], dU = (ex * fabsf(t.p0[0]) + ey * fabsf(t.p0[1]) + ez * fabsf(t.p0[2])) * slack;
    const float V = cx * t.h[0] + cy * t.h[1] + cz * t.h[2], dV = (ex * fabsf(t.h[0]) + ey * fabsf(t.h[1]) + ez * fabsf(t.h[2])) * slack;
    const float a = cx * t.g[0] + cy * t.g[1] + cz * t.g[2], da = (ex * fabsf(t.g[0]) + ey * fabsf(t.g[1]) + ez * fabsf(t.g[2])) * slack;
    const float eu = t.neg_eu, ev = t.neg_ev, ea = t.lim, aa = fabsf(a);
    if (!(aa - da > ea)) return false;  // the sign of a is not the same, for certain, over the whole bundle
    const float us = a < 0 ? -U : U, vs = a < 0 ? -V : V;
    return us + dU < -eu || vs + dV < -ev || (us - dU) + (vs - dV) > (aa + da) + eu + ev + ea;
}

// What a pixel's rounds test, when it is short enough: the faces that survive beam_cull_item32, in list order, each with its
// prepared float32 form in LDS -- kBeamFaceDwords dwords: BeamFace32 (12), the entry distance of its leaf (suffix minimum), its
// position in leaf order.  A round then reads no memory until a face passes the pre-test.  A pixel with more survivors
// keeps walking its leaf list (beam_test_prepared).
#ifndef RT_BEAM_FACES
#define RT_BEAM_FACES 8  // against the leaves a list holds, which share the LDS with it (launch_beam), on the 4K scan (7.8 survivors per listing pixel on average): 6 faces 67.2 ms, 8: 66.5, 10: 67.7; at a later state of the kernel 56.7 / 55.9 / 57.0
#endif
constexpr int kBeamFacesMax = RT_BEAM_FACES;
constexpr int kBeamFaceDwords = 16;
constexpr int kBeamCullDwords = (2 * kBeamPixels + 1 + 3) & ~3;  // beam_cull_item32's own LDS: prefix sums of the list lengths (kBeamPixels + 1), survivors per pixel (kBeamPixels); a multiple of 4 keeps the compact entries 16-byte aligned

// beam_cull_item (dev_bvh.h) on the prepared faces: kBeamLeafFaces lanes per listed leaf, one per face, each testing its
// face against the bundle of the leaf's pixel; the survivors become the leaf's mask and are appended, in list order, to the
// pixel's compact array (entry 12 holds the position of the face's leaf in the list until beam_patch_distances has run).
// beam_state[j * kBeamStateDwords + 25] receives the number of survivors of pixel j.  Must be entered by all 64 lanes.
template <bool PYR>
RT_DEV void beam_cull_item32(const BeamFace32* faces, int32_t* beam_state, int32_t* beam_cand, int cands, int32_t* prefix, int32_t* compact, int n_px, int lane) {
    int len = 0;
    if (lane < n_px) {
        const int32_t v = beam_state[lane * kBeamStateDwords + 15];
        len = (v & 0x100) ? 0 : (v & 0xff);  // a list that goes on is walked in instalments inside the rounds, unthinned
    }
    int incl = len;
#pragma unroll
    for (int off = 1; off < kBeamPixels; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    int32_t* const count = prefix + kBeamPixels + 1;
    if (lane < kBeamPixels) {
        prefix[lane + 1] = incl;
        count[lane] = 0;
    }
    if (lane == 0) prefix[0] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int total = __builtin_amdgcn_readfirstlane(prefix[kBeamPixels]);
    for (int base = 0; base < total; base += 64 / kBeamLeafFaces) {
        const int s = base + lane / kBeamLeafFaces, q = lane & (kBeamLeafFaces - 1);
        bool keep = false;
        int32_t* entry = beam_cand;
        int j = 0, at = 0;
        int32_t face = 0;
        BeamFace32 t{};
        if (s < total) {
#pragma unroll
            for (int k = 1; k < kBeamPixels; k++)
                if (s >= prefix[k]) j = k;
            at = s - prefix[j];
            entry = beam_cand + (j * cands + at) * 3;
            const uint32_t code = (uint32_t)~entry[0];
            if (q <= (int)(code & (uint32_t)(kBeamLeafFaces - 1))) {
                const int32_t* st = beam_state + j * kBeamStateDwords;
                face = (int32_t)(code >> RT_BEAM_LEAF_BITS) + q;
                t = faces[face];
                if (PYR) keep = !face32_rejects_dirs(t, __int_as_float(st[19]), __int_as_float(st[20]), __int_as_float(st[21]), __int_as_float(st[22]),
                                                     __int_as_float(st[23]), __int_as_float(st[24]));
                else keep = !face32_beside_footprint(t, __int_as_float(st[26]), __int_as_float(st[27]), __int_as_float(st[28]), __int_as_float(st[29])) &&
                            !face32_rejects_bundle(t, __int_as_float(st[16]), __int_as_float(st[17]), __int_as_float(st[18]), __int_as_float(st[19]),
                                                   __int_as_float(st[20]), __int_as_float(st[21]));
            }
        }
        const unsigned long long kept = __builtin_amdgcn_ballot_w64(keep);
        if (s < total && q == 0) entry[2] = (int32_t)((kept >> (lane & (64 - kBeamLeafFaces))) & (unsigned long long)kBeamLeafMask);
        if (keep) {  // survivors of one pixel sit in consecutive lanes: the slot is the pixel's count so far + the survivors below this lane
            const int first = prefix[j] - base;
            const int lo_lane = first > 0 ? first * kBeamLeafFaces : 0;
            const unsigned long long mine = kept & ((1ull << lane) - 1ull) & ~((1ull << lo_lane) - 1ull);
            const int slot = count[j] + (int)__popcll(mine);
            if (slot < kBeamFacesMax) {
                int32_t* e = compact + (j * kBeamFacesMax + slot) * kBeamFaceDwords;
                e[0] = __float_as_int(t.p0[0]); e[1] = __float_as_int(t.p0[1]); e[2] = __float_as_int(t.p0[2]);
                e[3] = __float_as_int(t.h[0]); e[4] = __float_as_int(t.h[1]); e[5] = __float_as_int(t.h[2]);
                e[6] = __float_as_int(t.g[0]); e[7] = __float_as_int(t.g[1]); e[8] = __float_as_int(t.g[2]);
                e[9] = __float_as_int(t.neg_eu); e[10] = __float_as_int(t.neg_ev); e[11] = __float_as_int(t.lim);
                e[12] = at;
                e[13] = face;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < n_px) {  // this iteration's survivors of pixel `lane`
            int lo = (prefix[lane] - base) * kBeamLeafFaces, hi = (prefix[lane + 1] - base) * kBeamLeafFaces;
            lo = lo < 0 ? 0 : lo;
            hi = hi > 64 ? 64 : hi;
            if (lo < hi) {
                const unsigned long long upto = hi >= 64 ? ~0ull : (1ull << hi) - 1ull;
                count[lane] += (int)__popcll(kept & upto & ~((1ull << lo) - 1ull));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < n_px) beam_state[lane * kBeamStateDwords + 25] = count[lane];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// After beam_suffix_min: every compact entry takes the entry distance its leaf now carries (a suffix minimum, so the
// distances of a pixel's entries never decrease and a round can stop at the first one beyond its hit).
RT_DEV void beam_patch_distances(const int32_t* beam_state, const int32_t* beam_cand, int cands, int32_t* compact, int n_px, int lane) {
    for (int k = lane; k < n_px * kBeamFacesMax; k += 64) {
        const int j = k / kBeamFacesMax, i = k - j * kBeamFacesMax;
        if (i < beam_state[j * kBeamStateDwords + 25]) {
            int32_t* e = compact + k * kBeamFaceDwords;
            e[12] = beam_cand[(j * cands + e[12]) * 3 + 1];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One prepared face (float64) against this lane's ray: shape_mesh.go:99-124 with what the bundle's rays share taken from the
// table; keeps the closest hit in bh.  Entered by all lanes of the wave; `maybe` = this lane's ray passed the pre-test.
// o, d: the ray (PYR: o is not read, the table holds s = o - p0; else d is the same for every lane).
template <bool PYR>
RT_DEV void beam_face64(const BeamTables& bt, int32_t face, V3 o, V3 d, double T, double tmin, bool maybe, BeamHit& bh) {
    const double e = 1e-8;
    double t = 0, u = 0, v = 0;
    bool ok = false;
    // The whole prepared face in one batch of scalar loads, all seven in flight before anything waits for one of them: left to
    // itself the compiler moves each load below the branch that first needs it, which makes up to four dependent round trips
    // of one (4 % of the 4K scan frame, 7 % of an eighth of it).
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const RT_CONST_AS f64x2* pv = (const RT_CONST_AS f64x2*)(bt.f64 + face);
    const f64x2 a0 = pv[0], a1 = pv[1], a2 = pv[2], a3 = pv[3], a4 = pv[4], a5 = pv[5], a6 = pv[6];
    asm volatile("" ::"s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6));
    const double pp[13] = {a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1], a4[0], a4[1], a5[0], a5[1], a6[0]};
    if (PYR) {
        if (maybe) {
            const V3 s = mk(pp[0], pp[1], pp[2]), edge1 = mk(pp[3], pp[4], pp[5]), edge2 = mk(pp[6], pp[7], pp[8]), q = mk(pp[9], pp[10], pp[11]);
            const V3 h = cross(d, edge2);
            const double a = dot(edge1, h);
            if (!(a > -e && a < e)) {
                const double f = 1 / a;
                u = f * dot(s, h);
                if (!(u < -e || u > 1.0 + e)) {
                    v = f * dot(d, q);
                    if (!(v < -e || u + v > 1.0 + e)) {
                        t = f * pp[12];
                        ok = !(t < tmin || t > bh.t);
                    }
                }
            }
        }
    } else {
        const double f = pp[12];
        // a face the reference rejects whatever the ray (shape_mesh.go:107-109): f is NaN, all bits of its exponent set
        if (((uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(__double_as_longlong(f) >> 32)) & 0x7ff00000u) == 0x7ff00000u) return;
        if (maybe) {
            const V3 s = o - mk(pp[0], pp[1], pp[2]);
            u = f * dot(s, mk(pp[9], pp[10], pp[11]));
            if (!(u < -e || u > 1.0 + e)) {
                const V3 q = cross(s, mk(pp[3], pp[4], pp[5]));
                v = f * dot(d, q);
                if (!(v < -e || u + v > 1.0 + e)) {
                    t = f * dot(mk(pp[6], pp[7], pp[8]), q);
                    ok = !(t < tmin || t > bh.t);
                }
            }
        }
    }
    // equal distances: the face latest in leaf order wins, as in mesh_collide_packet
    // (before the first hit bh.prim is -1 and no face is below it; a flag of its own beside it cost 3 % of the 4K scan frame)
    if (ok && !(t == bh.t && face < bh.prim)) {
        bh.tmax32 = up32(t - T);
        bh.prim = face;
        bh.t = t; bh.u = u; bh.v = v;
    }
}

// The n compact entries of a pixel against this lane's ray.  Must be entered by the lanes of a wave together.
// sound_lanes: the lanes whose `sound` is set, as the caller has them.
// (fx, fy, fz): the ray's origin in the bundle's float32 frame, or (PYR) its direction.
template <bool PYR>
RT_DEV void beam_test_compact(const BeamTables& bt, V3 o, V3 d, float fx, float fy, float fz, double T, const int32_t* compact, int n, double tmin,
                              bool sound, unsigned long long sound_lanes, BeamHit& bh) {
    // (Wave-wide "does any lane ..." questions are asked of the comparisons themselves, whose results are lane masks in scalar
    // registers already, and answered by scalar instructions; asked of a boolean combined from several, each costs two vector
    // instructions: 2 % of a frame with the outline culling on.)
    for (int i = 0; i < n; i++) {
        // the whole entry in one batch of LDS reads (every lane the same address): one round trip per entry instead of three
        // (distance, face, position), which measured 2 % on the 4K scan together with the batch of scalar loads in beam_face64
        typedef int32_t i32x4v __attribute__((ext_vector_type(4)));
        const i32x4v* ev = reinterpret_cast<const i32x4v*>(compact + i * kBeamFaceDwords);
        const i32x4v e0 = ev[0], e1 = ev[1], e2 = ev[2], e3 = ev[3];
        const int32_t e[14] = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3], e2[0], e2[1], e2[2], e2[3], e3[0], e3[1]};
        const float tb = __int_as_float(e[12]);
        BeamFace32 t;
        t.p0[0] = __int_as_float(e[0]); t.p0[1] = __int_as_float(e[1]); t.p0[2] = __int_as_float(e[2]);
        t.h[0] = __int_as_float(e[3]); t.h[1] = __int_as_float(e[4]); t.h[2] = __int_as_float(e[5]);
        t.g[0] = __int_as_float(e[6]); t.g[1] = __int_as_float(e[7]); t.g[2] = __int_as_float(e[8]);
        t.neg_eu = __int_as_float(e[9]); t.neg_ev = __int_as_float(e[10]); t.lim = __int_as_float(e[11]);
        const bool near_enough = tb <= bh.tmax32;
        const unsigned long long live_lanes = __builtin_amdgcn_ballot_w64(near_enough) & sound_lanes;
        if (live_lanes == 0) break;  // the distances never decrease: nothing after this entry begins nearer
        bool maybe;
        if (PYR) {
            maybe = sound && near_enough && !face32_rejects_dir(t, fx, fy, fz);
            RT_STAT(3, 1);
            if (__builtin_amdgcn_ballot_w64(maybe) == 0) continue;
        } else {
            float us, vs;
            {
#pragma clang fp contract(fast)
                const float sx = fx - t.p0[0], sy = fy - t.p0[1], sz = fz - t.p0[2];  // face32_rejects, its three comparisons kept apart
                us = sx * t.h[0] + sy * t.h[1] + sz * t.h[2];
                vs = sx * t.g[0] + sy * t.g[1] + sz * t.g[2];
            }
            const bool in_u = !(us < t.neg_eu), in_v = !(vs < t.neg_ev), in_w = !(us + vs > t.lim);
            RT_STAT(3, 1);
            if ((live_lanes & __builtin_amdgcn_ballot_w64(in_u) & __builtin_amdgcn_ballot_w64(in_v) & __builtin_amdgcn_ballot_w64(in_w)) == 0) continue;
            maybe = sound && near_enough && in_u && in_v && in_w;
        }
        RT_STAT(4, 1);
        beam_face64<PYR>(bt, __builtin_amdgcn_readfirstlane(e[13]), o, d, T, tmin, maybe, bh);
    }
}

// The faces of n listed leaves (those their masks name) against this lane's ray: beam_test (dev_bvh.h) on the prepared
// tables.  o: the ray's float64 origin; (fx, fy, fz): the same in the bundle's float32 frame; d: the camera's direction
// (the same for every lane).  Must be entered by the lanes of a wave together.
template <bool PYR>
RT_DEV void beam_test_prepared(const BeamTables& bt, V3 o, V3 d, float fx, float fy, float fz, double T, const int32_t* cand, int n, bool minima,
                               double tmin, bool sound, BeamHit& bh) {
    for (int c = 0; c < n; c++) {
        uint32_t mask = (uint32_t)__builtin_amdgcn_readfirstlane(cand[3 * c + 2]);
        if (mask == 0) continue;  // no face of this leaf can be hit by any ray of the bundle
        const float tb = __int_as_float(__builtin_amdgcn_readfirstlane(cand[3 * c + 1]));
        if (__builtin_amdgcn_ballot_w64(sound && tb <= bh.tmax32) == 0) {  // the leaf begins beyond every lane's hit
            if (minima) break;
            continue;
        }
        const int32_t first = (int32_t)((uint32_t)~__builtin_amdgcn_readfirstlane(cand[3 * c]) >> RT_BEAM_LEAF_BITS);
        RT_STAT(2, 1);
        for (; mask != 0; mask &= mask - 1) {
            const int32_t face = first + (__builtin_ffs((int)mask) - 1);
            const BeamFace32 t32 = load_face32_scalar(bt.f32, face);
            const bool maybe = sound && tb <= bh.tmax32 && !(PYR ? face32_rejects_dir(t32, fx, fy, fz) : face32_rejects(t32, fx, fy, fz));
            RT_STAT(3, 1);
            if (__builtin_amdgcn_ballot_w64(maybe) == 0) continue;
            RT_STAT(4, 1);
            beam_face64<PYR>(bt, face, o, d, T, tmin, maybe, bh);
        }
    }
}

// TextureImage.At (texture_image.go:87-188) with the image's description in scalars and the four texels of a bilinear
// sample fetched together: image_at (dev_shade.h) for the kernel whose every hit samples the same image.
RT_DEV V3 image_at_uniform(const uint8_t* px, int w, int h, int format, int interp, int wrap, double u, double v) {
    if (wrap == RT_WRAP_CLAMP) {
        u = go_min(go_max(u, 0.0), 1.0);
        v = go_min(go_max(v, 0.0), 1.0);
    } else {
        u = u - floor(u);
        v = v - floor(v);
    }
    v = 1.0 - v;
    const double x = u * (double)(w - 1);
    const double y = v * (double)(h - 1);
    uint32_t c[3];
    if (interp == RT_INTERP_BILINEAR) {
        const int x0 = (int)floor(x), y0 = (int)floor(y);
        const double fx = x - (double)x0, fy = y - (double)y0;
        const int xa = clampi(x0, 0, w - 1), xb = clampi(x0 + 1, 0, w - 1), ya = clampi(y0, 0, h - 1), yb = clampi(y0 + 1, 0, h - 1);
        uint32_t t[4][3];
        if (format == RT_IMG_RGBA16) {
            const RT_GLOBAL_AS unsigned long long* q = (const RT_GLOBAL_AS unsigned long long*)px;  // global, not flat, loads: they do not wait on LDS traffic
            const unsigned long long all[4] = {q[(size_t)ya * w + xa], q[(size_t)ya * w + xb], q[(size_t)yb * w + xa], q[(size_t)yb * w + xb]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t lo = (uint32_t)all[k], hi = (uint32_t)(all[k] >> 32);
                t[k][0] = lo & 0xffffu; t[k][1] = lo >> 16; t[k][2] = hi & 0xffffu;
            }
        } else {  // color.NRGBA.RGBA(): v = v * 0x101 * a / 0xff
            const RT_GLOBAL_AS uint32_t* q = (const RT_GLOBAL_AS uint32_t*)px;
            const uint32_t all[4] = {q[(size_t)ya * w + xa], q[(size_t)ya * w + xb], q[(size_t)yb * w + xa], q[(size_t)yb * w + xb]};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t a = all[k] >> 24;
                uint32_t r = all[k] & 0xffu, g = (all[k] >> 8) & 0xffu, b = (all[k] >> 16) & 0xffu;
                r |= r << 8; g |= g << 8; b |= b << 8;
                if (a != 0xff) {  // opaque texels (nearly all of them) keep v * 0x101
                    r *= a; r /= 0xff;
                    g *= a; g /= 0xff;
                    b *= a; b /= 0xff;
                }
                t[k][0] = r; t[k][1] = g; t[k][2] = b;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double c0 = (double)t[0][k] * (1 - fx) + (double)t[1][k] * fx;
            const double c1 = (double)t[2][k] * (1 - fx) + (double)t[3][k] * fx;
            c[k] = (uint32_t)(uint16_t)(c0 * (1 - fy) + c1 * fy);  // uint16(r) of color.NRGBA64
        }
    } else {
        DImage im;
        im.px = px; im.w = w; im.h = h; im.format = format; im.pad = 0;
        texel16(im, (int)round(x), (int)round(y), c);  // math.Round: half away from zero
    }
    const double r65535 = 1.0 / 65535.0;  // c / 65535.0 (texture_image.go:183-187), see div_by: exact for every 16-bit c
    return mk(div_by((double)c[0], 65535.0, r65535), div_by((double)c[1], 65535.0, r65535), div_by((double)c[2], 65535.0, r65535));
}

// Two float64 in [0, 1) from the four words of a Philox block, as Rng::float64 makes them: (bits >> 11) * 2^-53 of the
// low and the high 64 bits.  hi * 2^-32 and (lo >> 11) * 2^-53 have no bit in common, so their sum is exact.
RT_DEV double draw53(uint32_t lo, uint32_t hi) { return __builtin_fma((double)hi, 0x1.0p-32, (double)(lo >> 11) * 0x1.0p-53); }

}  // namespace rt
