// Textures, the Beckmann microfacet BRDF and the material scatter rules, float64, following
// phys/texture_*.go, microfacet.go and material_*.go.
#pragma once

#include "dev_shapes.h"

namespace rt {

// image.At(x, y).RGBA(): alpha-premultiplied 16-bit channels.
RT_DEV void texel16(const DImage& im, int x, int y, uint32_t c[3]) {
    if (im.format == RT_IMG_RGBA16) {
        const RT_GLOBAL_AS uint16_t* p = (const RT_GLOBAL_AS uint16_t*)im.px + ((size_t)y * im.w + x) * 4;  // global, not flat, loads
        c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
    } else {  // color.NRGBA.RGBA(): v = v * 0x101 * a / 0xff
        const uint32_t q = ((const RT_GLOBAL_AS uint32_t*)im.px)[(size_t)y * im.w + x];
        const uint32_t a = q >> 24;
        uint32_t r = q & 0xffu, g = (q >> 8) & 0xffu, b = (q >> 16) & 0xffu;
        r |= r << 8; g |= g << 8; b |= b << 8;
        if (a != 0xff) {  // opaque texels (nearly all of them) keep v * 0x101: v * 0x101 * 255 / 255
            r *= a; r /= 0xff;
            g *= a; g /= 0xff;
            b *= a; b /= 0xff;
        }
        c[0] = r; c[1] = g; c[2] = b;
    }
}

RT_DEV int clampi(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }

RT_DEV V3 image_at(const DScene& sc, const rt_texture& tx, double u, double v) {  // texture_image.go:87-188
    const DImage im = sc.images[tx.image];
    if (tx.wrap == RT_WRAP_CLAMP) {
        u = go_min(go_max(u, 0.0), 1.0);
        v = go_min(go_max(v, 0.0), 1.0);
    } else {
        u = u - floor(u);
        v = v - floor(v);
    }
    v = 1.0 - v;
    double x = u * (double)(im.w - 1);
    double y = v * (double)(im.h - 1);
    uint32_t c[3];
    if (tx.interp == RT_INTERP_BILINEAR) {
        int x0 = (int)floor(x), x1 = x0 + 1;
        int y0 = (int)floor(y), y1 = y0 + 1;
        double fx = x - (double)x0, fy = y - (double)y0;
        uint32_t c00[3], c10[3], c01[3], c11[3];
        texel16(im, clampi(x0, 0, im.w - 1), clampi(y0, 0, im.h - 1), c00);
        texel16(im, clampi(x1, 0, im.w - 1), clampi(y0, 0, im.h - 1), c10);
        texel16(im, clampi(x0, 0, im.w - 1), clampi(y1, 0, im.h - 1), c01);
        texel16(im, clampi(x1, 0, im.w - 1), clampi(y1, 0, im.h - 1), c11);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double c0 = (double)c00[k] * (1 - fx) + (double)c10[k] * fx;
            double c1 = (double)c01[k] * (1 - fx) + (double)c11[k] * fx;
            c[k] = (uint32_t)(uint16_t)(c0 * (1 - fy) + c1 * fy);  // uint16(r) of color.NRGBA64
        }
    } else {
        int ix = (int)round(x);  // math.Round: half away from zero
        int iy = (int)round(y);
        texel16(im, ix, iy, c);
    }
    // c / 65535.0 (texture_image.go:183-187), see div_by: exact for every 16-bit c
    const double r65535 = 1.0 / 65535.0;
    return mk(div_by((double)c[0], 65535.0, r65535), div_by((double)c[1], 65535.0, r65535), div_by((double)c[2], 65535.0, r65535));
}

RT_DEV V3 texture_at(const DScene& sc, int tex, double u, double v) {
    for (int guard = 0; guard < 16; guard++) {
        const rt_texture& tx = sc.textures[tex];
        if (tx.kind == RT_TEX_UNIFORM) return ld3(tx.color);  // texture_uniform.go:18-20
        if (tx.kind == RT_TEX_IMAGE) return image_at(sc, tx, u, v);
        double sines = sin(tx.frequency * u) * sin(tx.frequency * v);  // texture_checkerboard.go:63-69
        tex = sines < 0 ? tx.odd : tx.even;
    }
    return mk(1, 0, 1);
}

// ---- MicrofacetBRDF (microfacet.go:18-103) --------------------------------------------------

RT_DEV double mf_D(double roughness, V3 h, V3 n) {
    double rough = go_max(roughness, kEps);
    double ct = dot(n, h);
    if (ct <= 0) return 0;
    double m2 = rough * rough;
    double ct2 = ct * ct;
    double exponent = (ct2 - 1) / (m2 * ct2);
    return exp(exponent) / (kPi * m2 * ct2 * ct2);
}
RT_DEV double mf_G1(double roughness, V3 v, V3 n, V3 h) {
    double cv = go_max(0, dot(n, v));
    double ch = go_max(0, dot(h, v));
    if (cv <= 0 || ch <= 0) return 0;
    double tan_v = sqrt(1 - cv * cv) / cv;
    double a = 1 / (roughness * tan_v);
    if (a >= 1.6) return 1;
    return (3.535 * a + 2.181 * a * a) / (1 + 2.276 * a + 2.577 * a * a);
}
RT_DEV double mf_G(double roughness, V3 wo, V3 wi, V3 n, V3 h) {  // microfacet.go:66-68
    return mf_G1(roughness, wo, n, h) * mf_G1(roughness, wi, n, h);
}
RT_DEV V3 mf_F(V3 f0, V3 wo, V3 h) {  // microfacet.go:73-78
    double ct = go_max(0, dot(h, wo));
    double p5 = pow(1 - ct, 5.0);
    return f0 + (mk(1, 1, 1) - f0) * p5;
}
RT_DEV V3 mf_evaluate(double roughness, V3 f0, V3 wo, V3 wi, V3 n) {  // microfacet.go:83-103
    wo = unit(wo); wi = unit(wi); n = unit(n);
    V3 h = unit(wo + wi);
    double D = mf_D(roughness, h, n);
    double G = mf_G(roughness, wo, wi, n, h);
    V3 F = mf_F(f0, wo, h);
    double denom = 4 * go_max(0, dot(n, wo)) * go_max(0, dot(n, wi)) + kEps;
    return F * (D * G / denom);
}

RT_DEV double reflectance(double cos_theta, double ni, double nt) {  // material_dielectric.go:175-179
    double r0 = (ni - nt) / (ni + nt);
    r0 = r0 * r0;
    return r0 + (1 - r0) * pow(1 - cos_theta, 5.0);
}
RT_DEV V3 reflect_ray(V3 v, V3 n) { return unit(v - n * (2 * dot(v, n))); }  // material_dielectric.go:182-184
RT_DEV bool refract(V3 v, V3 n, double ni_over_nt, V3& out) {  // material_dielectric.go:187-196
    V3 uv = unit(v);
    double dt = dot(uv, n);
    double disc = 1 - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (disc > 0) {
        out = (uv - n * dt) * ni_over_nt - n * sqrt(disc);
        return true;
    }
    return false;
}

// What ComputeDirectLighting needs of a hit once Material.Resolve is done (material_lambertian.go:18-33,
// material_metal.go:56-67, material_dielectric.go:105-127): the shading normal (flipped when a dielectric is hit
// from inside), the unit direction back along the ray, and the Lambertian albedo / Metal colour.
struct LitSurface {
    V3 n, wo, albedo;
    int inside;
};
RT_DEV LitSurface lit_surface(const rt_material& m, V3 normal, V3 d, V3 lambert_albedo) {
    LitSurface s;
    s.n = unit(normal);
    s.wo = unit(d * -1.0);  // s.outgoing.Unit() (material_metal.go:59)
    s.albedo = mk(0, 0, 0);
    s.inside = 0;
    if (m.kind == RT_MAT_LAMBERTIAN) s.albedo = lambert_albedo;  // Material.Resolve has just evaluated the texture here
    else if (m.kind == RT_MAT_METAL) s.albedo = ld3(m.albedo);
    else if (!(dot(s.wo, s.n) > 0)) { s.inside = 1; s.n = s.n * -1.0; }  // material_dielectric.go:112-120
    return s;
}

// The shadow ray towards a point light (light.go:29-35 and the loop heads of ComputeDirectLighting).
RT_DEV Ray shadow_ray(const rt_material& m, const rt_light& L, V3 p, V3 n, V3& dir, double& dist) {
    dir = ld3(L.position) - p;
    dist = length(dir);
    dir = dir / dist;
    const V3 wi = m.kind == RT_MAT_LAMBERTIAN ? dir : unit(dir);
    return Ray{p + n * kEps, wi};
}

// One unoccluded light's contribution (material_lambertian.go:47-52, material_metal.go:95-106, material_dielectric.go:159-170).
RT_DEV V3 light_contribution(const rt_material& m, const rt_light& L, V3 n, V3 wo, V3 albedo, int inside, V3 dir, V3 wi) {
    if (m.kind == RT_MAT_LAMBERTIAN) {
        const double ndl = go_max(0, dot(n, dir));
        return (albedo * ld3(L.intensity)) * ndl;
    }
    if (m.kind == RT_MAT_METAL) {
        const V3 val = mf_evaluate(m.fuzz, ld3(m.albedo), wo, wi, n);
        const double ct = go_max(0, dot(n, wi));
        return (ld3(L.intensity) * val) * ct;
    }
    double eta_i = m.ior_exterior, eta_t = m.ior_interior;
    if (inside) { const double tmp = eta_i; eta_i = eta_t; eta_t = tmp; }
    const double cos_i = go_max(0, dot(wi, n));
    const double fresnel = reflectance(cos_i, eta_i, eta_t);
    const V3 val = mf_evaluate(m.roughness, mk(1, 1, 1), wo, wi, n) * fresnel;
    const double ct = go_max(0, dot(n, wi));
    return (ld3(L.intensity) * val) * ct;
}

struct Scatter {
    V3 emission;
    V3 albedo;  // Lambertian: the texture value at the hit (the light loop needs it again)
    V3 dir[2];
    V3 rad[2];
    int n;
};

// Material.Resolve for the six material kinds.  `in_rad` is ray.radiance, `d` the incoming
// direction, c the collision.
RT_DEV void resolve(const DScene& sc, const rt_material& m, V3 d, V3 in_rad, const Col& c, Rng& rng, Scatter& out) {
    out.emission = mk(0, 0, 0);
    out.n = 0;
    switch (m.kind) {
        case RT_MAT_LAMBERTIAN: {  // material_lambertian.go:59-81
            V3 n = unit(c.normal);
            out.dir[0] = cosine_weighted_hemisphere(rng, n);
            out.albedo = texture_at(sc, m.texture, c.u, c.v);
            out.rad[0] = in_rad * out.albedo;
            out.n = 1;
            break;
        }
        case RT_MAT_METAL: {  // material_metal.go:28-48
            V3 reflected = reflect_ray(unit(d), c.normal);
            V3 dir = reflected + in_unit_sphere(rng) * m.fuzz;
            if (dot(dir, c.normal) > 0) {
                out.dir[0] = unit(dir);
                out.rad[0] = in_rad * ld3(m.albedo);
                out.n = 1;
            }
            break;
        }
        case RT_MAT_DIELECTRIC: {  // material_dielectric.go:29-97
            V3 outward;
            double ni_over_nt, cosine, n1, n2;
            if (dot(d, c.normal) > 0) {
                outward = c.normal * -1.0;
                ni_over_nt = m.ior_interior / m.ior_exterior;
                n1 = m.ior_interior; n2 = m.ior_exterior;
                cosine = dot(d, c.normal) / length(d);
                cosine = sqrt(1 - ni_over_nt * ni_over_nt * (1 - cosine * cosine));
            } else {
                outward = c.normal;
                ni_over_nt = m.ior_exterior / m.ior_interior;
                n1 = m.ior_exterior; n2 = m.ior_interior;
                cosine = -dot(d, c.normal) / length(d);
            }
            V3 refracted;
            bool ok = refract(d, outward, ni_over_nt, refracted);
            double reflect_prob = 1.0;
            if (ok) {
                reflect_prob = reflectance(cosine, n1, n2);
                if (m.roughness > 0) refracted = unit(refracted + in_unit_sphere(rng) * m.roughness);
            }
            V3 reflected = reflect_ray(d, c.normal);
            if (m.roughness > 0) reflected = unit(reflected + in_unit_sphere(rng) * m.roughness);
            // the returned slice holds the refracted ray first when there is one (fixed slots: no indexed stores)
            if (ok) {
                out.dir[0] = refracted;
                out.rad[0] = in_rad * (1 - reflect_prob);
                out.dir[1] = reflected;
                out.rad[1] = in_rad * reflect_prob;
                out.n = 2;
            } else {
                out.dir[0] = reflected;
                out.rad[0] = in_rad * reflect_prob;
                out.n = 1;
            }
            break;
        }
        case RT_MAT_EMITTER: {  // material_emitter.go:22-25
            out.emission = texture_at(sc, m.texture, c.u, c.v) * in_rad;
            break;
        }
        case RT_MAT_DEBUG_UV: {  // material_debug_uv.go:18-28
            if (c.u < 0.0 || c.u > 1.0) { out.emission = mk(1.0, 0.0, 0.0); break; }
            out.emission = mk(go_min(go_max(c.u, 0.0), 1.0), go_min(go_max(c.v, 0.0), 1.0), 0.5);
            break;
        }
        default: {  // material_debug_normal.go:45-81
            V3 n = c.normal;
            double len = sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
            if (!(len > 0)) len = 1;
            double r = 0.5 * (n.x / len + 1.0), g = 0.5 * (n.y / len + 1.0), b = 0.5 * (n.z / len + 1.0);
            r = r < 0 ? 0 : (r > 1 ? 1 : r);
            g = g < 0 ? 0 : (g > 1 ? 1 : g);
            b = b < 0 ? 0 : (b > 1 ? 1 : b);
            out.emission = mk(r, g, b);
            break;
        }
    }
}

}  // namespace rt
