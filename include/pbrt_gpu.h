/*
 * pbrt_gpu.h -- C ABI of the MI355X-native path-tracing hot path.
 *
 * This is the drop-in boundary.  Everything above it
 * (scene-file front end, graphics state, BVH construction, Film::WriteImage)
 * is host C++; everything below it is hand-written HIP for gfx950.  The entry
 * points are what a pbrt-v3 maintainer would bind from
 *   Integrator::Render          src/core/integrator.h:57, integrator.cpp:228-339
 *   Scene::Intersect/IntersectP src/core/scene.cpp:45-55
 * Plain C types only: pointers + sizes, no C++ or torch types.
 *
 * Memory rules: every pointer in PgSceneDesc is caller-owned HOST memory and
 * is copied at pg_scene_create(); the device copy is owned by the PgScene and
 * freed by pg_scene_destroy().  Film/ray/hit buffers are caller-owned and may
 * live in host or device memory (PgMemKind).
 *
 * Error model (mirrors pbrt's "report and continue", error.cpp:62-102, but
 * with codes): every call returns PG_OK (0) or a negative PgStatus;
 * pg_last_error() returns a thread-local message.  No C++ exceptions cross
 * the ABI.
 */
#ifndef PBRT_GPU_H
#define PBRT_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 29

typedef enum PgStatus {
    PG_OK = 0,
    PG_ERR_INVALID = -1,     /* bad argument / malformed scene description   */
    PG_ERR_UNSUPPORTED = -2, /* feature outside the implemented closed set   */
    PG_ERR_DEVICE = -3,      /* HIP runtime error (no GPU, OOM, launch fail) */
    PG_ERR_OVERFLOW = -4     /* a fixed-size device queue overflowed         */
} PgStatus;

typedef enum PgMemKind { PG_MEM_HOST = 0, PG_MEM_DEVICE = 1 } PgMemKind;

/* ---- flattened scene ---------------------------------------------------- */

/* Bit-identical to pbrt's LinearBVHNode (src/accelerators/bvh.cpp:95-104):
 * depth-first order, first child = index+1, second child = offset.         */
typedef struct PgBVHNode {
    float bmin[3];
    float bmax[3];
    int32_t offset;  /* leaf: first primitive; interior: second child index */
    uint16_t nprims; /* 0 => interior                                        */
    uint8_t axis;    /* interior: split axis                                 */
    uint8_t pad;
} PgBVHNode;

/* Triangle flags (per primitive, BVH order). */
#define PG_TRI_FLIP_NORMAL 1u /* reverseOrientation ^ transformSwapsHandedness (triangle.cpp:346-348) */
#define PG_TRI_REVERSE_ORIENTATION 2u /* reverseOrientation alone (triangle.cpp:416)            */
#define PG_TRI_HAS_N 4u       /* mesh has per-vertex shading normals  */
#define PG_TRI_HAS_UV 8u      /* mesh has per-vertex uv               */
#define PG_TRI_HAS_S 16u      /* mesh has per-vertex tangents         */
#define PG_PRIM_SPHERE 32u    /* not a triangle: spheres[indices[3*k]] (PgSphere below) */
#define PG_TRI_ALPHA 128u      /* the mesh has an alpha / shadow-alpha texture: alphas[tri_alpha[k]] (triangle.cpp:333-338, :531-569) */
#define PG_PRIM_INSTANCE 64u  /* a TransformedPrimitive: instances[indices[3*k]] (PgInstance below).  Among the top-level primitives, and -- ABI 29 --
                                 among an object definition's: pbrtShape under an animated transformation between ObjectBegin and ObjectEnd adds its
                                 TransformedPrimitive to the instance being defined (api.cpp:1386-1419), so a hit can lie under two transforms, the
                                 ObjectInstance's around the moving shape's.  One level: the object such a primitive wraps holds shapes only (the
                                 reference has no ObjectInstance inside a definition, api.cpp:1549-1552). */

typedef enum PgMaterialType {
    PG_MAT_NONE = 0,   /* no material: primitive is a medium boundary (bsdf == nullptr) */
    PG_MAT_MATTE = 1,  /* materials/matte.cpp:45-62   */
    PG_MAT_PLASTIC = 2,/* materials/plastic.cpp:45-70 */
    PG_MAT_MIRROR = 3, /* materials/mirror.cpp:44-56: SpecularReflection(Kr, FresnelNoOp)              */
    PG_MAT_GLASS = 4,  /* materials/glass.cpp:45-96, smooth only: FresnelSpecular(Kr, Kt, 1, eta)      */
    PG_MAT_TEXTURED = 6, /* a material with a non-constant texture among its parameters: its BxDFs are evaluated per hit from
                          textured[textured_index] (PgTexturedMaterial below) */
    PG_MAT_LOBES = 5   /* any other material (uber, metal, substrate, translucent, mix, rough glass ...): defined by
                          its BxDF list alone, bxdfs[first_bxdf .. first_bxdf + n_bxdfs)                          */
} PgMaterialType;

/* One BxDF of a material's BSDF (core/reflection.h).  With constant textures Material::ComputeScatteringFunctions adds
 * the same BxDFs with the same parameters at every point, so a material IS its BxDF list (in the order of the Add()
 * calls) plus BSDF::eta; only the shading frame varies per hit.  Every material carries its list, also types 1-4, whose
 * `type` additionally tells the device that the list has that material's fixed shape (a specialised kernel then reads the
 * legacy fields instead). */
typedef enum PgBxDFType {
    PG_BXDF_LAMBERT_R = 1,     /* LambertianReflection(R),                     reflection.h:355-374 */
    PG_BXDF_LAMBERT_T = 2,     /* LambertianTransmission(T),                   reflection.h:376-395 */
    PG_BXDF_OREN_NAYAR = 3,    /* OrenNayar(R, sigma): on_a, on_b = its A, B,  reflection.h:397-433 */
    PG_BXDF_SPECULAR_R = 4,    /* SpecularReflection(R, fresnel),              reflection.h:274-298 */
    PG_BXDF_SPECULAR_T = 5,    /* SpecularTransmission(T, etaA, etaB, Radiance), reflection.h:300-329 */
    PG_BXDF_FRESNEL_SPECULAR = 6, /* FresnelSpecular(R, T, etaA, etaB, Radiance), reflection.h:331-353 */
    PG_BXDF_MICROFACET_R = 7,  /* MicrofacetReflection(R, TrowbridgeReitz(alpha_x, alpha_y), fresnel), reflection.h:435-457 */
    PG_BXDF_MICROFACET_T = 8,  /* MicrofacetTransmission(T, TrowbridgeReitz, etaA, etaB, Radiance),    reflection.h:459-484 */
    PG_BXDF_FRESNEL_BLEND = 9  /* FresnelBlend(Rd = R, Rs = T, TrowbridgeReitz),                         reflection.h:486-510 */
} PgBxDFType;
typedef enum PgFresnelType {
    PG_FRESNEL_NOOP = 0,       /* FresnelNoOp: 1 */
    PG_FRESNEL_DIELECTRIC = 1, /* FresnelDielectric(eta_a, eta_b) */
    PG_FRESNEL_CONDUCTOR = 2   /* FresnelConductor(1, cond_eta, cond_k) */
} PgFresnelType;
#define PG_MAX_BXDFS 8         /* BSDF::MaxBxDFs, reflection.h:209 */
#define PG_MAX_BXDF_SCALES 3   /* nesting depth of MixMaterial this ABI carries */
typedef struct PgBxDF {
    int32_t type;        /* PgBxDFType */
    int32_t fresnel;     /* PgFresnelType, for SPECULAR_R and MICROFACET_R */
    float R[3];          /* R; Rd of FresnelBlend */
    float T[3];          /* T; Rs of FresnelBlend */
    float eta_a, eta_b;  /* dielectric indices: Fresnel term and refraction */
    float cond_eta[3], cond_k[3]; /* conductor */
    float alpha_x, alpha_y;       /* TrowbridgeReitzDistribution's alphas as its constructor leaves them (microfacet.h:112-116) */
    float on_a, on_b;    /* OrenNayar A, B */
    int32_t n_scales;    /* ScaledBxDF wrappers around this BxDF (mixmat.cpp:60-65), innermost first */
    float scale[PG_MAX_BXDF_SCALES][3];
} PgBxDF;

typedef struct PgMaterial {
    int32_t type;    /* PgMaterialType */
    float kd[3];     /* constant-texture value, already clamped >= 0 is NOT assumed */
    float ks[3];
    float sigma;     /* matte: Oren-Nayar sigma (degrees); 0 => Lambertian */
    float roughness; /* plastic */
    int32_t remap_roughness;
    float kr[3];     /* mirror, glass */
    float kt[3];     /* glass */
    float eta;       /* glass: index of refraction */
    int32_t first_bxdf, n_bxdfs; /* this material's BxDFs in PgSceneDesc.bxdfs, n_bxdfs <= PG_MAX_BXDFS */
    float bsdf_eta;  /* BSDF::eta (reflection.h:167-172): the index the path's Russian roulette sees */
    int32_t textured_index; /* PG_MAT_TEXTURED: index into PgSceneDesc.textured */
} PgMaterial;

/* ---- textures (core/texture.h and the classes under textures/) ----------------------------------------------------------------------
 * A parameter of a material or of another texture is either a constant or a reference to a texture node. */
typedef struct PgTexRef {
    int32_t tex;     /* >= 0: textures[tex]; -1: the constant v (a float parameter uses v[0]) */
    float v[3];
} PgTexRef;
typedef enum PgTextureType {
    PG_TEX_SCALE = 1,          /* ScaleTexture: tex1 * tex2,                    textures/scale.h:49-66   */
    PG_TEX_MIX = 2,            /* MixTexture: (1 - amount) * tex1 + amount * tex2, textures/mix.h:49-68  */
    PG_TEX_CHECKERBOARD_2D = 3,/* Checkerboard2DTexture, aa_none or closed-form box filter, checkerboard.h:52-110 */
    PG_TEX_CHECKERBOARD_3D = 4,/* Checkerboard3DTexture,                         checkerboard.h:112-135 */
    PG_TEX_UV = 5,             /* UVTexture (spectrum),                          textures/uv.h:49-66      */
    PG_TEX_BILERP = 6,         /* BilerpTexture,                                 textures/bilerp.h:49-69  */
    PG_TEX_IMAGEMAP = 7,       /* ImageTexture over images[image],               textures/imagemap.h:77-123 */
    /* Perlin-noise textures over IdentityMapping3D (w2t) -- FBm / Turbulence of core/texture.cpp:164-246 */
    PG_TEX_FBM = 8,            /* FBmTexture: octaves, omega,                    textures/fbm.h:49-65     */
    PG_TEX_WRINKLED = 9,       /* WrinkledTexture: octaves, omega,               textures/wrinkled.h:49-65 */
    PG_TEX_WINDY = 10,         /* WindyTexture,                                  textures/windy.h:49-64   */
    PG_TEX_MARBLE = 11,        /* MarbleTexture (spectrum): octaves, omega, noise_scale, variation, textures/marble.h:49-92 */
    PG_TEX_DOTS = 12           /* DotsTexture over a 2D mapping: tex1 = outsideDot, tex2 = insideDot, textures/dots.h:49-83 */
} PgTextureType;
typedef enum PgMappingType {   /* TextureMapping2D, core/texture.h:51-110 */
    PG_MAP_UV = 0, PG_MAP_SPHERICAL = 1, PG_MAP_CYLINDRICAL = 2, PG_MAP_PLANAR = 3
} PgMappingType;
typedef struct PgTexture {
    int32_t type;              /* PgTextureType */
    int32_t is_float;          /* Texture<Float> (1) or Texture<Spectrum> (0) */
    int32_t mapping;           /* PgMappingType (2D textures) */
    float su, sv, du, dv;      /* UVMapping2D; du, dv also PlanarMapping2D's ds, dt */
    float vs[3], vt[3];        /* PlanarMapping2D */
    float w2t[16];             /* WorldToTexture: spherical / cylindrical mappings and IdentityMapping3D */
    PgTexRef tex1, tex2, amount; /* operands (amount: the float texture of mix) */
    int32_t aa_none;           /* checkerboard: AAMethod::None */
    float v00[3], v01[3], v10[3], v11[3]; /* bilerp */
    int32_t image;             /* imagemap: index into PgSceneDesc.images */
    int32_t octaves;           /* fbm / wrinkled / marble */
    float omega, noise_scale, variation; /* "roughness"; marble's "scale" and "variation" */
} PgTexture;
/* MIPMap<Float> / MIPMap<RGBSpectrum> (core/mipmap.h): the pyramid as its constructor leaves it (power-of-two resampling,
 * box-filtered levels), levels row-major in texels[], 1 or 3 floats per texel; level i is max(1, width >> i) x max(1, height >> i). */
#define PG_MAX_MIP_LEVELS 32  /* 1 + Log2Int(max resolution) for any int resolution (mipmap.h:141) */
typedef struct PgImage {
    int32_t is_float;          /* MIPMap<Float> */
    int32_t n_levels;
    int32_t width, height;     /* level 0 */
    int32_t wrap;              /* ImageWrap: 0 repeat, 1 black, 2 clamp */
    int32_t trilinear;         /* doTrilinear; otherwise EWA */
    float max_anisotropy;
    int64_t level_offset[PG_MAX_MIP_LEVELS]; /* in floats, into PgSceneDesc.texels */
} PgImage;
/* A material whose BxDF list depends on the hit: the material's kind and its parameters as the Create*Material
 * functions read them; ComputeScatteringFunctions is evaluated per hit on the device. */
typedef enum PgMaterialKind {
    PG_KIND_MATTE = 1, PG_KIND_PLASTIC = 2, PG_KIND_MIRROR = 3, PG_KIND_GLASS = 4, PG_KIND_UBER = 5, PG_KIND_METAL = 6,
    PG_KIND_SUBSTRATE = 7, PG_KIND_TRANSLUCENT = 8, PG_KIND_MIX = 9
} PgMaterialKind;
/* TriangleMesh::alphaMask / shadowAlphaMask (shapes/triangle.h:66-67): float textures that cut hits out of a mesh --
 * Triangle::Intersect rejects a hit whose alpha evaluates to 0, IntersectP one whose alpha or shadow alpha does. */
typedef struct PgAlphaMask {
    int32_t has_alpha, has_shadow_alpha;
    PgTexRef alpha, shadow_alpha;
} PgAlphaMask;
typedef struct PgTexturedMaterial {
    int32_t kind;              /* PgMaterialKind */
    /* spectrum parameters, by kind: matte Kd | plastic Kd Ks | mirror Kr | glass Kr Kt | uber Kd Ks Kr Kt opacity |
     * metal eta k | substrate Kd Ks | translucent Kd Ks reflect transmit | mix amount */
    PgTexRef s[5];
    /* float parameters: matte sigma | plastic roughness | glass uroughness vroughness index | uber roughness uroughness
     * vroughness eta | metal roughness uroughness vroughness | substrate uroughness vroughness | translucent roughness */
    PgTexRef f[4];
    int32_t has_u, has_v;      /* uber / metal: "uroughness" / "vroughness" were given (GetFloatTextureOrNull) */
    int32_t remap_roughness;
    int32_t sub[2];            /* mix: the two materials (indices into materials[]) */
    int32_t has_bump;          /* "bumpmap" given: Material::Bump (material.cpp:46-85) displaces the shading geometry first */
    PgTexRef bump;
} PgTexturedMaterial;

/* scene.lights, in declaration order (api.cpp:1308-1327 for LightSource, :1353-1363 for area lights):
 * one DiffuseAreaLight per emissive triangle (lights/diffuse.cpp:43-87), plus the delta lights
 * PointLight (lights/point.cpp), SpotLight (lights/spot.cpp) and DistantLight (lights/distant.cpp). */
typedef enum PgLightType {
    PG_LIGHT_AREA = 0,   /* DiffuseAreaLight on triangle `prim`           */
    PG_LIGHT_POINT = 1,
    PG_LIGHT_SPOT = 2,
    PG_LIGHT_DISTANT = 3,
    PG_LIGHT_INFINITE = 4, /* InfiniteAreaLight (lights/infinite.cpp), constant or over an environment map */
    PG_LIGHT_PROJECTION = 5, /* ProjectionLight (lights/projection.cpp:44-101): a point light behind a slide */
    PG_LIGHT_GONIO = 6     /* GonioPhotometricLight (lights/goniometric.h:50-88): a point light with a measured distribution */
} PgLightType;
/* Light::flags & (DeltaPosition | DeltaDirection), light.h:55-58 */
#define PG_LIGHT_IS_DELTA(type) ((type) == PG_LIGHT_POINT || (type) == PG_LIGHT_SPOT || (type) == PG_LIGHT_DISTANT || (type) == PG_LIGHT_PROJECTION || (type) == PG_LIGHT_GONIO)

typedef struct PgLight {
    int32_t type;      /* PgLightType                                               */
    int32_t prim;      /* area: index (BVH order) of the emitting triangle, else -1 */
    float L[3];        /* area: Lemit = L * scale; point/spot: I * scale; distant: L * scale */
    int32_t two_sided; /* area                                                      */
    float area;        /* area: shape->Area() as computed by the host               */
    float pos[3];      /* point/spot: pLight (world); distant: wLight (unit, world) */
    float w2l[9];      /* spot: upper 3x3 of WorldToLight, row-major (Falloff)      */
    float cos_total_width, cos_falloff_start; /* spot (spot.cpp:48-49)              */
    float world_radius; /* distant, infinite: Preprocess()'s bounding-sphere radius (distant.h:55-57) */
    /* infinite (infinite.cpp:46-85): upper 3x3 of LightToWorld; Lmap = images[env_image] (MIPMap<RGBSpectrum> of the map
     * times L * scale, or 1x1 without "mapname"; wrap repeat); the Distribution2D over env_nu x env_nv = (2 width) x
     * (2 height) scalar values Lmap->Lookup(st, fwidth).y() * sin(theta) at env_tables[env_table]: per row v its func[nu],
     * cdf[nu+1], funcInt (2 nu + 2 floats), then the marginal over rows: func[nv], cdf[nv+1], funcInt (sampling.h:57-150).
     * env_power = Lmap->Lookup((.5, .5), .5), the radiance Power() integrates (infinite.cpp:87-91).                      */
    float l2w[9];
    int32_t env_image, env_nu, env_nv;
    int64_t env_table;
    float env_power[3];
    /* projection / goniometric: pos = pLight, L = I * scale, w2l = upper 3x3 of WorldToLight; env_image = the MIPMap of
     * "mapname" (-1: no map, the light is uniform inside its frustum / sphere), env_power = its Lookup((.5, .5), .5) (or 1);
     * projection only: lightProjection (row-major), screenBounds (x0, y0, x1, y1), hither, cos_total_width (projection.cpp:44-69) */
    float proj[16];
    float screen[4];
    float hither;
} PgLight;

typedef enum PgLightStrategy {
    PG_LIGHTS_UNIFORM = 0,
    PG_LIGHTS_POWER = 1,
    PG_LIGHTS_SPATIAL = 2
} PgLightStrategy;

/* A Sphere shape (shapes/sphere.h:46-79), kept in object space as the reference keeps it.  A primitive k with
 * PG_PRIM_SPHERE set in tri_flags[k] is spheres[indices[3*k]]; its tri_material / tri_light entries mean what they mean
 * for a triangle, and an area light whose `prim` is such a primitive samples the sphere (sphere.cpp:205-304). */
typedef enum PgQuadricShape {
    PG_SHAPE_SPHERE = 0,        /* shapes/sphere.cpp   */
    PG_SHAPE_CYLINDER = 1,      /* shapes/cylinder.cpp: radius, z_min, z_max, phi_max */
    PG_SHAPE_DISK = 2,          /* shapes/disk.cpp: height, radius, inner_radius, phi_max */
    /* geometry only: the reference has no Sample() for these three (cone.cpp:205-208 etc.), so they cannot be area lights */
    PG_SHAPE_CONE = 3,          /* shapes/cone.cpp: radius, height, phi_max (z_min = 0, z_max = height) */
    PG_SHAPE_PARABOLOID = 4,    /* shapes/paraboloid.cpp: radius, z_min, z_max, phi_max */
    PG_SHAPE_HYPERBOLOID = 5    /* shapes/hyperboloid.cpp: p1, p2, ah, ch, z_min, z_max, phi_max (radius = rMax) */
} PgQuadricShape;
typedef struct PgSphere {       /* a quadric: the record began as the sphere's and kept its name */
    float o2w[16], w2o[16];     /* ObjectToWorld / WorldToObject, row-major (its m and mInv, transform.h:112-205) */
    float radius, z_min, z_max, theta_min, theta_max, phi_max; /* as the constructor clamps them, sphere.h:50-60 */
    int32_t reverse_orientation;/* Shape::reverseOrientation               */
    int32_t swaps_handedness;   /* Shape::transformSwapsHandedness         */
    int32_t shape;              /* PgQuadricShape */
    float height, inner_radius; /* disk */
    float area;                 /* Shape::Area() */
    float p1[3], p2[3], ah, ch; /* hyperboloid: the end points as its constructor leaves them and its implicit coefficients (hyperboloid.cpp:43-65) */
} PgSphere;

/* Object instancing (api.cpp:1509-1588).  An object definition is a run of primitives in the primitive arrays after the
 * n_tris top-level ones and, when it has more than one primitive, its own BVHAccel: a run of nodes after the n_nodes
 * top-level ones, laid out exactly as that BVHAccel's LinearBVHNode array (child / primitive offsets relative to the
 * object's own first node / first primitive).  An instance is a primitive with PG_PRIM_INSTANCE: top-level, or (ABI 29) a moving
 * shape's TransformedPrimitive inside another object definition's run. */
typedef struct PgObject {
    int32_t first_node, n_nodes; /* n_nodes == 0: a single primitive without an accelerator (api.cpp:1567) */
    int32_t first_prim, n_prims;
} PgObject;
typedef struct PgInstance {      /* TransformedPrimitive (primitive.h:92-117) */
    float i2w[16], w2i[16];      /* InstanceToWorld (the AnimatedTransform's startTransform) and its inverse, row-major */
    int32_t object;
    int32_t identity;            /* Transform::IsIdentity() of InstanceToWorld (primitive.cpp:86-87) */
    /* ABI 26 -- a moving instance or shape (api.cpp:1386-1419 wraps an animated shape's primitives in a TransformedPrimitive too, :1576-1586 an
     * animated ObjectInstance): TransformedPrimitive::Intersect[P] interpolate PrimitiveToWorld at the ray's time (primitive.cpp:78-80,
     * :99-101; AnimatedTransform::Interpolate, transform.cpp:1144-1169): the start transform up to time[0], the end transform from time[1]
     * on, in between Translate(lerp T) * Slerp(R).ToTransform() * Transform(lerp S) -- whose inverse is the product of the three
     * factors' inverses, Transform(lerp S)'s by Gauss-Jordan (transform.cpp:83-135) -- of the two decompositions (Decompose,
     * :1103-1142), which the host computes once.  The top-level primitive's bounds are AnimatedTransform::MotionBounds
     * (transform.cpp:1215-1247): Union(start box, end box), or with rotation (Dot(R[0], R[1]) < 0.9995) the corners' paths bounded at the
     * zeros of their derivatives -- pbrt_host_motion_bounds (pbrt_host.h) for a host that builds its own BVH.  animated = 0: the fields below are unused. */
    int32_t animated;            /* AnimatedTransform::actuallyAnimated */
    float time[2];               /* startTime, endTime (the file's TransformTimes) */
    float i2w_end[16], w2i_end[16]; /* endTransform and its inverse */
    float T[2][3];
    float R[2][4];               /* (v.x, v.y, v.z, w); R[1] already flipped onto R[0]'s hemisphere (transform.cpp:410) */
    float S[2][9];               /* the upper 3x3 of S[0], S[1], row-major */
} PgInstance;

/* HomogeneousMedium (media/homogeneous.h:49-71) with its HenyeyGreenstein phase function (core/medium.h:86-100).
 * tri_medium_inside / tri_medium_outside give each primitive's MediumInterface (core/medium.h:102-116) as indices into
 * media[], -1 = no medium (vacuum); a primitive whose two sides agree is not a medium transition and a ray keeps its
 * current medium across it (primitive.cpp:121-125). */
typedef struct PgMedium {
    float sigma_a[3], sigma_s[3], sigma_t[3];
    float g;
} PgMedium;

/* GridDensityMedium (media/grid.h:49-96), the "heterogeneous" medium of MakeMedium (api.cpp:700-722).  Its PgMedium entry holds
 * sigma_a, sigma_s (after "scale"), g, and sigma_t = sigma_a + sigma_s in all three channels -- the constructor requires a
 * spectrally uniform sigma_t (grid.h:64-68), the host reports the same error; PgSceneDesc.media_grid[m] is the index of medium
 * m's grid here, -1 for a HomogeneousMedium.
 * libpbrt_gpu.so renders it since round 3: k_shade<., ., ., GRID> in two phases around the transmittance rays, k_through<., GRID>
 * (DESIGN.md section 4 "Heterogeneous media"). */
typedef struct PgDensityGrid {
    int32_t nx, ny, nz;
    int32_t reserved;
    int64_t density_offset;     /* first of nx*ny*nz floats in PgSceneDesc.grid_density, x fastest: density[(z*ny + y)*nx + x] (grid.h:77-81) */
    float sigma_t;              /* (sigma_a + sigma_s)[0] */
    float inv_max_density;      /* 1 / max of the density values (grid.h:69-72) */
    float world_to_medium[16];  /* Inverse(medium2world * Translate(p0) * Scale(p1 - p0)).m, row-major: world -> the unit cube */
} PgDensityGrid;

/* SubsurfaceMaterial / KdSubsurfaceMaterial (materials/subsurface.cpp, kdsubsurface.cpp) with constant parameters.  The
 * surface BSDF is the material's BxDF list like any other material's (FresnelSpecular, or the two microfacet lobes:
 * subsurface.cpp:57-86); PgSceneDesc.material_bssrdf[m] >= 0 says that ComputeScatteringFunctions also sets
 * si->bssrdf = TabulatedBSSRDF(si, material, mode, eta, sigma_a, sigma_s, table) (core/bssrdf.h:141-165) -- unless R and T are
 * both black (the early return at subsurface.cpp:55: then the material has no BxDFs and no BSSRDF).  Identity matters: the probe
 * rays of SeparableBSSRDF::Sample_Sp accept hits on primitives with the SAME Material object (bssrdf.cpp:301), so every
 * Material / MakeNamedMaterial directive of these two types owns its own PgMaterial entry.
 * `table` is BSSRDFTable(n_rho, n_radius) after ComputeBeamDiffusionBSSRDF(g, eta) (bssrdf.cpp:149-180), laid out in
 * bssrdf_tables as rhoSamples[n_rho], radiusSamples[n_radius], profile[n_rho * n_radius], rhoEff[n_rho],
 * profileCDF[n_rho * n_radius]; materials with equal (g, eta) share one table.
 * libpbrt_gpu.so renders the BSSRDF branch of Li (path.cpp:152-174, volpath.cpp:151-176) since round 3: k_shade<., ., SSS>,
 * k_sss_probe, k_sss_exit (DESIGN.md section 4 "Subsurface scattering"). */
typedef struct PgBSSRDF {
    float eta;
    /* TabulatedBSSRDF::sigma_t and ::rho as its constructor derives them (bssrdf.h:146-150) from the coefficients the material hands
     * it: scale * Clamp(sigma_a / sigma_s) (subsurface.cpp:87-88), or SubsurfaceFromDiffuse(Kd, scale * mfp) (kdsubsurface.cpp:88-91) */
    float sigma_t[3], rho[3];
    int32_t n_rho, n_radius;      /* 100, 64 (subsurface.h:73) */
    int64_t table;                /* first float of the table in PgSceneDesc.bssrdf_tables */
    /* A texture among the material's parameters (or a bump map): the surface BSDF is then a PG_MAT_TEXTURED material of kind
     * PG_KIND_GLASS (the two ComputeScatteringFunctions build the same lobes from Kr, Kt, uroughness, vroughness and eta as
     * glass.cpp does from its parameters), there is a BSSRDF at a hit iff that BSDF has a BxDF (the early return), and sigma_t / rho are
     * derived per hit as the constructor does from
     *   textured == 1 (subsurface):   sigma_a = scale * Clamp(a),  sigma_s = scale * Clamp(b)            (subsurface.cpp:87-88)
     *   textured == 2 (kdsubsurface): SubsurfaceFromDiffuse(table, Clamp(a) = Kd, scale * Clamp(b) = mfp) (kdsubsurface.cpp:88-91)
     * textured == 0: sigma_t / rho above are final. */
    int32_t textured;
    float scale;
    PgTexRef a, b;
    /* SeparableBSSRDF::material as an index into materials[]: the probe rays keep hits on primitives of THIS material.  The
     * material's own index -- except for a MixMaterial whose first component is a subsurface material: si->bssrdf is then that
     * component's (mixmat.cpp:52-53), whose `material` is the component, not the mix, so only primitives that carry the component
     * itself are admissible exit points. */
    int32_t match_material;
} PgBSSRDF;

typedef struct PgSceneDesc {
    int32_t abi_version;        /* PG_ABI_VERSION */
    /* acceleration structure, BVHAccel after flattenBVHTree (bvh.cpp:640-658) */
    int32_t n_nodes;
    const PgBVHNode *nodes;
    /* primitives in BVH (orderedPrims) order */
    int32_t n_tris;
    const int32_t *indices;     /* 3*n_tris vertex indices                  */
    const uint32_t *tri_flags;  /* n_tris PG_TRI_* bits                     */
    const int32_t *tri_material;/* n_tris index into materials              */
    const int32_t *tri_light;   /* n_tris index into lights, -1 = not emissive */
    /* vertex data, world space (triangle.cpp:73-74) */
    int32_t n_verts;
    const float *P;             /* 3*n_verts                                */
    const float *N;             /* 3*n_verts or NULL                        */
    const float *UV;            /* 2*n_verts or NULL                        */
    const float *S;             /* 3*n_verts or NULL                        */
    int32_t n_materials;
    const PgMaterial *materials;
    int32_t n_lights;
    const PgLight *lights;
    int32_t light_strategy;     /* PgLightStrategy as resolved by CreateLightSampleDistribution (lightdistrib.cpp:48-66) */
    /* Halton digit permutations (lowdiscrepancy.cpp:2490-2504), first n_perm_dims
     * prime bases concatenated; perm_sums[d] = offset of base d.  May be absent
     * (n_perm_dims = 0) when sobol_matrices is given and only that sampler is used. */
    int32_t n_perm_dims;
    const uint16_t *perms;
    const int32_t *perm_sums;   /* n_perm_dims+1 entries */
    int32_t n_spheres;
    const PgSphere *spheres;
    int32_t n_bxdfs;
    const PgBxDF *bxdfs;
    /* instancing: nodes[] holds n_nodes_all >= n_nodes entries, the primitive arrays n_prims_all >= n_tris entries */
    int32_t n_nodes_all, n_prims_all;
    int32_t n_objects;
    const PgObject *objects;
    int32_t n_instances;
    const PgInstance *instances;
    int32_t n_textures;
    const PgTexture *textures;
    int32_t n_textured;
    const PgTexturedMaterial *textured;
    int32_t n_images;
    const PgImage *images;
    int64_t n_texel_floats;
    const float *texels;
    int32_t n_media;
    const PgMedium *media;
    const int32_t *tri_medium_inside, *tri_medium_outside; /* n_prims_all each, or both NULL when no primitive has a medium */
    int32_t n_alphas;
    const PgAlphaMask *alphas;
    const int32_t *tri_alpha;   /* per primitive (n_prims_all): index into alphas for triangles with PG_TRI_ALPHA; may be NULL */
    int64_t n_env_floats;
    const float *env_tables;    /* the infinite lights' Distribution2D tables (PgLight.env_table) */
    const float *ewa_lut;       /* MIPMap::weightLut, 128 entries (mipmap.h:178-184); may be NULL without images */
    /* SobolSampler tables (core/sobolmatrices.h:49-52), NULL unless PgRenderDesc.sampler == 1: SobolMatrices32
     * [1024 * 52], VdCSobolMatrices [25][52], VdCSobolMatricesInv [26][52] */
    const uint32_t *sobol_matrices;
    const uint64_t *vdc_sobol;
    const uint64_t *vdc_sobol_inv;
    const int32_t *noise_perm;  /* NoisePerm, 512 entries (core/texture.cpp:51-78); NULL unless a noise texture is present */
    const uint32_t *cmaxmin;    /* CMaxMinDist [17][32] (core/lowdiscrepancy.cpp:249-...); NULL unless the sampler is maxmindist */
    /* ABI 23: GridDensityMedium tables; all zero / NULL when every medium is homogeneous */
    int32_t n_grids;
    const PgDensityGrid *grids;
    const int32_t *media_grid;  /* n_media entries: index into grids, -1 = HomogeneousMedium; NULL when n_grids == 0 */
    int64_t n_density_floats;
    const float *grid_density;
    /* ABI 24: subsurface scattering tables; all zero / NULL when no material has a BSSRDF */
    int32_t n_bssrdfs;
    const PgBSSRDF *bssrdfs;
    const int32_t *material_bssrdf; /* n_materials entries: index into bssrdfs, -1 = none; NULL when n_bssrdfs == 0 */
    int64_t n_bssrdf_floats;
    const float *bssrdf_tables;
} PgSceneDesc;

/* ---- render description -------------------------------------------------- */

typedef enum PgSamplerKind {
    PG_SAMPLER_HALTON = 0,        /* samplers/halton.cpp  */
    PG_SAMPLER_SOBOL = 1,         /* samplers/sobol.cpp   */
    PG_SAMPLER_RANDOM = 2,        /* samplers/random.cpp: every value straight from the tile's RNG */
    PG_SAMPLER_STRATIFIED = 3,    /* samplers/stratified.cpp */
    PG_SAMPLER_ZEROTWO = 4,       /* samplers/zerotwosequence.cpp ("02sequence", "lowdiscrepancy"); spp = the power of two it rounds up to */
    PG_SAMPLER_MAXMINDIST = 5     /* samplers/maxmin.cpp; needs PgSceneDesc.cmaxmin; spp rounded as its constructor does */
} PgSamplerKind;

typedef struct PgRenderDesc {
    int32_t abi_version;
    /* camera: PerspectiveCamera (cameras/perspective.cpp:45-144) or OrthographicCamera (cameras/orthographic.cpp:44-118) */
    int32_t integrator;         /* 0 = PathIntegrator (integrators/path.cpp), 1 = VolPathIntegrator (integrators/volpath.cpp) */
    int32_t camera_medium;      /* Camera::medium: index into PgSceneDesc.media, -1 = none */
    int32_t camera_type;        /* 0 = perspective, 1 = orthographic, 2 = environment (cameras/environment.cpp:43-56) */
    float raster_to_camera[16]; /* row-major Matrix4x4 */
    float dx_camera[3], dy_camera[3]; /* ProjectiveCamera::dxCamera / dyCamera (perspective.cpp:60-63, orthographic.cpp:57-58): ray differentials */
    float camera_to_world[16];  /* AnimatedTransform CameraToWorld: its startTransform (the only one of a camera that does not move) */
    float lens_radius, focal_distance;
    float shutter_open, shutter_close;
    /* A moving camera (ABI 25): Camera::CameraToWorld is an AnimatedTransform (core/transform.h:331-362, api.cpp:1725-1730) that every
     * camera ray is carried through at its own time = Lerp(CameraSample::time, shutterOpen, shutterClose) (perspective.cpp:89-91, :139):
     * AnimatedTransform::operator()(Ray) (transform.cpp:1171-1181) = the start transform up to camera_time[0], the end transform from
     * camera_time[1] on, and in between Translate(lerp T) * Slerp(R).ToTransform() * Transform(lerp S) (Interpolate, :1144-1169) of the
     * two transforms' decompositions (Decompose, :1103-1142), which the host computes once.  camera_animated = 0: the fields are unused. */
    int32_t camera_animated;         /* AnimatedTransform::actuallyAnimated */
    float camera_time[2];            /* startTime, endTime (the file's TransformTimes) */
    float camera_to_world_end[16];   /* endTransform */
    float camera_T[2][3];            /* T[0], T[1] */
    float camera_R[2][4];            /* R[0], R[1] as (v.x, v.y, v.z, w); R[1] already flipped onto R[0]'s hemisphere (transform.cpp:410) */
    float camera_S[2][9];            /* the upper 3x3 of S[0], S[1], row-major */
    /* film (film.cpp:45-86) */
    int32_t full_res[2];
    int32_t cropped_pixel_bounds[4]; /* x0,y0,x1,y1 */
    int32_t sample_bounds[4];
    float filter_radius[2];          /* Filter::radius (filter.h:50-66)                                        */
    /* Film's 16x16 filterTable (film.cpp:68-77), filled by the host for box / gaussian / mitchell / sinc / triangle.
     * filter_general = 0: box filter with radius <= 0.5 -- every sample lands in its own pixel (a rare second pixel
     * is reported as a PgStraySample) and a tile's film block is its 16x16 pixels.  Only for frames in which no film
     * position `(float)pixel + u` can round UP onto the next pixel: pgh_box_filter_needs_gather() below says so, and
     * pg_render refuses filter_general = 0 where it returns 1 (such a sample is added to the next pixel BEFORE that
     * pixel's own samples, FilmTile::AddSample, film.h:121-161: a summation order only the gathering path reproduces).
     * filter_general = 1: any other filter -- a tile's film block is its FilmTile pixel bounds (film.cpp:95-106)
     * before clipping: (16 + halo[0] + halo[2]) x (16 + halo[1] + halo[3]) entries, row-major, entry (0,0) = pixel
     * (tile.x0 - halo[0], tile.y0 - halo[1]); no stray samples are produced.                                     */
    int32_t filter_general;
    int32_t tile_halo[4];            /* x-low, y-low, x-high, y-high                                            */
    int32_t tile_pixels;             /* PgFilmPixel entries per tile: 256, or the block size above              */
    float filter_table[256];
    float film_scale;
    float max_sample_luminance;
    /* sampler: HaltonSampler (samplers/halton.cpp:65-127) */
    int32_t spp;
    int32_t base_scales[2], base_exponents[2];
    int32_t sample_stride;
    int32_t mult_inverse[2];
    int32_t sample_at_pixel_center;
    /* sampler = 1: SobolSampler (samplers/sobol.h:48-71, sobol.cpp:41-59) instead; spp then is the power of two it rounds up to.
     * sampler = 2 .. 5: the samplers that draw from ONE RNG stream per 16x16 tile, seeded with the tile's index in the
     * full-frame tiling (integrator.cpp:247-248) and consumed pixel by pixel, sample by sample, path by path --
     * PgSamplerKind below.  The device then keeps one path per tile in flight (the order is part of the result). */
    int32_t sampler;
    int32_t sobol_resolution, sobol_log2_resolution;
    int32_t sampler_dims;       /* PixelSampler: nSampledDimensions ("integer dimensions"); beyond them Get1D / Get2D fall back to the RNG */
    int32_t strat_samples[2];   /* StratifiedSampler: xPixelSamples, yPixelSamples (spp = their product) */
    int32_t strat_jitter;       /* StratifiedSampler: jitterSamples */
    /* integrator: PathIntegrator (integrators/path.cpp:190-213) */
    int32_t max_depth;
    float rr_threshold;
    int32_t pixel_bounds[4];
    /* sharding: this call renders the 16x16 tiles t of the full-frame tiling
     * with (t % tile_count) == tile_first ... i.e. t = tile_first + k*tile_step.
     * Single GPU: tile_first=0, tile_step=1.                                 */
    int32_t tile_first, tile_step;
} PgRenderDesc;

/* Can a film position `(float)p + u` of GetCameraSample (sampler.cpp:46-52) round up to p + 1 for a pixel and a sample of the
 * frame `rd` describes (sampler fields, spp and sample_bounds filled in)?  1 = it cannot be ruled out: a box-filter frame must
 * then be rendered with filter_general = 1 (the tile blocks with their one-pixel halo), see above.  Header-only (a front end
 * fills its PgRenderDesc without the device library); the library exports the same function as pg_box_filter_needs_gather().
 *  - HaltonSampler: u0 = RadicalInverse(0, index >> base_exponents[0]), u1 = RadicalInverse(1, index / base_scales[1])
 *    (halton.cpp:118-127); the largest value either takes over the arguments the frame's sample indices reach is found by going
 *    through them, evaluated as the sampler evaluates it (lowdiscrepancy.cpp:389-403, 427-436).  The pixel with the largest
 *    coordinate has the coarsest float spacing, and a tie rounds to p + 1 (its significand is even).  1920 pixels wide: from
 *    sample index 2 097 024 on, i.e. from 68 samples per pixel;
 *  - samples at the pixel centre: never;
 *  - every other sampler (Sobol', the PixelSamplers' RNG): numbers up to OneMinusEpsilon occur, which round up from pixel 1 on. */
static inline int pgh_box_filter_needs_gather(const PgRenderDesc *rd) {
    const int xMax = rd->sample_bounds[2] - 1, yMax = rd->sample_bounds[3] - 1;
    volatile float sx, sy; /* (volatile: the sums are rounded to float whatever the compiler keeps them in) */
    float u0Max = 0.99999994f, u1Max = 0.99999994f; /* OneMinusEpsilon */
    if (rd->sample_at_pixel_center || xMax < 0 || yMax < 0) return 0;
    if (rd->sampler == 0) {
        const uint64_t stride = rd->sample_stride > 1 ? (uint64_t)rd->sample_stride : 1;
        const uint64_t maxIndex = (uint64_t)(rd->spp > 0 ? rd->spp : 1) * stride - 1;
        const uint64_t a0Max = maxIndex >> rd->base_exponents[0], a1Max = maxIndex / (uint64_t)(rd->base_scales[1] > 0 ? rd->base_scales[1] : 1);
        uint64_t a0;
        if (a0Max > ((uint64_t)1 << 26) || a1Max > ((uint64_t)1 << 26)) return 1; /* not enumerated */
        u0Max = u1Max = 0;
        for (a0 = 0; a0 <= a0Max; ++a0) { /* ReverseBits64(a) * 0x1p-64 */
            uint64_t r = 0, v = a0;
            int i;
            float u;
            for (i = 0; i < 64 && v; ++i, v >>= 1) if (v & 1) r |= (uint64_t)1 << (63 - i);
            u = (float)((double)r * 5.4210108624275222e-20);
            if (u > u0Max) { u0Max = u; sx = (float)xMax + u0Max; if (sx >= (float)(xMax + 1)) return 1; } /* one such sample decides */
        }
        for (a0 = 0; a0 <= a1Max; ++a0) { /* RadicalInverseSpecialized<3> */
            const float invBase = (float)1 / (float)3;
            uint64_t reversedDigits = 0, a = a0;
            volatile float invBaseN = 1, u;
            while (a) { const uint64_t next = a / 3, digit = a - next * 3; reversedDigits = reversedDigits * 3 + digit; invBaseN = invBaseN * invBase; a = next; }
            u = (float)reversedDigits * invBaseN;
            if (u > 0.99999994f) u = 0.99999994f;
            if (u > u1Max) { u1Max = u; sy = (float)yMax + u1Max; if (sy >= (float)(yMax + 1)) return 1; }
        }
    }
    sx = (float)xMax + u0Max; sy = (float)yMax + u1Max;
    return (sx >= (float)(xMax + 1) || sy >= (float)(yMax + 1)) ? 1 : 0;
}

/* One film pixel as accumulated by FilmTile::AddSample (film.h:121-161):
 * RGB contribution sum (tile-local, pre-XYZ) and filter weight sum.          */
typedef struct PgFilmPixel { float rgb[3]; float weight; } PgFilmPixel;

/* A sample whose box-filter footprint also covers a neighbouring pixel
 * (film.h:127-132 when the sample offset is exactly 0); applied on the host
 * in the reference's order.                                                  */
typedef struct PgStraySample { int32_t px, py; int32_t src_px, src_py; float rgb[3]; float weight; } PgStraySample;

/* Ray-traversal statistics; the reference's own counters (scene.cpp:40-42,
 * integrator.cpp:48, triangle.cpp:45) plus node fetches, which define the
 * algorithmic byte count of SURVEY.md section 8(d).                          */
typedef struct PgCounters {
    uint64_t camera_rays;
    uint64_t closest_rays; /* Scene::Intersect calls  */
    uint64_t shadow_rays;  /* Scene::IntersectP calls */
    uint64_t node_visits;  /* nodes[cur] fetches (bvh.cpp:672/710), both traversal kernels */
    uint64_t tri_tests;    /* Triangle::Intersect[P] calls as the reference counts them
                              (triangle.cpp:45), i.e. including light_tri_tests          */
    uint64_t light_tri_tests; /* of those: Shape::Pdf's single-triangle test outside the BVH (shape.cpp:80) */
    /* per traversal kernel (closest-hit = BVHAccel::Intersect, shadow = IntersectP) */
    uint64_t closest_node_visits, closest_tri_tests;
    uint64_t shadow_node_visits, shadow_tri_tests;
    uint64_t closest_launches, shadow_launches;
    double closest_ms, shadow_ms; /* HIP-event time inside the traversal kernels */
    double render_ms;
    /* the other kernels of the wavefront (path integrator): launches, items and HIP-event times */
    uint64_t shade_launches, resolve_launches;
    uint64_t shade_items; /* path vertices handed to the shading kernel (= main-queue rays traced) */
    uint64_t mis_rays;    /* of closest_rays: the BSDF-sampled rays of EstimateDirect (integrator.cpp:164-212) */
    double shade_ms, resolve_ms, generate_ms, film_ms;
    /* ABI 27 -- which shading kernels the counted frames ran (a bit mask): bit m = k_shade<m, ...> (0 the baked-in BxDF shapes of matte /
     * plastic / mirror / glass; 1 a material's BxDF list; 3 the packed lists k_material evaluated ahead of the launch; 2 the material
     * evaluators inside the shading kernel: scenes with BSSRDF materials or grid media); PG_SHADING_MATERIAL_PREPASS = k_material ran;
     * PG_SHADING_LISTS_DID_NOT_FIT = a textured scene ran k_shade<2> only because k_material's lists found no room in device memory --
     * same image, about 1.5 x the shading time: a caller that measures should know (bench.py: config.shading_mode). */
    uint64_t shading_modes;
    /* ABI 28 -- the integrators' own statistics (path.cpp:45-46, volpath.cpp:45-47), as the reference prints them under "Integrator":
     * "Zero-radiance paths" = paths_zero_radiance / paths_total (PathIntegrator: the vertices whose BSDF has a non-specular BxDF, and of
     * those the ones whose direct lighting Ld came back black, path.cpp:119-126; volpath does not count them); "Path length" =
     * path_length_sum / path_length_count avg [range path_length_min - path_length_max] over the Li calls' final `bounces`
     * (ReportValue, path.cpp:186 / volpath.cpp:187; min / max are 0 while count is 0); "Volume interactions" / "Surface interactions"
     * (volpath.cpp:87, :99). */
    uint64_t paths_total, paths_zero_radiance;
    uint64_t path_length_sum, path_length_count, path_length_min, path_length_max;
    uint64_t volume_interactions, surface_interactions;
} PgCounters;
#define PG_SHADING_MATERIAL_PREPASS 0x100u
#define PG_SHADING_LISTS_DID_NOT_FIT 0x200u

/* pg_scene_set_option: per-scene switches a host may change between frames (none changes an image).
 * PG_OPT_OVERLAP_SHADOW (0 / 1): each bounce's any-hit launch on a second stream beside the next closest-hit launch (+1 % of a frame; the
 * kernels' HIP-event times in PgCounters then overlap).  Default: the environment's PG_OVERLAP_SHADOW when the scene is created, else 0. */
#define PG_OPT_OVERLAP_SHADOW 1

typedef struct PgScene PgScene;

/* Number of visible HIP devices; negative PgStatus on failure. */
int pg_device_count(void);
/* Select the device subsequent calls on this thread use. */
int pg_set_device(int device);
const char *pg_last_error(void);

int pg_scene_create(const PgSceneDesc *desc, PgScene **out);
void pg_scene_destroy(PgScene *scene);

/* HLBVH construction on the device: BVHAccel::HLBVHBuild (bvh.cpp:404-483: Morton codes of the centroids :415-429, radix
 * sort :432 / :140-180, treelets of equal top-12 Morton bits :437-456 built by emitLBVH :485-535, buildUpperSAH over the
 * treelet roots :537-638) followed by flattenBVHTree (:640-658).  bounds: n x {pMin.xyz, pMax.xyz} world bounds of the
 * primitives (host memory).  Outputs (host memory): nodes[] with room for 2n entries, *n_nodes, and ordered_prims[k] =
 * index of the primitive at position k of the leaf order.  The result is the reference's single-thread build bit for bit
 * (with several threads the reference permutes whole leaf ranges, never the tree).                                  */
int pg_hlbvh_build(int32_t n, const float *bounds, int32_t max_prims_in_node, PgBVHNode *nodes, int32_t *n_nodes, int32_t *ordered_prims);

/* Number of 16x16 tiles this (tile_first, tile_step) shard owns; pg_render
 * writes desc->tile_pixels PgFilmPixel entries per tile.                     */
int pg_render_tile_count(const PgRenderDesc *desc);

/* SamplerIntegrator::Render for the shard in desc: fills film[tile_pixels*tiles]
 * (tile-major, row-major inside the tile's block, zero outside the image) and
 * up to max_strays stray samples; *n_strays receives the count.
 * film/strays live in `mem` memory; `stream` is a hipStream_t (NULL = default). */
int pg_render(PgScene *scene, const PgRenderDesc *desc, PgFilmPixel *film,
              PgStraySample *strays, int32_t max_strays, int32_t *n_strays,
              int mem, void *stream);

/* The frame sharded over n devices of one node from ONE host process -- what main/pbrt.cpp:94-100 + tools/imgtool.cpp:190-285
 * do across machines with crop windows.  scenes[r] is the scene created on device r (pg_set_device + pg_scene_create, the
 * same PgSceneDesc everywhere); desc describes the whole frame (tile_first = 0, tile_step = 1).  One host thread per
 * device renders the tiles t = r (mod n) of the full-frame tiling into a packed shard [film | strays | count] on its own
 * device; the shards are then gathered on scenes[0]'s device by ONE ncclGather (RCCL over xGMI, rccl.h:745; librccl is opened
 * on first use) and come back to the host in one transfer:
 * film[r] receives rank r's tile_pixels * pg_render_tile_count(rank r's desc) pixels, strays[r] / n_strays[r] its stray
 * samples (max_strays entries each).  Rendering rank r alone with pg_render(tile_first = r, tile_step = n) gives the same
 * bytes.  Device ids may repeat (several shards on one GPU: a functional check on a single-GPU box): RCCL wants one rank
 * per GPU, so such a list -- like a box without librccl, or PG_SHARD_GATHER=peer -- is gathered with one peer-to-peer copy
 * per rank into the same layout.  PG_SHARD_GATHER=rccl makes the absence of RCCL an error instead.                     */
int pg_render_sharded(PgScene *const *scenes, int32_t n, const PgRenderDesc *desc, PgFilmPixel *const *film,
                      PgStraySample *const *strays, int32_t max_strays, int32_t *n_strays);
/* How the last pg_render_sharded of this process gathered: "rccl", "peer (<why RCCL was not used>)", or "none".          */
const char *pg_shard_transport(void);
/* pgh_box_filter_needs_gather() above as a symbol, for bindings that cannot use the C header's inline function */
int pg_box_filter_needs_gather(const PgRenderDesc *rd);

/* Batched Scene::Intersect: rays as SoA (ox..dz, tmax), n rays.  Outputs
 * prim (-1 = miss), t, b0, b1, b2 (barycentrics exactly as computed by
 * Triangle::Intersect, triangle.cpp:280-285).                                */
int pg_intersect(PgScene *scene, int32_t n, const float *o, const float *d,
                 const float *tmax, int32_t *prim, float *t, float *bary,
                 int mem, void *stream);
/* Batched Scene::IntersectP: occluded[i] = 1 if any hit.                     */
int pg_intersect_p(PgScene *scene, int32_t n, const float *o, const float *d,
                   const float *tmax, uint8_t *occluded, int mem, void *stream);

int pg_counters(PgScene *scene, PgCounters *out);
int pg_counters_reset(PgScene *scene);
int pg_scene_set_option(PgScene *scene, int32_t option, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* PBRT_GPU_H */
