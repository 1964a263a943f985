"""Seeded procedural "Sponza-class" atrium: the measurement stand-in for BASELINE configs 3-4.

The reference lists media/scenes/Sponza.gltf first in config.json (config.json:2-7) but the asset
is git-ignored and not available offline (reference .gitignore:11-16).
This generator emits a scene of the same class as the Khronos sample Sponza: ~262 k triangles,
an atrium of ~30 x 12 x 18 units (long axis x) with two storeys of colonnades (fluted columns +
arches), gallery slabs, outer walls, hanging cloth, floor clutter; >= 20 PBR materials mixing
metallic 0/1 and roughness 0..1; optional procedural RGBA8 base-colour / metallic-roughness /
normal textures; NO lights in the file, so the reference's 8 fallback point lights apply
(hello_vulkan.cpp:247-321; lightsCount = 8; only the first, (1,5,-1.33), is inside the building,
as with the real Sponza).  Everything is parametric surfaces with analytic-quality normals,
tangents and UVs.  Deterministic for a given (seed, target_triangles).

Harness-side measurement input, not part of the product library.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
import vkrt_amd  # noqa: E402,F401
from vkrt_amd.flat_scene import FlatScene, MAT_DTYPE, NODE_DTYPE, PRIM_DTYPE, fallback_lights  # noqa: E402

# default interior camera: from the west end, looking along +x down the nave
DEFAULT_CAMERA = dict(eye=(-12.5, 4.2, 0.6), center=(6.0, 3.6, -0.4), up=(0, 1, 0), fov=60.0)


def _surface(fn, nu, nv, flip=False, uv_scale=(1.0, 1.0), closed_u=False):
    """Tessellate P = fn(u, v), u,v in [0,1], into an (nu x nv)-quad grid with normals/tangents/uvs."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(u, v, indexing="xy")  # shape (nv+1, nu+1)
    P = fn(U, V)
    h = 1e-4
    Pu = (fn(np.clip(U + h, 0, 1) if not closed_u else U + h, V) - fn(np.clip(U - h, 0, 1) if not closed_u else U - h, V))
    Pv = (fn(U, np.clip(V + h, 0, 1)) - fn(U, np.clip(V - h, 0, 1)))
    N = np.cross(Pu, Pv)
    if flip:
        N = -N
    ln = np.linalg.norm(N, axis=-1, keepdims=True)
    bad = ln[..., 0] < 1e-12
    N = N / np.where(ln < 1e-12, 1.0, ln)
    N[bad] = np.array([0.0, 1.0, 0.0])
    T = Pu - np.sum(Pu * N, axis=-1, keepdims=True) * N
    lt = np.linalg.norm(T, axis=-1, keepdims=True)
    badt = lt[..., 0] < 1e-12
    T = T / np.where(lt < 1e-12, 1.0, lt)
    if badt.any():
        alt = np.cross(N, np.array([0.0, 0.0, 1.0]))
        la = np.linalg.norm(alt, axis=-1, keepdims=True)
        alt = np.where(la < 1e-6, np.cross(N, np.array([1.0, 0.0, 0.0])), alt)
        alt = alt / np.linalg.norm(alt, axis=-1, keepdims=True)
        T[badt] = alt[badt]
    w = np.where(np.sum(np.cross(N, T) * Pv, axis=-1) < 0, -1.0, 1.0)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nv + 1, nu + 1)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, 1:], idx[1:, :-1]
    if flip:
        tris = np.stack([a, c, b, a, d, c], axis=-1)
    else:
        tris = np.stack([a, b, c, a, c, d], axis=-1)
    return dict(
        pos=P.reshape(-1, 3).astype(np.float32),
        nrm=N.reshape(-1, 3).astype(np.float32),
        tan=np.concatenate([T.reshape(-1, 3), w.reshape(-1, 1)], axis=1).astype(np.float32),
        uv=np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1).reshape(-1, 2).astype(np.float32),
        idx=tris.reshape(-1).astype(np.uint32),
    )


def _merge(parts):
    off, out = 0, dict(pos=[], nrm=[], tan=[], uv=[], idx=[])
    for p in parts:
        for k in ("pos", "nrm", "tan", "uv"):
            out[k].append(p[k])
        out["idx"].append(p["idx"] + np.uint32(off))
        off += p["pos"].shape[0]
    return {k: np.concatenate(v, axis=0) for k, v in out.items()}


def _quad(p0, eu, ev, nu, nv, uv_scale=(1, 1), flip=False):
    p0, eu, ev = (np.asarray(x, np.float64) for x in (p0, eu, ev))
    return _surface(lambda U, V: p0 + U[..., None] * eu + V[..., None] * ev, nu, nv, flip=flip, uv_scale=uv_scale)


def _box(lo, hi, n=(1, 1, 1), uv=1.0):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    f = []
    f.append(_quad(lo, ez, ey, n[2], n[1], (uv * d[2], uv * d[1])))                 # -x
    f.append(_quad(lo + ex, ey, ez, n[1], n[2], (uv * d[1], uv * d[2])))            # +x
    f.append(_quad(lo, ex, ez, n[0], n[2], (uv * d[0], uv * d[2])))                 # -y
    f.append(_quad(lo + ey, ez, ex, n[2], n[0], (uv * d[2], uv * d[0])))            # +y
    f.append(_quad(lo, ey, ex, n[1], n[0], (uv * d[1], uv * d[0])))                 # -z
    f.append(_quad(lo + ez, ex, ey, n[0], n[1], (uv * d[0], uv * d[1])))            # +z
    return _merge(f)


def _lathe(profile, nu, nv, flutes=0, flute_depth=0.0, uv_scale=(4.0, 4.0)):
    """Surface of revolution about +y of profile(v) -> (radius, height); optional fluting."""
    def fn(U, V):
        r, y = profile(V)
        th = 2.0 * np.pi * U
        if flutes:
            r = r * (1.0 - flute_depth * (0.5 + 0.5 * np.cos(flutes * th)))
        return np.stack([r * np.cos(th), y, -r * np.sin(th)], axis=-1)
    return _surface(fn, nu, nv, uv_scale=uv_scale, closed_u=True)


def _column(height, radius, seg, rings):
    def shaft(V):
        # entasis: slight bulge, narrower at the top
        return radius * (1.0 - 0.18 * V + 0.06 * np.sin(np.pi * V)), 0.35 + (height - 0.8) * V
    def base(V):
        return radius * (1.55 - 0.45 * V + 0.12 * np.sin(3 * np.pi * V)), 0.35 * V
    def capital(V):
        return radius * (0.86 + 0.75 * V ** 1.5 + 0.08 * np.sin(4 * np.pi * V)), height - 0.45 + 0.45 * V
    parts = [
        _lathe(shaft, seg, rings, flutes=16, flute_depth=0.09),
        _lathe(base, seg, max(4, rings // 4)),
        _lathe(capital, seg, max(4, rings // 4)),
    ]
    return _merge(parts)


def _arch(span, rise, thick, depth, nseg, nprof):
    """Semicircular-ish arch spanning x in [0, span], swept rectangular profile, in the xy plane."""
    def make(side):
        # side: 0 intrados (inner), 1 extrados (outer), 2 front, 3 back
        def fn(U, V):
            th = np.pi * U
            r_in, r_out = span * 0.5 - 0.02, span * 0.5 + thick
            cx = span * 0.5
            if side == 0:
                r = np.full_like(U, r_in); z = depth * (V - 0.5)
            elif side == 1:
                r = np.full_like(U, r_out); z = depth * (0.5 - V)
            elif side == 2:
                r = r_in + (r_out - r_in) * V; z = np.full_like(U, 0.5 * depth)
            else:
                r = r_out - (r_out - r_in) * V; z = np.full_like(U, -0.5 * depth)
            return np.stack([cx - r * np.cos(th), rise * np.sin(th) * r / r_in, z], axis=-1)
        return _surface(fn, nseg, nprof, uv_scale=(6.0, 1.0))
    return _merge([make(s) for s in range(4)])


def _cloth(width, drop, nu, nv, rng):
    ph = rng.uniform(0, 2 * np.pi, 4)
    amp = rng.uniform(0.08, 0.2)
    def fn(U, V):
        x = width * (U - 0.5)
        y = -drop * V
        sag = 0.35 * np.sin(np.pi * U) * (0.3 + 0.7 * V)
        z = amp * np.sin(7.0 * np.pi * U + ph[0]) * (0.25 + V) + 0.05 * np.sin(19.0 * np.pi * U + 5.0 * V + ph[1]) \
            + 0.04 * np.sin(11.0 * V + ph[2])
        return np.stack([x, y - sag * 0.4, z], axis=-1)
    return _surface(fn, nu, nv, uv_scale=(3.0, 3.0))


def _blob(radius, nu, nv, rng):
    k = rng.uniform(0.0, 0.25, 3)
    ph = rng.uniform(0, 2 * np.pi, 3)
    def fn(U, V):
        th, phi = 2 * np.pi * U, np.pi * (0.001 + 0.998 * V)
        r = radius * (1.0 + k[0] * np.sin(3 * th + ph[0]) * np.sin(phi) + k[1] * np.sin(5 * phi + ph[1]) * 0.5)
        return np.stack([r * np.sin(phi) * np.cos(th), -r * np.cos(phi) + radius, -r * np.sin(phi) * np.sin(th)], axis=-1)
    return _surface(fn, nu, nv, uv_scale=(2.0, 1.0), closed_u=True)


def _vase(height, radius, nu, nv, rng):
    a = rng.uniform(0.2, 0.5)
    def prof(V):
        return radius * (0.55 + a * np.sin(np.pi * (0.15 + 0.8 * V)) ** 2 + 0.12 * V), height * V
    return _lathe(prof, nu, nv, uv_scale=(2.0, 2.0))


def _trs(t=(0, 0, 0), ry=0.0, s=(1, 1, 1)):
    c, sn = np.cos(ry), np.sin(ry)
    R = np.array([[c, 0, sn, 0], [0, 1, 0, 0], [-sn, 0, c, 0], [0, 0, 0, 1]], np.float64)
    S = np.diag([s[0], s[1], s[2], 1.0])
    T = np.eye(4)
    T[:3, 3] = t
    return T @ R @ S


# ---- procedural textures ---------------------------------------------------------------------------
def _value_noise(n, cells, rng):
    g = rng.uniform(0, 1, (cells + 1, cells + 1))
    g[-1, :] = g[0, :]
    g[:, -1] = g[:, 0]
    t = np.linspace(0, cells, n, endpoint=False)
    i = t.astype(int)
    f = t - i
    f = f * f * (3 - 2 * f)
    a = g[np.ix_(i, i)]; b = g[np.ix_(i, i + 1)]; c = g[np.ix_(i + 1, i)]; d = g[np.ix_(i + 1, i + 1)]
    fx, fy = f[None, :], f[:, None]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def _fbm(n, rng, octaves=4):
    out = np.zeros((n, n))
    amp, cells, tot = 1.0, 4, 0.0
    for _ in range(octaves):
        out += amp * _value_noise(n, cells, rng)
        tot += amp
        amp *= 0.5
        cells *= 2
    return out / tot


def make_textures(rng, size=512):
    """Returns a list of {"rgba8","is_srgb"}: [0..3] base colour (sRGB), [4..5] metallic-roughness
    (G = roughness, B = metalness; linear), [6..7] tangent-space normal maps (linear)."""
    tex = []
    yy, xx = np.mgrid[0:size, 0:size] / float(size)
    def rgba(r, g, b, a=None):
        a = np.ones_like(r) if a is None else a
        return (np.clip(np.stack([r, g, b, a], -1), 0, 1) * 255.0 + 0.5).astype(np.uint8)
    n0 = _fbm(size, rng)
    brick = ((np.floor(yy * 16) % 2) * 0.5 + xx * 8) % 1.0
    mortar = ((brick < 0.06) | ((yy * 16) % 1.0 < 0.1)).astype(float)
    tex.append(dict(rgba8=rgba(0.75 - 0.3 * mortar + 0.15 * n0, 0.62 - 0.3 * mortar + 0.12 * n0, 0.5 - 0.25 * mortar + 0.1 * n0), is_srgb=True))
    n1 = _fbm(size, rng)
    checker = ((np.floor(xx * 8) + np.floor(yy * 8)) % 2)
    tex.append(dict(rgba8=rgba(0.55 + 0.35 * checker * n1, 0.5 + 0.3 * checker, 0.45 + 0.3 * (1 - checker) * n1), is_srgb=True))
    n2 = _fbm(size, rng, 5)
    stripes = 0.5 + 0.5 * np.sin(xx * 40 * np.pi)
    tex.append(dict(rgba8=rgba(0.3 + 0.6 * stripes * n2, 0.1 + 0.2 * n2, 0.12 + 0.5 * (1 - stripes)), is_srgb=True))
    n3 = _fbm(size, rng, 5)
    tex.append(dict(rgba8=rgba(0.8 * n3 + 0.15, 0.78 * n3 + 0.15, 0.7 * n3 + 0.12), is_srgb=True))
    n4 = _fbm(size, rng)
    tex.append(dict(rgba8=rgba(np.zeros_like(n4), 0.25 + 0.7 * n4, (n4 > 0.55).astype(float)), is_srgb=False))
    n5 = _fbm(size, rng)
    tex.append(dict(rgba8=rgba(np.zeros_like(n5), 0.1 + 0.5 * n5, np.ones_like(n5)), is_srgb=False))
    for _ in range(2):
        hgt = _fbm(size, rng, 5)
        dx = np.roll(hgt, -1, 1) - np.roll(hgt, 1, 1)
        dy = np.roll(hgt, -1, 0) - np.roll(hgt, 1, 0)
        nz = np.ones_like(hgt)
        nn = np.stack([-dx * 24.0, -dy * 24.0, nz], -1)
        nn /= np.linalg.norm(nn, axis=-1, keepdims=True)
        tex.append(dict(rgba8=rgba(nn[..., 0] * 0.5 + 0.5, nn[..., 1] * 0.5 + 0.5, nn[..., 2] * 0.5 + 0.5), is_srgb=False))
    return tex


def make_materials(rng, with_textures, emissive=False):
    n = 24
    m = np.zeros(n, MAT_DTYPE)
    for k in ("pbrBaseColorTexture", "metallicRoughnessTexture", "normalTexture", "emissiveTexture"):
        m[k] = -1
    base = rng.uniform(0.25, 0.95, (n, 3))
    m["pbrBaseColorFactor"][:, :3] = base
    m["pbrBaseColorFactor"][:, 3] = 1.0
    # half dielectric (metallic 0), a quarter metal (metallic 1), the rest in between
    metal = np.zeros(n)
    metal[n // 2: 3 * n // 4] = 1.0
    metal[3 * n // 4:] = rng.uniform(0.1, 0.9, n - 3 * n // 4)
    m["metallicFactor"] = metal
    m["roughnessFactor"] = np.concatenate([rng.uniform(0.35, 1.0, n // 2), rng.uniform(0.0, 0.6, n - n // 2)])
    m["roughnessFactor"][[n // 2, n // 2 + 1]] = (0.0, 1.0)  # clamp edge cases (rchit:128-129)
    if with_textures:
        for i in range(n):
            if i % 3 != 2:
                m["pbrBaseColorTexture"][i] = i % 4
                m["pbrBaseColorFactor"][i, :3] = 0.6 + 0.4 * base[i]
            if i % 4 == 1:
                m["metallicRoughnessTexture"][i] = 4 + (i // 4) % 2
                m["metallicFactor"][i] = 1.0
                m["roughnessFactor"][i] = 1.0
            if i % 5 in (0, 3):
                m["normalTexture"][i] = 6 + (i // 5) % 2
    if emissive and with_textures:
        # raytrace.rchit:83-87 / frag_shader.frag:190-192: emissiveFactor * texture(emissiveTexture) (an sRGB image,
        # hello_vulkan.cpp:417-443), on dielectrics, on a mirror-like metal (so emission is also picked up at depth > 0,
        # after a specular bounce) and on a normal-mapped material
        for i, tex in ((0, 2), (3, 1), (5, 2), (12, 3), (13, 0), (16, 2), (21, 1)):
            m["emissiveTexture"][i] = tex
            m["emissiveFactor"][i] = (0.9, 0.55 + 0.02 * i, 0.2 + 0.03 * i)
        m["emissiveFactor"][7] = (0.3, 0.05, 0.4)  # factor without a texture
    return m


def mixed_lights():
    """Scene lights of all three KHR_lights_punctual kinds as loadGltfLights maps them (hello_vulkan.cpp:207-223:
    point 0, directional 1, spot 2; position = translation of the light node).  The path tracer's directLight only
    evaluates type 0 (gltf.glsl:136-154); frag_shader.frag:205-208 treats every other type as directional."""
    from vkrt_amd.flat_scene import LIGHT_DTYPE
    L = np.zeros(5, LIGHT_DTYPE)
    L[0] = ((1.0, 5.0, -1.33), (1.0, 0.95, 0.9), 50.0, 0)
    L[1] = ((0.35, 1.0, 0.2), (1.0, 0.9, 0.7), 1.5, 1)     # directional
    L[2] = ((-6.0, 9.0, 2.0), (0.6, 0.8, 1.0), 40.0, 2)    # spot
    L[3] = ((8.0, 3.0, 0.5), (1.0, 0.4, 0.3), 30.0, 0)
    L[4] = ((-9.0, 7.5, -6.5), (0.2, 1.0, 0.4), 25.0, 1)   # directional, from inside the building
    return L


def build_atrium(target_triangles=262144, seed=1, with_textures=True, variant=None):
    """Returns (FlatScene, info dict) with exactly `target_triangles` instanced triangles
    (the tessellation scale is calibrated downwards, then small clutter tops the count up).
    variant="emissive_mixed_lights": same geometry; seven materials get an sRGB emissive texture and the file carries
    five lights (2 point, 2 directional, 1 spot) instead of relying on the fallback lights.
    variant="nonuniform": the tessellation of an artist-made scene like the real Sponza instead of uniform grids: floor, walls
    and slabs are a handful of room-sized triangles; 3-cm mouldings, cornices and pilaster edges run the length of the building
    as needle triangles (aspect up to 500 : 1); the drapery is long thin strips with a second, nearly coincident layer; banners
    hang as 20-cm ribbons; the triangle budget goes into dense small detail instead (fluted columns, statues' worth of blobs).
    The boxes of the big and the thin triangles overlap hundreds of small ones: the case spatial splits exist for."""
    d = float(np.sqrt(target_triangles / 360000.0))
    for _ in range(8):
        flat, info = _build_atrium(target_triangles, seed, with_textures, d, variant)
        if info["triangles"] <= target_triangles + 64:
            return flat, info
        d *= float(np.sqrt(target_triangles / info["triangles"])) * 0.985
    return flat, info


def _build_atrium(target_triangles, seed, with_textures, d, variant=None):
    rng = np.random.default_rng(seed)
    def n_(x, lo=2):
        return max(lo, int(round(x * d)))

    meshes, nodes = [], []  # meshes: (geom dict, material) ; nodes: (matrix, mesh index)
    def add_mesh(geom, mat):
        meshes.append((geom, mat))
        return len(meshes) - 1
    def place(mi, M):
        nodes.append((M, mi))

    X0, X1, Y1, Z0, Z1 = -15.0, 15.0, 12.0, -9.0, 9.0
    GZ = 4.6       # colonnade line |z|
    H1 = 5.4       # storey height
    # "nonuniform" = all four ingredients of the artist-like tessellation; "nonuniform:walls,trim" etc. = a subset (attribution runs):
    #   walls = room-sized floor / wall / slab triangles, trim = mouldings + pilasters + ribbons (needles), cloth = drapery as strips,
    #   layer = the second drapery layer 1 cm in front of the first
    parts = set()
    if variant is not None and variant.startswith("nonuniform"):
        parts = set(variant.split(":", 1)[1].split(",")) if ":" in variant else {"walls", "trim", "cloth", "layer"}
        assert parts <= {"walls", "trim", "cloth", "layer"}, variant
    else:
        assert variant in (None, "emissive_mixed_lights"), variant
    coarse = "walls" in parts
    def big(x):  # grid resolution of the large flat surfaces: a few room-sized triangles in the non-uniform variant
        return 1 if coarse else n_(x)
    # floor + roof over the galleries + outer walls (large polygons, coarse grids)
    place(add_mesh(_quad((X0, 0, Z1), (X1 - X0, 0, 0), (0, 0, Z0 - Z1), 2 if coarse else n_(60), 2 if coarse else n_(36), (15, 9)), 0), np.eye(4))
    for zs in (-1, 1):
        z_in, z_out = zs * GZ, zs * Z1
        zl, zh = min(z_in, z_out), max(z_in, z_out)
        place(add_mesh(_box((X0, H1, zl), (X1, H1 + 0.5, zh), (big(30), 1, big(6)), uv=0.5), 1), np.eye(4))           # gallery slab
        place(add_mesh(_box((X0, 2 * H1 + 0.6, zl), (X1, Y1, zh), (big(30), 1, big(6)), uv=0.5), 2), np.eye(4))        # roof slab
        wall = _quad((X0, 0, zs * Z1), (X1 - X0, 0, 0), (0, Y1, 0), big(40), big(16), (10, 4), flip=(zs > 0))
        place(add_mesh(wall, 3), np.eye(4))
    for xs in (-1, 1):
        wall = _quad((xs * X1, 0, Z0), (0, 0, Z1 - Z0), (0, Y1, 0), big(24), big(16), (6, 4), flip=(xs < 0))
        place(add_mesh(wall, 4), np.eye(4))
    if "trim" in parts:
        # mouldings and cornices: 3-cm profiles running the length of the building in two pieces (needle triangles, 15 m x 3 cm),
        # on the outer walls, under the gallery slabs and along the colonnade lines; pilaster edges up the walls
        trim = add_mesh(_box((0, 0, 0), (X1 - X0, 0.03, 0.03), (2, 1, 1)), 9)
        for zs in (-1, 1):
            for y in (0.02, 1.1, 2.9, H1 - 0.05, H1 + 0.52, H1 + 1.6, 2 * H1 + 0.55, Y1 - 0.4):
                place(trim, _trs((X0, y, zs * (Z1 - 0.04) - 0.015)))
            for y in (H1 - 0.02, H1 + 0.5, 2 * H1 + 0.58):
                place(trim, _trs((X0, y, zs * (GZ - 0.55))))
                place(trim, _trs((X0, y, zs * (GZ + 0.5))))
        pil = add_mesh(_box((0, 0, 0), (0.04, Y1, 0.04), (1, 2, 1)), 4)
        for zs in (-1, 1):
            for x in np.linspace(X0 + 0.8, X1 - 0.8, 14):
                place(pil, _trs((x, 0, zs * (Z1 - 0.05) - 0.02)))
        # banners: 20-cm ribbons hanging the height of a storey, long thin triangles in one column
        for i in range(10):
            rib = add_mesh(_quad((0, 0, 0), (0.2, 0, 0), (0, -5.0, 0.02 * i), 1, n_(40, 8)), 10 + i % 4)
            place(rib, _trs((X0 + 3.0 + 2.6 * i, 2 * H1 + 0.4, (-1 if i % 2 else 1) * 3.6), ry=0.3 * i))

    # columns: one mesh per storey, instanced (TLAS-instance semantics, hello_vulkan.cpp:1035-1043)
    ncol = 12
    xs_col = np.linspace(X0 + 1.6, X1 - 1.6, ncol)
    col_lo = add_mesh(_column(H1, 0.42, n_(40, 8), n_(26, 4)), 5)
    col_hi = add_mesh(_column(H1 - 0.2, 0.34, n_(40, 8), n_(22, 4)), 6)
    for zs in (-1, 1):
        for i, x in enumerate(xs_col):
            place(col_lo, _trs((x, 0, zs * GZ), ry=0.37 * i))
            place(col_hi, _trs((x, H1 + 0.5, zs * GZ), ry=0.21 * i, s=(1.0, 1.0, 1.0 + 0.15 * (i % 3))))
    # arches between neighbouring columns on both storeys
    span = float(xs_col[1] - xs_col[0])
    arch_lo = add_mesh(_arch(span, 1.25, 0.28, 0.7, n_(28, 6), n_(3, 1)), 7)
    arch_hi = add_mesh(_arch(span, 1.05, 0.22, 0.6, n_(24, 6), n_(3, 1)), 8)
    for zs in (-1, 1):
        for i in range(ncol - 1):
            place(arch_lo, _trs((xs_col[i], H1 - 1.45, zs * GZ)))
            place(arch_hi, _trs((xs_col[i], 2 * H1 - 0.85, zs * GZ)))
    # balustrade rail boxes on the upper gallery
    rail = add_mesh(_box((0, 0, -0.06), (span - 0.5, 0.12, 0.06), (n_(6), 1, 1)), 9)
    for zs in (-1, 1):
        for i in range(ncol - 1):
            place(rail, _trs((xs_col[i] + 0.25, H1 + 1.45, zs * (GZ - 0.3))))

    # hanging cloth (unique meshes: dense, thin, non-planar)
    ncloth = 8
    for i in range(ncloth):
        # non-uniform variant: the drapery as long thin strips (many columns, few rows) with a second layer 1 cm in front of it
        cl = add_mesh(_cloth(3.2, 4.2, n_(260, 8), n_(10, 2), rng) if "cloth" in parts else _cloth(3.2, 4.2, n_(84, 8), n_(72, 8), rng), 10 + i % 4)
        x = X0 + 4.0 + (X1 - X0 - 8.0) * (i // 2) / max(1, ncloth // 2 - 1)
        place(cl, _trs((x, 9.6, (-1 if i % 2 else 1) * 2.1), ry=0.5 * np.pi + 0.1 * i))
        if "layer" in parts:
            place(cl, _trs((x + 0.01, 9.58, (-1 if i % 2 else 1) * 2.1), ry=0.5 * np.pi + 0.1 * i + 0.004))

    # floor clutter: blobs and vases, unique and instanced
    blob_meshes = [add_mesh(_blob(rng.uniform(0.3, 0.6), n_(40, 8), n_(24, 6), rng), 14 + i % 6) for i in range(10)]
    vase_meshes = [add_mesh(_vase(rng.uniform(0.7, 1.4), rng.uniform(0.2, 0.4), n_(36, 8), n_(28, 6), rng), 18 + i % 6) for i in range(8)]
    for i in range(44):
        mi = blob_meshes[i % len(blob_meshes)] if i % 2 == 0 else vase_meshes[i % len(vase_meshes)]
        x = rng.uniform(X0 + 1.5, X1 - 1.5)
        z = rng.uniform(-GZ + 0.9, GZ - 0.9) if i % 3 else rng.choice([-1, 1]) * rng.uniform(GZ + 0.8, Z1 - 0.8)
        y = 0.0 if i % 5 else H1 + 0.5
        sc = rng.uniform(0.7, 1.5)
        place(mi, _trs((x, y, z), ry=rng.uniform(0, 2 * np.pi), s=(sc, sc * rng.uniform(0.8, 1.3), sc)))

    def count_tris():
        return sum(len(meshes[mi][0]["idx"]) // 3 for _, mi in nodes)

    # top up with extra detailed clutter until the instanced triangle count reaches the target
    guard = 0
    while count_tris() < target_triangles and guard < 4000:
        guard += 1
        remaining = target_triangles - count_tris()
        if remaining < 2:
            break
        nu = int(np.clip(np.sqrt(remaining / 2.0 * 1.6), 1, 48))
        nv = int(np.clip(remaining // (2 * nu), 1, 32))
        mi = add_mesh(_blob(rng.uniform(0.15, 0.4), nu, nv, rng), int(rng.integers(0, 24)))
        place(mi, _trs((rng.uniform(X0 + 1, X1 - 1), 0.0, rng.uniform(-GZ + 0.6, GZ - 0.6)), ry=rng.uniform(0, 6.28)))

    # ---- flatten into the reference's arrays ------------------------------------------------------
    P, N, T, UV, IDX = [], [], [], [], []
    prims = np.zeros(len(meshes), PRIM_DTYPE)
    vo = io = 0
    for i, (g, mat) in enumerate(meshes):
        P.append(g["pos"]); N.append(g["nrm"]); T.append(g["tan"]); UV.append(g["uv"]); IDX.append(g["idx"])
        prims[i] = (io, len(g["idx"]), vo, g["pos"].shape[0], mat)
        vo += g["pos"].shape[0]
        io += len(g["idx"])
    nd = np.zeros(len(nodes), NODE_DTYPE)
    for i, (M, mi) in enumerate(nodes):
        nd[i]["worldMatrix"] = np.asarray(M, np.float32).T.reshape(-1)
        nd[i]["primMesh"] = mi
    textures = make_textures(rng) if with_textures else []
    emis = variant == "emissive_mixed_lights"
    flat = FlatScene(np.concatenate(P), np.concatenate(N), np.concatenate(T), np.concatenate(UV), np.concatenate(IDX),
                     prims, make_materials(rng, with_textures, emissive=emis), mixed_lights() if emis else fallback_lights(), nd, textures)
    info = dict(triangles=flat.instanced_triangle_count, unique_triangles=int(sum(len(g["idx"]) // 3 for g, _ in meshes)),
                prim_meshes=len(meshes), nodes=len(nodes), materials=24, textures=len(textures), seed=seed, tess_scale=d,
                bounds=((X0, 0.0, Z0), (X1, Y1, Z1)), camera=DEFAULT_CAMERA)
    return flat, info


if __name__ == "__main__":
    target = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    sc, info = build_atrium(target)
    print(info)


def rotate_scene(flat, camera, ry_deg, rx_deg=0.0):
    """The whole scene turned about y and then tilted about x (degrees), in place, with its camera: no large triangle stays aligned with
    the coordinate axes -- the case in which triangle pre-splitting pays (profiles/r05_split_rotated.jsonl).  Lights (the reference's
    fallback lights, hello_vulkan.cpp:247-321) stay where they are.  Returns the rotated camera keywords."""
    ry, rx = np.deg2rad(ry_deg), np.deg2rad(rx_deg)
    Ry = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    R4 = np.eye(4)
    R4[:3, :3] = Rx @ Ry
    wm = flat.nodes["worldMatrix"].reshape(-1, 4, 4).astype(np.float64)  # column-major storage: the stored 4x4 is M^T
    flat.nodes["worldMatrix"] = np.einsum("nij,jk->nik", wm, R4.T).reshape(flat.nodes["worldMatrix"].shape).astype(np.float32)
    cam = dict(camera)
    for key in ("eye", "center", "up"):
        cam[key] = tuple((R4[:3, :3] @ np.asarray(cam[key], np.float64)).tolist())
    return cam
